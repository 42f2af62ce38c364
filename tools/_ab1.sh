cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh "32" 10 "NBSS_HIP_FLAVOUR=prev NBSS_X=cur NBSS_HIP_FLAVOUR=prev NBSS_X=cur NBSS_HIP_FLAVOUR=prev NBSS_X=cur"
export NBSS_SIDE_STREAM=0
bash tools/ab_env.sh "32" 10 "NBSS_HIP_FLAVOUR=prev NBSS_X=cur NBSS_HIP_FLAVOUR=prev NBSS_X=cur"
unset NBSS_SIDE_STREAM
bash tools/ab_env.sh "4 16" 20 "NBSS_HIP_FLAVOUR=prev NBSS_X=cur"
