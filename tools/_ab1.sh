cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh "2 8" 20 "NBSS_FOLD_TWO_PASS=1 NBSS_FOLD_TWO_PASS=0 NBSS_FOLD_TWO_PASS=1 NBSS_FOLD_TWO_PASS=0"
bash tools/ab_env.sh "32" 10 "NBSS_FOLD_TWO_PASS=1 NBSS_FOLD_TWO_PASS=0 NBSS_FOLD_TWO_PASS=1 NBSS_FOLD_TWO_PASS=0"
timeout 300 python -m pytest tests/test_determinism.py tests/test_train_step.py -m gpu -x -q 2>&1 | tail -3
