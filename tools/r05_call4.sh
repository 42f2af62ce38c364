cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_graph_step.py tests/test_nbc_native.py tests/test_nbc2_native.py -m gpu -q -x 2>&1 | tail -5
bash tools/ab.sh "prev prod" "full_bwd fconv_bwd mhsa_bwd mhsa_fwd"
