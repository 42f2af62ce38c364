cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_determinism.py tests/test_kernels_bwd.py -m gpu -q -x 2>&1 | tail -4
bash tools/ab.sh "prev prod prev prod" ""
NBSS_SIDE_STREAM=0 bash tools/ab.sh "prev prod" ""
