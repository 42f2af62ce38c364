cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_bwd.py -m gpu -q -x -k "tconvffn" 2>&1 | tail -3
bash tools/ab.sh "prev prod prev prod" "tconvffn_bwd"
