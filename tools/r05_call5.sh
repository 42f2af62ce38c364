cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_determinism.py tests/test_kernels_bwd.py tests/test_train_step.py tests/test_side_stream.py tests/test_nb_models.py tests/test_nbc2_native.py -m gpu -q -x 2>&1 | tail -8
bash tools/ab.sh "prev prod" ""
