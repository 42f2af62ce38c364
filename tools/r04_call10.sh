#!/bin/bash
# round 4, call 10: per-workgroup affine rows + reduce instead of same-address atomics (LN backward, fconv_g, tchain) — parity, rates, trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_large.py -m gpu -x -q -k "(bwd and not headline) or train_step" 2>&1 | tail -3
python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04h_large.json
python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r04h_large_b8.json
bash tools/large_prof.sh 4 2>&1 | tail -40
cp gpurun_out/large_rocprof.md gpurun_out/r04h_large_rocprof.md
