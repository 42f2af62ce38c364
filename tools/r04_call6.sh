#!/bin/bash
# round 4, call 6: the T-conv chain kernel (tchain.hip) + the four-launch T-ConvFFN forward on the device — parity, large train step with / without, kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_large.py -m gpu -x -q -k "tconvffn or network or dropin" 2>&1 | tail -3
python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04d_large_chain.json
NBSS_TCHAIN_OFF=1 python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04d_large_nochain.json
python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r04d_large_chain_b8.json
bash tools/large_prof.sh 4 2>&1 | tail -45
cp gpurun_out/large_rocprof.md gpurun_out/r04d_large_rocprof.md
