#!/bin/bash
# round 4, call 8: generic attention with transposing reads, staged GEMM stores, chain copy-out, all-groups T-conv weight gradient — parity, rates, trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_large.py tests/test_gemm_g.py tests/test_nbc2_native.py -m gpu -x -q -k "not headline" 2>&1 | tail -3
python tools/gemm_g_bench.py 2>&1 | tail -1 | tee gpurun_out/r04f_gemm_tile.json
python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04f_large.json
python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r04f_large_b8.json
bash tools/large_prof.sh 4 2>&1 | tail -42
cp gpurun_out/large_rocprof.md gpurun_out/r04f_large_rocprof.md
