#!/bin/bash
# round 4, call 9: gemm_g with per-tile address setup, 8-lane LayerNorm kernels — parity of the touched paths, rates, trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_large.py tests/test_gemm_g.py -m gpu -x -q -k "gemm or (bwd and not headline) or network_forward" 2>&1 | tail -3
python tools/gemm_g_bench.py 2>&1 | tail -1 | tee gpurun_out/r04g_gemm_tile.json
python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04g_large.json
python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r04g_large_b8.json
bash tools/large_prof.sh 4 2>&1 | tail -40
cp gpurun_out/large_rocprof.md gpurun_out/r04g_large_rocprof.md
