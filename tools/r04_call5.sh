#!/bin/bash
# round 4, call 5: the dense tile GEMM (gemm_g.hip) on the device — parity, per-problem rate against the previous kernel, large train step with each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gemm_g.py -m gpu -x -q 2>&1 | tail -3
python tools/gemm_g_bench.py 2>&1 | tail -1 | tee gpurun_out/r04c_gemm_tile.json
NBSS_GEMM_V1=1 python tools/gemm_g_bench.py 2>&1 | tail -1 | tee gpurun_out/r04c_gemm_v1.json
python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04c_large_tile.json
NBSS_GEMM_V1=1 python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04c_large_v1.json
timeout 400 python -m pytest tests/test_large.py -m gpu -x -q 2>&1 | tail -3
