#!/bin/bash
# round 4, call 7: fconv_g.hip (fused F-conv backward), gemm_g.hip epilogue without waits, tchain.hip with the fast SiLU / hoisted loads — parity, rates, trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_large.py tests/test_gemm_g.py -m gpu -x -q -k "fconv_bwd or tconvffn_bwd or gemm or network_forward" 2>&1 | tail -3
python tools/gemm_g_bench.py 2>&1 | tail -1 | tee gpurun_out/r04e_gemm_tile.json
python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04e_large.json
NBSS_FCONVG_OFF=1 python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04e_large_nofconvg.json
python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r04e_large_b8.json
bash tools/large_prof.sh 4 2>&1 | tail -42
cp gpurun_out/large_rocprof.md gpurun_out/r04e_large_rocprof.md
