#!/bin/bash
# round 5, first call (prepared at the end of round 4, when the GPU budget was spent): what round 4 left unverified on the device, then the baselines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 1. the native NBC inference path (nbss_amd/nbc.py; emulator-verified only) against the torch.nn module on the device
NBSS_RUN_UNVERIFIED=1 timeout 180 python -m pytest tests/test_nbc_native.py -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r05a_nbc_native.log
# 2. large train step: rates at batch 4 / 8, the knobs one at a time (what each round-4 kernel is worth in the final build)
for kv in "" NBSS_WGRAD_TILE=0 NBSS_TCHAIN_OFF=1 NBSS_FCONVG_OFF=1 NBSS_GEMM_V1=1; do
  echo "== ${kv:-default}"; env $kv timeout 60 python tools/large_rate.py 4 3 2>&1 | tail -1
done | tee gpurun_out/r05a_large_knobs.txt
timeout 60 python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r05a_large_b8.json
# 3. in-order trace of the large step (pure kernel times: the two-stream trace stretches the gradient stream's launches)
NBSS_SIDE_STREAM=0 timeout 120 bash tools/large_prof.sh 4 2>&1 | tail -36
cp gpurun_out/large_rocprof.md gpurun_out/r05a_large_rocprof_inorder.md
