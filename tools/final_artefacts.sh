#!/bin/bash
# Reduced end-of-round collection (GPU budget nearly spent): full -m gpu suite, smoke, default bench line, rocprofv3 --stats summary,
# PMC traffic / MFMA passes of the kernels listed in NBSS_PMC_KERNELS (merged into the tracked per-kernel files by the caller).
TAG=${1:-r02e}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${TAG}_pytest_gpu.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-200 gpurun_out/${TAG}_bench.json
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_rocprof.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (batch 32; 7 steps + one-time table/pack kernels); commit ${NBSS_COMMIT}"
python tools/gpu_idle.py gpurun_out/${TAG}_prof | head -2 | tee gpurun_out/${TAG}_gpu_idle.txt
python tools/wgrad_breakdown.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_wgrad_breakdown.txt
find gpurun_out/${TAG}_prof -name "*.db" -delete; rm -rf gpurun_out/${TAG}_prof
bash tools/pmc_traffic.sh 32 > /dev/null 2>&1; python tools/pmc_traffic.py 32 | grep -E "ratio|hbm_bytes|_bwd|_fwd"; rm -rf gpurun_out/traffic
bash tools/pmc_mfma.sh 32 > /dev/null 2>&1; python tools/pmc_mfma.py 32 | tail -20; rm -rf gpurun_out/mfma
