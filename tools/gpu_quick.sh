#!/bin/bash
# A short GPU-box call: the -m gpu suite, the default bench line, a rocprofv3 --stats summary and the PMC passes of a few kernels.
#   tools/gpu_quick.sh <tag> [pytest-args]      NBSS_PMC_KERNELS="tconvffn_bwd mhsa_bwd" selects the PMC passes ("" = none)
TAG=${1:-r02q}
shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
python - <<PY
import json
b = json.loads(open("gpurun_out/${TAG}_bench.json").read())
print(b["value"], b["utt_per_s_by_batch"], b["cpu_baseline"] and b["cpu_baseline"]["value"])
print(json.dumps(b["kernel_ms_per_step"]))
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_rocprof.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (batch 32; 7 steps + one-time table/pack kernels); commit ${NBSS_COMMIT}"
python tools/gpu_idle.py gpurun_out/${TAG}_prof | tee gpurun_out/${TAG}_gpu_idle.txt
python tools/wgrad_breakdown.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_wgrad_breakdown.txt
find gpurun_out/${TAG}_prof -name "*.db" -delete
head -30 gpurun_out/${TAG}_rocprof.md
if [ -n "${NBSS_PMC_KERNELS}" ]; then
  bash tools/pmc_traffic.sh 32 > /dev/null 2>&1; python tools/pmc_traffic.py 32 | grep -E "ratio|hbm_bytes|_fwd|_bwd"; rm -rf gpurun_out/traffic
  bash tools/pmc_mfma.sh 32 > /dev/null 2>&1; python tools/pmc_mfma.py 32; rm -rf gpurun_out/mfma
fi
