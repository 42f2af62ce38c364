#!/bin/bash
# kernel traces of the same build, two-stream (default) and in order (NBSS_SIDE_STREAM=0): tools/r04_trace.sh <tag>
TAG=${1:-r04a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for MODE in two inorder; do
  [ $MODE = inorder ] && export NBSS_SIDE_STREAM=0 || unset NBSS_SIDE_STREAM
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
  python tools/rocprof_summary.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_rocprof_${MODE}.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (batch 32; 7 steps + one-time table/pack kernels); walks: ${MODE}; commit ${NBSS_COMMIT}"
  python tools/gpu_idle.py gpurun_out/${TAG}_prof | head -12 | tee gpurun_out/${TAG}_gpu_idle_${MODE}.txt
  find gpurun_out/${TAG}_prof -name "*.db" -delete; rm -rf gpurun_out/${TAG}_prof
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$MODE', round(d['value'],1), 'utt/s', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if v > 0.3})"
done
