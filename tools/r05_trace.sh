#!/bin/bash
# kernel traces of the same build: walks in order (pure kernel times) and with the second stream (the default)   tools/r05_trace.sh <tag>
TAG=${1:-r05a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for MODE in inorder two; do
  [ $MODE = inorder ] && export NBSS_SIDE_STREAM=0 || unset NBSS_SIDE_STREAM
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
  python tools/rocprof_summary.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_rocprof_${MODE}.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (batch 32; 7 steps + one-time table/pack kernels); walks: ${MODE}; commit ${NBSS_COMMIT}"
  python tools/gpu_idle.py gpurun_out/${TAG}_prof | head -2 | tee gpurun_out/${TAG}_gpu_idle_${MODE}.txt
  find gpurun_out/${TAG}_prof -name "*.db" -delete; rm -rf gpurun_out/${TAG}_prof
done
head -34 gpurun_out/${TAG}_rocprof_inorder.md | cut -c1-120
