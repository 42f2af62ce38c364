#!/bin/bash
# where a small-batch step goes: kernel-trace at batch 2 and 8 (busy vs wall, per-kernel table)
cd $GRAFT_REPO_ROOT
for b in 2 8; do
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/sb${b}_prof && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sb${b}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --batch $b --no-cpu-baseline > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/sb${b}_prof gpurun_out/sb${b}_rocprof.md "batch $b, 12 steps"
python tools/gpu_idle.py gpurun_out/sb${b}_prof | head -3 | tee gpurun_out/sb${b}_gpu_idle.txt
find gpurun_out/sb${b}_prof -name "*.db" -delete
done
