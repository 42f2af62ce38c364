#!/bin/bash
# where a small-batch step goes: kernel-trace at batch 2 and 8, walks in order and with the gradient stream (busy vs wall, per-kernel table)
#   tools/small_batch_prof.sh [tag]
TAG=${1:-sb}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 2 8; do for MODE in inorder two; do
[ $MODE = inorder ] && export NBSS_SIDE_STREAM=0 || unset NBSS_SIDE_STREAM
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}${b}_prof
( cd /tmp && export TMPDIR=/tmp && rm -rf $P && timeout 200 rocprofv3 --kernel-trace --stats -d $P -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --batch $b --no-cpu-baseline > /dev/null 2>&1 )
python tools/rocprof_summary.py $P gpurun_out/${TAG}${b}_rocprof_${MODE}.md "batch $b, 12 steps, walks: ${MODE}"
python tools/gpu_idle.py $P | head -3 | tee gpurun_out/${TAG}${b}_gpu_idle_${MODE}.txt
find $P -name "*.db" -delete; rm -rf $P
done; done
unset NBSS_SIDE_STREAM
for b in 2 8; do for k in "" "NBSS_SIDE_STREAM=0" "NBSS_GRAPH=1" "NBSS_GRAPH=1 NBSS_SIDE_STREAM=0"; do
echo "batch $b [$k]: $(env $k timeout 120 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')")"
done; done | tee gpurun_out/${TAG}_rates.txt
