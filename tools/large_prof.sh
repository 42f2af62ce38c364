#!/bin/bash
# kernel-trace of the SpatialNet-large train step (generic backward, csrc/gbwd.hip): tools/large_prof.sh [batch]
cd $GRAFT_REPO_ROOT
cat > /tmp/large_step.py <<PY
import sys, torch, json
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
from nbss_amd._lib import hip
print(json.dumps(bench.large_train_rate(hip(), torch.device("cuda:0"), batch=${1:-4}, steps=3)))
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/large_prof && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/large_prof -- python /tmp/large_step.py 2>&1 | tail -1 )
python tools/rocprof_summary.py gpurun_out/large_prof gpurun_out/large_rocprof.md "SpatialNet-large train step, batch ${1:-4}, 4 steps"
find gpurun_out/large_prof -name "*.db" -delete
head -40 gpurun_out/large_rocprof.md
