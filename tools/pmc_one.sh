#!/bin/bash
# PMC counters of ONE sub-block kernel (GPU box): tools/pmc_one.sh <run_one kernel> <batch> "<counter set 1>" "<counter set 2>" ...   (one --pmc pass per set)
K=$1; B=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc1_$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$i -- python $GRAFT_REPO_ROOT/tools/run_one.py $K $B 2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc1_$i/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] = float(r["Counter_Value"])   # last dispatch
for k, v in acc.items():
    if "at::" in k or "pack" in k: continue
    print(k, {a: (round(b / 1e6, 3)) for a, b in v.items()}, "(x1e6)")
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc1_$i
  i=$((i+1))
done
