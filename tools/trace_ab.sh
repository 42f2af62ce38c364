#!/bin/bash
# in-order kernel trace of the bench per flavour, one GPU-box call: tools/trace_ab.sh "<flavours>" "<kernel name patterns (egrep)>"
cd $GRAFT_REPO_ROOT
export NBSS_SIDE_STREAM=0
for FL in ${1:-prod}; do
  [ "$FL" = prod ] && unset NBSS_HIP_FLAVOUR || export NBSS_HIP_FLAVOUR=$FL
  D=/tmp/tab_$FL; rm -rf $D
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
  python tools/rocprof_summary.py $D /tmp/tab_$FL.md "$FL" > /dev/null
  echo "== $FL"; grep -E "${2:-fwd|bwd|tailw}" /tmp/tab_$FL.md | awk -F'|' '{printf "%-60s %8s\n", $2, $5}' | head -16
  rm -rf $D
done
