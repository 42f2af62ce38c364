#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NBSS_HIP_FLAVOUR=phase timeout 120 python tools/phase_prof.py mhsa_bwd 32 251 2>&1 | tail -12 | tee gpurun_out/r04b_phase_mhsa_bwd.txt
for FL in prod dyearly; do
  [ "$FL" = prod ] && unset NBSS_HIP_FLAVOUR || export NBSS_HIP_FLAVOUR=$FL
  echo "== $FL"; python tools/run_one.py mhsa_bwd 32 10 2>/dev/null | tail -1
done
unset NBSS_HIP_FLAVOUR
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r04b_prof --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_one.py mhsa_bwd 32 3 > /dev/null 2>&1 )
f=$(find gpurun_out/r04b_prof -name "*kernel_trace.csv" | head -1); head -1 $f; grep mhsa_bwd_h $f | head -2
rm -rf gpurun_out/r04b_prof
