#!/bin/bash
# HBM traffic per launch of every sub-block kernel (GPU box).  Separate --pmc passes for FETCH_SIZE and WRITE_SIZE, no trace
# domains other than --kernel-trace.  Output: gpurun_out/pmc_traffic.json   (see tools/pmc_traffic.py for the gfx950 corrections)
B=${1:-8}
cd /tmp && export TMPDIR=/tmp
for K in ${NBSS_PMC_KERNELS:-fconv_fwd full_fwd mhsa_fwd tconvffn_fwd fconv_bwd full_bwd mhsa_bwd tconvffn_bwd}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic/${K}_$C -- python $GRAFT_REPO_ROOT/tools/run_one.py $K $B 2 > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT && python tools/pmc_traffic.py $B
