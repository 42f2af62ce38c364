#!/bin/bash
# Round-3 starting diagnostics in ONE GPU-box call: short bench of the product build, in-kernel phase shares of every sub-block
# kernel (phase flavour) and the SQ stall counters of all eight sub-block kernels.   tools/diag_r03.sh <tag>
TAG=${1:-r03a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
for K in tconvffn_fwd fconv_bwd full_bwd mhsa_bwd; do NBSS_HIP_FLAVOUR=phase timeout 120 python tools/phase_prof.py $K 32 2>&1 | tail -40; done > gpurun_out/${TAG}_phase_prof.txt
NBSS_HIP_FLAVOUR=phase timeout 120 python tools/phase_prof.py tconvffn_bwd 32 224 tconvffn_bwd_s 2>&1 | tail -40 >> gpurun_out/${TAG}_phase_prof.txt
cat gpurun_out/${TAG}_phase_prof.txt
NBSS_PMC_KERNELS="tconvffn_bwd mhsa_bwd fconv_bwd full_bwd tconvffn_fwd mhsa_fwd fconv_fwd full_fwd" bash tools/pmc_stall.sh 32
cp gpurun_out/pmc_stall.txt gpurun_out/${TAG}_pmc_stall.txt
