#!/bin/bash
# MFMA utilisation per launch of every sub-block kernel (GPU box): SQ counters in one --pmc pass, GRBM_GUI_ACTIVE in another
# (separate passes, only --kernel-trace beside them).  Output: gpurun_out/pmc_mfma.json (tools/pmc_mfma.py).
B=${1:-32}
cd /tmp && export TMPDIR=/tmp
for K in ${NBSS_PMC_KERNELS:-fconv_fwd full_fwd mhsa_fwd tconvffn_fwd fconv_bwd full_bwd mhsa_bwd tconvffn_bwd}; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mfma/${K}_SQ -- python $GRAFT_REPO_ROOT/tools/run_one.py $K $B 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mfma/${K}_GRBM -- python $GRAFT_REPO_ROOT/tools/run_one.py $K $B 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT && python tools/pmc_mfma.py $B
