#!/bin/bash
# One diagnostic GPU-box call: in-kernel phase shares of the big kernels (needs `python -m nbss_amd.build phase` first) and the
# wait / issue / active split of every kernel's wave cycles (tools/pmc_stall.sh).  Output: gpurun_out/<tag>_phase_prof.txt, pmc_stall.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export NBSS_HIP_FLAVOUR=phase
( python tools/phase_prof.py tconvffn_bwd 32 224 tconvffn_bwd_s
  python tools/phase_prof.py mhsa_bwd 32 251
  python tools/phase_prof.py fconv_bwd 32 251
  python tools/phase_prof.py full_bwd 32 251
  python tools/phase_prof.py tconvffn_fwd 32 224 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02a_phase_prof.txt
unset NBSS_HIP_FLAVOUR
bash tools/pmc_stall.sh 32
