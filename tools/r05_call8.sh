cd $GRAFT_REPO_ROOT
export NBSS_HIP_FLAVOUR=phase
( timeout 60 python tools/phase_prof.py tconvffn_bwd 32 224 tconvffn_bwd_v
  timeout 60 python tools/phase_prof.py fconv_bwd 32 251 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b_phase_prof.txt
unset NBSS_HIP_FLAVOUR
NBSS_PMC_KERNELS="tconvffn_bwd mhsa_bwd fconv_bwd full_bwd tconvffn_fwd mhsa_fwd" bash tools/pmc_stall.sh 32 2>&1 | grep -v "pack_kernel\|at::native" | tee gpurun_out/r05b_pmc_stall.txt
