#!/bin/bash
# round 5, call 1: what round 4 left unverified on the device (native NBC), in-kernel phase shares of the big kernels at HEAD, baseline rates
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NBSS_RUN_UNVERIFIED=1 timeout 180 python -m pytest tests/test_nbc_native.py -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r05a_nbc_native.log
export NBSS_HIP_FLAVOUR=phase
( timeout 60 python tools/phase_prof.py tconvffn_bwd 32 224 tconvffn_bwd_v
  timeout 60 python tools/phase_prof.py tconvffn_fwd 32 224
  timeout 60 python tools/phase_prof.py mhsa_bwd 32 251
  timeout 60 python tools/phase_prof.py fconv_bwd 32 251
  timeout 60 python tools/phase_prof.py full_bwd 32 251 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05a_phase_prof.txt
unset NBSS_HIP_FLAVOUR
for K in tconvffn_bwd tconvffn_fwd mhsa_bwd mhsa_fwd fconv_bwd fconv_fwd full_bwd full_fwd; do timeout 60 python tools/run_one.py $K 32 5 2>/dev/null | tail -1; done | tee gpurun_out/r05a_run_one.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05a_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r05a_bench.json')); print(d['value'], d.get('utt_per_s_by_batch'), d['roofline'].get('frac_in_order')); print({k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if v>0.3})"
