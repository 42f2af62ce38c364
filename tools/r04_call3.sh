#!/bin/bash
cd $GRAFT_REPO_ROOT
for FL in prod stagex v1; do
  [ "$FL" = prod ] && unset NBSS_HIP_FLAVOUR || export NBSS_HIP_FLAVOUR=$FL
  echo "== $FL"; python tools/run_one.py mhsa_bwd 32 10 2>/dev/null | tail -1
done
unset NBSS_HIP_FLAVOUR
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'utt/s', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if v > 0.3})"
for K in tconvffn_bwd fconv_bwd full_bwd tconvffn_fwd fconv_fwd full_fwd; do
  echo "#### $K"
  bash tools/pmc_one.sh $K 32 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum" 2>&1 | grep -v "amdgpu.ids\|wgrad\|finalize\|reduce"
done
