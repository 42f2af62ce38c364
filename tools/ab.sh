#!/bin/bash
# quick A/B on the GPU box: tools/ab.sh "<flavours>" "<kernels>"   ('' flavour = the product build)
# prints the per-call time of each listed sub-block kernel and a short bench line (utt/s + per-kernel ms/step) per flavour
cd $GRAFT_REPO_ROOT
for FL in ${1:-prod}; do
  [ "$FL" = prod ] && unset NBSS_HIP_FLAVOUR || export NBSS_HIP_FLAVOUR=$FL
  echo "== flavour $FL"
  for K in $2; do python tools/run_one.py $K 32 5 2>/dev/null | tail -1; done
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'utt/s', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if v > 0.5})"
done
