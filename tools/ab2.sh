#!/bin/bash
# A/B inside one GPU-box call where the two builds differ in the ops signatures: "base" = the previous round's library run from its own tree
# is not possible (C ABI changed) -> compare per-step bench lines of the flavour builds of THIS tree:  tools/ab2.sh "<flavours>"
cd $GRAFT_REPO_ROOT
for FL in ${1:-prod}; do
  [ "$FL" = prod ] && unset NBSS_HIP_FLAVOUR || export NBSS_HIP_FLAVOUR=$FL
  echo "== flavour $FL"
  for K in $2; do python tools/run_one.py $K 32 5 2>/dev/null | tail -1; done
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'utt/s', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if v > 0.5})"
done
