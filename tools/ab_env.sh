#!/bin/bash
# A/B of environment knobs inside one GPU-box call:  tools/ab_env.sh "<batches>" <steps> "VAR=a VAR=b ..."   (each setting is one bench run per batch)
cd $GRAFT_REPO_ROOT
for b in ${1:-32}; do
  for kv in ${3:-X=0}; do
    env $kv python bench.py --steps ${2:-10} --warmup 3 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch', d['config']['batch_per_gpu'], '$kv', round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
