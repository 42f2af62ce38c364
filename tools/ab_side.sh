#!/bin/bash
# A/B of the walks' second stream (csrc/side.h) inside one GPU-box call: NBSS_SIDE_STREAM=0 (in order) vs default, at the batches given
cd $GRAFT_REPO_ROOT
for b in ${1:-2 8 32}; do
  for side in 0 1; do
    NBSS_SIDE_STREAM=$side python bench.py --steps ${2:-10} --warmup 3 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch', d['config']['batch_per_gpu'], 'side', $side, round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
