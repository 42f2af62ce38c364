#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_graph_step.py tests/test_batch_invariance.py tests/test_online.py tests/test_side_stream.py -m gpu -q -x 2>&1 | tail -15
for b in 2 8 32; do python bench.py --steps 8 --warmup 4 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')"; done
