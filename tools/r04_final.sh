#!/bin/bash
# round 4, last call: the -m gpu suite at the final commit, the default bench line, the large train step's kernel trace, the opt-in tile weight gradient A/B
# (most important first: the call is cut off when the round's GPU budget ends)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "commit ${NBSS_COMMIT:-?}" > gpurun_out/r04i_pytest_gpu.log
timeout 660 python -m pytest tests -m gpu -q >> gpurun_out/r04i_pytest_gpu.log 2>&1
tail -4 gpurun_out/r04i_pytest_gpu.log
timeout 300 python bench.py 2> gpurun_out/r04i_bench.err | tail -1 > gpurun_out/r04i_bench.json
cut -c1-600 gpurun_out/r04i_bench.json
python -c "import json; d=json.load(open('gpurun_out/r04i_bench.json')); print('large', d.get('utt_per_s_large')); print('sweep', d.get('utt_per_s_by_batch')); print('in-order', d['roofline'].get('utt_per_s_in_order'), d['roofline'].get('frac_in_order'), d['roofline'].get('frac'))"
timeout 120 bash tools/large_prof.sh 4 2>&1 | tail -3
cp gpurun_out/large_rocprof.md gpurun_out/r04i_large_rocprof.md
NBSS_WGRAD_TILE=1 timeout 60 python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04i_large_wgrad_tile.json
NBSS_WGRAD_TILE=1 timeout 60 python tools/large_rate.py 8 3 2>&1 | tail -1 | tee gpurun_out/r04i_large_wgrad_tile_b8.json
