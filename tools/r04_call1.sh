#!/bin/bash
# round-4 call 1: parity of the single-sweep attention backward on the GPU + A/B against the two-sweep kernel (flavour v1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_bwd.py tests/test_side_stream.py tests/test_e2e_headline.py -m gpu -q -x -k "mhsa or side or headline" > gpurun_out/r04a_pytest_mhsa.log 2>&1; tail -n 4 gpurun_out/r04a_pytest_mhsa.log
bash tools/ab.sh "prod v1" "mhsa_bwd" 2>&1 | tee gpurun_out/r04a_ab.txt
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04a_prof -- python $GRAFT_REPO_ROOT/tools/run_one.py mhsa_bwd 32 5 > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/r04a_prof gpurun_out/r04a_mhsa_bwd_rocprof.md "run_one.py mhsa_bwd 32 5" ; head -20 gpurun_out/r04a_mhsa_bwd_rocprof.md
find gpurun_out/r04a_prof -name "*.db" -delete
for b in 2 8; do python bench.py --steps 8 --warmup 3 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')"; done | tee gpurun_out/r04a_small_batch.txt
