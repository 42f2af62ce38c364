#!/bin/bash
# Everything a round's profiles/ entry is made of, in one GPU-box call:  tools/round_artefacts.sh <tag>
#   gpurun_out/<tag>_pytest_gpu.log   full -m gpu suite
#   gpurun_out/<tag>_bench.json       default bench.py line (with cpu_baseline)
#   gpurun_out/<tag>_rocprof.md       rocprofv3 --kernel-trace --stats summary of a short bench run
#   gpurun_out/<tag>_batch_sweep.txt  utterances/s vs per-GPU batch
#   gpurun_out/pmc_traffic.json       FETCH_SIZE / WRITE_SIZE passes per sub-block kernel at the bench batch
TAG=${1:-r02}
BATCH=${2:-32}
# NBSS_COMMIT=<git rev-parse --short HEAD> is passed by the caller (.git does not travel to the GPU box) and stamped into the PMC files
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
for b in 2 4 8 16 31 32 48; do python bench.py --steps 5 --warmup 3 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')"; done | tee gpurun_out/${TAG}_batch_sweep.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_rocprof.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (batch 32; 7 steps + one-time table/pack kernels)"
python tools/gpu_idle.py gpurun_out/${TAG}_prof | tee gpurun_out/${TAG}_gpu_idle.txt
python tools/wgrad_breakdown.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_wgrad_breakdown.txt
# the rows around the hot path: on-device simulator throughput (f4) and the OnlineSpatialNet streaming step, native vs torch.nn (f2)
python tools/sim_throughput.py 32 12 2>/dev/null | tail -1 > gpurun_out/${TAG}_sim_throughput.json; cat gpurun_out/${TAG}_sim_throughput.json
python tools/online_throughput.py 16 32 2>/dev/null | tail -1 > gpurun_out/${TAG}_online_throughput.json; cut -c1-400 gpurun_out/${TAG}_online_throughput.json
find gpurun_out/${TAG}_prof -name "*.db" -delete
# SpatialNet-large train step (generic backward): kernel trace
bash tools/large_prof.sh 4 > /dev/null 2>&1; cp gpurun_out/large_rocprof.md gpurun_out/${TAG}_large_rocprof.md; head -12 gpurun_out/${TAG}_large_rocprof.md | tail -5
# PMC passes LAST: on this pool a bench run that follows rocprofv3 --pmc passes was measured 12 % slower (mhsa_fwd 2x), so the
# timed runs above must not come after them.  bench.py reads roofline.traffic from profiles/pmc_traffic.json (the previous
# collection); the fresh figure for the same kernel is patched into the saved line here.
bash tools/pmc_traffic.sh $BATCH > /dev/null 2>&1; python tools/pmc_traffic.py $BATCH | head -5; rm -rf gpurun_out/traffic
bash tools/pmc_mfma.sh $BATCH > /dev/null 2>&1; python tools/pmc_mfma.py $BATCH; rm -rf gpurun_out/mfma
python - <<PY
import json
b = json.loads(open("gpurun_out/${TAG}_bench.json").read())
t = json.load(open("gpurun_out/pmc_traffic.json"))
k = b["roofline"]["kernel"]
if t.get("batch") == b["config"]["batch_per_gpu"] and k in t["kernels"]:
    b["roofline"]["traffic"] = t["kernels"][k]["hbm_bytes"]
    b["roofline"]["traffic_commit"] = t.get("commit")
    b["roofline"]["traffic_stale"] = t.get("csrc_hash") != b["roofline"].get("csrc_hash")
    b["roofline"]["traffic_source"] = "tools/pmc_traffic.sh passes of the same round_artefacts.sh call"
m = json.load(open("gpurun_out/pmc_mfma.json"))
if k in m["kernels"]:
    b["roofline"]["mfma_util"] = m["kernels"][k]["mfma_busy_frac"]
open("gpurun_out/${TAG}_bench.json", "w").write(json.dumps(b) + "\n")
print("traffic", b["roofline"]["traffic"])
PY
