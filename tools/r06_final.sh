#!/bin/bash
# Round-6 collection (same steps as round 5) at the final kernel sources, one GPU-box call:  NBSS_COMMIT=<short hash> tools/r05_final.sh <tag>
#   default bench line (cpu_baseline, batch sweep and SpatialNet-large inside), kernel traces in order / two streams, batch sweep, the rows around
#   the path (simulator, online step, NBC2), the large step's trace, then — LAST (a bench after --pmc passes ran 12 % slower on this pool) — the PMC passes
#   of every sub-block kernel, patched into the saved bench line
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
bash tools/r05_trace.sh ${TAG} > /dev/null 2>&1; head -14 gpurun_out/${TAG}_rocprof_inorder.md | tail -6 | cut -c1-110
for b in 2 4 8 16 31 32 48; do timeout 120 python bench.py --steps 5 --warmup 3 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms/step')"; done | tee gpurun_out/${TAG}_batch_sweep.txt
if [ -z "$NBSS_FINAL_LITE" ]; then   # (NBSS_FINAL_LITE=1: the headline path only — the rows around it keep their earlier artefacts)
timeout 120 python tools/sim_throughput.py 32 12 2>/dev/null | tail -1 > gpurun_out/${TAG}_sim_throughput.json; cut -c1-200 gpurun_out/${TAG}_sim_throughput.json
timeout 120 python tools/online_throughput.py 16 32 2>/dev/null | tail -1 > gpurun_out/${TAG}_online_throughput.json; cut -c1-300 gpurun_out/${TAG}_online_throughput.json
timeout 200 python tools/nbc2_throughput.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_nbc2_throughput.json; cut -c1-300 gpurun_out/${TAG}_nbc2_throughput.json
timeout 200 bash tools/large_prof.sh 4 > /dev/null 2>&1; cp gpurun_out/large_rocprof.md gpurun_out/${TAG}_large_rocprof.md; head -12 gpurun_out/${TAG}_large_rocprof.md | tail -5 | cut -c1-110
fi
bash tools/pmc_traffic.sh 32 > /dev/null 2>&1; python tools/pmc_traffic.py 32 | grep -E "ratio|_bwd|_fwd" | head -12; rm -rf gpurun_out/traffic
bash tools/pmc_mfma.sh 32 > /dev/null 2>&1; python tools/pmc_mfma.py 32 | tail -12; rm -rf gpurun_out/mfma
python - <<PY
import json
b = json.loads(open("gpurun_out/${TAG}_bench.json").read())
t = json.load(open("gpurun_out/pmc_traffic.json"))
k = b["roofline"]["kernel"]
if t.get("batch") == b["config"]["batch_per_gpu"] and k in t["kernels"]:
    b["roofline"]["traffic"] = t["kernels"][k]["hbm_bytes"]
    b["roofline"]["traffic_commit"] = t.get("commit")
    b["roofline"]["traffic_stale"] = t.get("csrc_hash") != b["roofline"].get("csrc_hash")
    b["roofline"]["traffic_source"] = "tools/pmc_traffic.sh passes of the same r06_final.sh call"
m = json.load(open("gpurun_out/pmc_mfma.json"))
if k in m["kernels"]:
    b["roofline"]["mfma_util"] = m["kernels"][k]["mfma_busy_frac"]
open("gpurun_out/${TAG}_bench.json", "w").write(json.dumps(b) + "\n")
print("dominant", k, "frac", b["roofline"]["frac"], "traffic", b["roofline"]["traffic"], "stale", b["roofline"]["traffic_stale"])
PY
# where the waves of the big kernels spend their cycles (one SQ --pmc pass per sub-block), after everything that is timed
NBSS_PMC_KERNELS="mhsa_fwd tconvffn_fwd mhsa_bwd tconvffn_bwd fconv_bwd full_bwd" bash tools/pmc_stall.sh 32 > /dev/null 2>&1; cp gpurun_out/pmc_stall.txt gpurun_out/${TAG}_pmc_stall.txt; grep -E "tailw|bwd_q|bwd_h|full_bwd|fconv_bwd|fwd_s|mhsa_fwd" gpurun_out/${TAG}_pmc_stall.txt | cut -c1-150 | sort -u | head -12
