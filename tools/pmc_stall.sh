#!/bin/bash
# Where do the waves of a kernel spend their time?  One SQ --pmc pass per kernel (GPU box):
#   SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked in s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY  (quad-cycles)
# Output: gpurun_out/pmc_stall.txt
B=${1:-32}
cd /tmp && export TMPDIR=/tmp
for K in ${NBSS_PMC_KERNELS:-tconvffn_bwd mhsa_bwd fconv_bwd full_bwd}; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stall/${K} -- python $GRAFT_REPO_ROOT/tools/run_one.py $K $B 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT && python - <<'PY' | tee gpurun_out/pmc_stall.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/stall/*")):
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]] = float(r["Counter_Value"])  # last dispatch wins
    for k, v in per.items():
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc < 1e6: continue
        print(f"{k:50s} " + " ".join(f"{n[3:]}={v.get(n, 0) / wc:.3f}" for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")) + f"  wave_cycles={wc:.3g}")
PY
rm -rf gpurun_out/stall
