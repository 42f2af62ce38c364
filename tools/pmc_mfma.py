"""Fold the passes of tools/pmc_mfma.sh into MFMA utilisation per launch of each sub-block kernel.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs): the share of SIMD-cycles of the kernel's wall time
in which a matrix pipe was busy (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs by rocprofv3's
aggregation; GRBM_GUI_ACTIVE is summed over the 8 XCDs and is divided by 8 here).  valu_active_frac: the same for SQ_ACTIVE_INST_VALU
(quad-cycles -> x4)."""
import csv, glob, hashlib, json, os, sys
from pathlib import Path


def csrc_hash():  # same as bench.py: which kernel sources these counters belong to
    h = hashlib.sha1()
    for f in sorted((Path(__file__).resolve().parent.parent / "nbss_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:12]


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = {"batch": B, "commit": os.environ.get("NBSS_COMMIT"), "csrc_hash": csrc_hash(), "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "kernels": {}}
MEMBERS = {"tconvffn_bwd": ["tconvffn_bwd", "tconvffn_du", "tailw_kernel"], "mhsa_bwd": ["mhsa_bwd", "tailw_kernel"], "mhsa_fwd": ["mhsa_fwd", "mhsa_kv", "mhsa_flash"]}
for k in ["fconv_fwd", "full_fwd", "mhsa_fwd", "tconvffn_fwd", "fconv_bwd", "full_bwd", "mhsa_bwd", "tconvffn_bwd"]:
    vals, per = {}, {}
    for tag in ("SQ", "GRBM"):
        for f in glob.glob(f"gpurun_out/mfma/{k}_{tag}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if any(m in r["Kernel_Name"] for m in MEMBERS.get(k, [k])) and "wgrad" not in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
                    per[(r["Counter_Name"], r["Kernel_Name"])] = float(r["Counter_Value"])  # last launch of each member kernel wins
    for (cn, _), v in per.items():
        vals[cn] = vals.get(cn, 0.0) + v
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in vals or "GRBM_GUI_ACTIVE" not in vals:
        continue
    simd_cycles = vals["GRBM_GUI_ACTIVE"] / 8 * 1024
    vals["mfma_busy_frac"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    if "SQ_ACTIVE_INST_VALU" in vals:
        vals["valu_active_frac"] = 4 * vals["SQ_ACTIVE_INST_VALU"] / simd_cycles
    out["kernels"][k] = vals
json.dump(out, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
print(json.dumps({k: {"mfma_busy_frac": round(v["mfma_busy_frac"], 4), "valu_active_frac": round(v.get("valu_active_frac", 0), 4)} for k, v in out["kernels"].items()}, indent=1))
