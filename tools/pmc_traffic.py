"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into HBM bytes per launch.
MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
read stream (128-B requests tallied as 64 B), so it is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import csv, glob, hashlib, json, os, sys
from pathlib import Path


def csrc_hash():  # same as bench.py: which kernel sources these counters belong to
    h = hashlib.sha1()
    for f in sorted((Path(__file__).resolve().parent.parent / "nbss_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:12]


B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = 129 * 251 * 96 * 2 * B
out = {"batch": B, "commit": os.environ.get("NBSS_COMMIT"), "csrc_hash": csrc_hash(), "units": "bytes per launch", "correction": "2 * FETCH_SIZE KiB (gfx950 half-count) + WRITE_SIZE KiB", "kernels": {}}
MEMBERS = {"tconvffn_bwd": ["tconvffn_bwd", "tconvffn_du", "tailw_kernel"], "mhsa_bwd": ["mhsa_bwd", "tailw_kernel"], "mhsa_fwd": ["mhsa_fwd", "mhsa_kv", "mhsa_flash"]}
KERNELS = os.environ.get("NBSS_PMC_KERNELS", "fconv_fwd full_fwd mhsa_fwd tconvffn_fwd fconv_bwd full_bwd mhsa_bwd tconvffn_bwd").split()
for k in ["fconv_fwd", "full_fwd", "mhsa_fwd", "tconvffn_fwd", "fconv_bwd", "full_bwd", "mhsa_bwd", "tconvffn_bwd"]:
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = {}  # kernel name -> counter of its last dispatch; a sub-block that is several kernels (tconvffn_bwd: K1 + du) is their sum
        for f in glob.glob(f"gpurun_out/traffic/{k}_{c}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and any(m in r["Kernel_Name"] for m in MEMBERS.get(k, [k])) and "wgrad" not in r["Kernel_Name"]:
                    v[r["Kernel_Name"]] = float(r["Counter_Value"])
        vals[c] = sum(v.values()) if v else None
    if vals["FETCH_SIZE"] is None or vals["WRITE_SIZE"] is None:
        continue
    hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    alg = (2 if k.endswith("fwd") else 3) * S
    out["kernels"][k] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "hbm_bytes": hbm, "algorithmic_bytes": alg,
                         "ratio": hbm / alg}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
