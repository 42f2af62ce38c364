"""bench.py's multi-process launch path without a GPU: `python bench.py --gpus 2 --dry-run` re-executes itself under torch.distributed.run
(one rank per would-be GPU, rendezvous on 127.0.0.1), runs the bucketed gradient all-reduce skeleton over gloo and prints the contract line
with the world size every rank saw — so that the driver's N > 1 scaling run needs no edits to bench.py."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_bench_self_launch_dry_run_world2():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world"] == [2, 2] and d["dry_run"] and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["grad_elements"] == 1191092


def test_roofline_entry_names_the_larger_bound():
    sys.path.insert(0, str(ROOT))
    import bench
    e = bench.roofline_entry("mhsa_bwd", 32, 1.5 * 4, 4, 0.2)  # 1.5 ms per launch
    assert e["bound"] == "mfma" and e["unit"] == "TFLOP/s" and abs(e["frac_mfma"] - 2 * 5.507e9 * 32 / 2.5e15 / 1.5e-3) < 1e-9
    assert e["frac"] == e["frac_mfma"] > e["frac_hbm"] and abs(e["frac_hbm"] - 3 * bench.S_BYTES_BF16 * 32 / 8e12 / 1.5e-3) < 1e-9
    f = bench.roofline_entry("fconv_bwd", 32, 0.43 * 2, 2, 0.1)
    assert f["bound"] == "hbm" and f["unit"] == "GB/s" and f["frac"] == f["frac_hbm"]
