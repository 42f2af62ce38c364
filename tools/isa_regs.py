"""Register / scratch / LDS summary of every kernel in one csrc/*.hip (device-only -S build): python tools/isa_regs.py fconv [more ...]"""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -Wno-unused-variable -Wno-unused-but-set-variable -Wno-pass-failed -S --cuda-device-only".split()
for stem in [a for a in sys.argv[1:] if not a.startswith("-D")]:
    out = Path(tempfile.gettempdir()) / f"{stem}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *[a for a in sys.argv if a.startswith("-D")], "-o", str(out), str(ROOT / "nbss_amd" / "csrc" / f"{stem}.hip")], check=True, stderr=subprocess.DEVNULL)
    s = out.read_text()
    for b in s.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].replace("void ", "")[:64]
        g = lambda k: re.search(rf"\.{k}:\s+(\d+)", b).group(1)
        print(f"{dn:66s} vgpr={g('vgpr_count'):>3s} agpr={b.splitlines()[0].strip():>3s} sgpr={g('sgpr_count'):>3s} spill={g('vgpr_spill_count'):>3s} scratch={g('private_segment_fixed_size')}")
