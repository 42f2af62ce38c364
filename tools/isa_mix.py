"""Static instruction mix of every kernel in a device .s file (from tools/isa_regs.py: /tmp/<stem>.s): python tools/isa_mix.py /tmp/tconvffn_s.s [filter]
Counts are static (loops count once); useful to compare variants of one kernel."""
import collections, re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "kernel"
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if flt not in name:
        continue
    body = body.split("s_endpgm")[0]
    c = collections.Counter()
    for ln in body.splitlines():
        m = re.match(r"\s+([a-z]\w+)", ln)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_pk_"): c["v_pk"] += 1
        elif op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): c["trans"] += 1
        elif op.startswith("v_cvt"): c["cvt"] += 1
        elif op.startswith(("v_perm", "v_and", "v_or", "v_lshl", "v_lshr", "v_bfi", "v_bfe", "v_alignb")): c["bitops"] += 1
        elif op.startswith("v_cndmask"): c["cndmask"] += 1
        elif op.startswith(("v_mov", "v_accvgpr")): c["mov"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        elif op.startswith("ds_"): c["ds"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
    print(name[:70], dict(sorted(c.items(), key=lambda kv: -kv[1])))
