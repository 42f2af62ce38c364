"""group wgrad dispatches of a rocprofv3 kernel-trace db by kernel + launch geometry"""
import glob, sqlite3, sys, collections
con = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
rows = con.execute("select name, grid_x, grid_y, workgroup_x, lds_size, duration from kernels where name like '%wgrad%'").fetchall()
agg = collections.defaultdict(list)
for name, gx, gy, wx, lds, d in rows:
    agg[(name.split('(')[0].replace('void ', '')[:22], gx // wx, gy, lds)].append(d / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"wgrad total {tot/1e3:.2f} ms over {len(rows)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:22s} xbl={k[1]:5d} ybl={k[2]:3d} lds={k[3]:6d}  n={len(v):4d}  avg={sum(v)/len(v):8.1f} us  total={sum(v)/1e3:7.2f} ms")
