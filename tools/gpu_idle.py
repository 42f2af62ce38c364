"""GPU busy vs wall time over the steady-state part of a rocprofv3 kernel-trace db (launch gaps show up as idle)."""
import glob, sqlite3, sys
con = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
rows = con.execute("select start, end, name from kernels order by start").fetchall()
# steady state: last 60 % of the dispatches
rows = rows[int(len(rows) * 0.4):]
busy = sum(e - s for s, e, _ in rows)
wall = rows[-1][1] - rows[0][0]
gaps = sorted(((rows[i + 1][0] - rows[i][1]) / 1e3, rows[i][2].split("(")[0][-40:], rows[i + 1][2].split("(")[0][-40:]) for i in range(len(rows) - 1))
print(f"dispatches {len(rows)}  wall {wall/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {100*(1-busy/wall):.1f} %")
pos = [g for g in gaps if g[0] > 0]
print(f"gaps: n={len(pos)} mean {sum(g[0] for g in pos)/max(1,len(pos)):.2f} us; largest:")
for g in gaps[-8:]:
    print(f"  {g[0]:8.1f} us  after {g[1]}  before {g[2]}")
