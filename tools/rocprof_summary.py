"""Dump the per-kernel summary (name, calls, total/avg duration in us, %) of a rocprofv3 --kernel-trace --stats run
(rocpd sqlite output) as markdown, for committing under profiles/."""
import glob
import sqlite3
import sys


def main():
    root, out, title = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    dbs = glob.glob(f"{root}/**/*.db", recursive=True)
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\n{title}\n\nsource db: `{dbs[0]}` (durations in microseconds)\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            f.write(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |\n")
    print(out)


if __name__ == "__main__":
    main()
