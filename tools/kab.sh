#!/bin/bash
# per-KERNEL A/B inside one GPU-box call (durations from rocprofv3 --kernel-trace, walks in order so that nothing co-runs):
#   tools/kab.sh "<flavours>" "<run_one.py names>" [batch] [min us]        ('prod' = the product build; flavours from `python -m nbss_amd.build flavour <name> -D...`)
# prints, per flavour and sub-block, every kernel of at least [min us] (default 15) with its average duration
cd $GRAFT_REPO_ROOT
B=${3:-32}
MIN=${4:-15}
export NBSS_SIDE_STREAM=0
for FL in ${1:-prod}; do
  [ "$FL" = prod ] && unset NBSS_HIP_FLAVOUR || export NBSS_HIP_FLAVOUR=$FL
  for K in $2; do
    D=/tmp/kab_${FL}_${K}
    rm -rf $D
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -- python $GRAFT_REPO_ROOT/tools/run_one.py $K $B ${KAB_ITERS:-8} > /dev/null 2>&1 )
    python - "$D" "$FL" "$K" "$MIN" <<'PY'
import glob, sqlite3, sys
root, fl, k, mn = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
dbs = glob.glob(f"{root}/**/*.db", recursive=True)
if not dbs:
    print(f"{fl:10s} {k:14s} NO TRACE"); sys.exit(0)
rows = sqlite3.connect(dbs[0]).execute("select name, total_calls, average from top_kernels").fetchall()
print(f"{fl:10s} {k:14s} " + "  ".join(f"{n.split('(')[0].replace('void ', '')[:34]}={a:.1f}" for n, c, a in rows if a >= mn and "elementwise" not in n and "pack_kernel" not in n))
PY
    rm -rf $D
  done
done
