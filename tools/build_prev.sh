#!/bin/bash
# Build the library of a git revision as the flavour "prev" (nbss_amd/lib/libnbss_hip_prev.so) for a same-box A/B against the working tree:
#   tools/build_prev.sh [rev]        (default HEAD)        then e.g.  gpurun -- 'bash tools/ab_env.sh "2 8 32" 10 "NBSS_HIP_FLAVOUR=prev NBSS_X=cur"'
# The revision is checked out as a worktree under gpurun_out/ (git-ignored, not shipped) and removed afterwards.
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
WT=gpurun_out/wt_prev
git worktree remove --force $WT 2>/dev/null || true
git worktree add -f $WT $REV -q
( cd $WT && python -c "from nbss_amd import build; build.build_hip(verbose=True, flavour='prev')" )
cp $WT/nbss_amd/lib/libnbss_hip_prev.so nbss_amd/lib/
git worktree remove --force $WT
ls -la nbss_amd/lib/libnbss_hip_prev.so
