#!/bin/bash
# round 6, GPU call 2: upper bound of a fold consolidation (folds knocked out), the two-pass / four-wave attention forward probe, side stream on / off at the batches of the sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== mhsa_fwd: product vs two passes of a head pair, 4 waves, two workgroups per CU (probe: weights from global, 32 spilled registers)"
NBSS_HIP_FLAVOUR=mh2p timeout 300 python -m pytest tests/test_kernels_fwd.py -m gpu -q -x -k "mhsa" 2>&1 | tail -2
bash tools/kab.sh "prod mh2p" "mhsa_fwd" 32 100
echo "== folds knocked out (timing only) vs product; side stream off"
bash tools/ab_env.sh "2 8 32" 10 "NBSS_X=prod NBSS_HIP_FLAVOUR=foldko NBSS_SIDE_STREAM=0"
