#!/bin/bash
# round 6, GPU call 1: the changed parity tests on the device, this box's per-kernel baseline, the tail-kernel knock-outs, a first bench A/B vs the round-5 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_nb_native_vs_reference.py tests/test_gpu_simulation.py tests/test_kernels_bwd.py -m gpu -q -x -k "reference or simulation or full" 2>&1 | tail -3
echo "== per-kernel baseline (in order, batch 32)"
bash tools/kab.sh "prod" "fconv_fwd full_fwd mhsa_fwd tconvffn_fwd tconvffn_fwd_infer fconv_bwd full_bwd" 32 10
echo "== tail kernels: role split and knock-outs"
bash tools/kab.sh "prod tw4 twko1 twko2 twko3" "mhsa_bwd tconvffn_bwd" 32 10
echo "== mhsa_fwd 16 waves x 1 strip"
bash tools/kab.sh "mh1" "mhsa_fwd" 32 100
echo "== bench: round-5 library vs working tree, streams"
bash tools/ab_env.sh "32" 8 "NBSS_HIP_FLAVOUR=prev NBSS_X=cur NBSS_SIDE_STREAM=0 NBSS_HIP_FLAVOUR=tw4"
bash tools/ab_env.sh "2 8" 10 "NBSS_HIP_FLAVOUR=prev NBSS_X=cur"
