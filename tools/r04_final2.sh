#!/bin/bash
# round 4, after the last call: the tile weight gradient (wgrad_g.hip) as the default — every large-geometry test and the A/B test on the device, the rate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "commit ${NBSS_COMMIT:-?} (default: tile weight gradient on)" > gpurun_out/r04j_pytest_gpu_large.log
timeout 230 python -m pytest tests/test_large.py tests/test_wgrad_g.py -m gpu -q >> gpurun_out/r04j_pytest_gpu_large.log 2>&1
tail -3 gpurun_out/r04j_pytest_gpu_large.log
timeout 40 python tools/large_rate.py 4 3 2>&1 | tail -1 | tee gpurun_out/r04j_large.json
