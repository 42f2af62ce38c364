#!/bin/bash
# the whole -m gpu suite + smoke on the box     tools/r05_suite.sh <tag>
# (serial on purpose: with pytest-xdist — 4 workers, one file each — the CPU sides of the tests, each sized for all host threads, got in each other's
#  way: 39 tests in 600 s against 164 in 535 s)
TAG=${1:-r05s}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
