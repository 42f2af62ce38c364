#!/bin/bash
# build the product library (and the emulator) if stale, then run one GPU-box call:  tools/gp.sh <timeout s> '<command>'
cd "$(dirname "$0")/.."
python -m nbss_amd.build all > /tmp/gp_build.log 2>&1 || { tail -30 /tmp/gp_build.log; exit 1; }
gpurun --timeout ${1:-900} -- "$2"
