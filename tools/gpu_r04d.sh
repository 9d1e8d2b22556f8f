#!/bin/bash
# round 4, second pass over composite_bwd: walk statistics (flag 512), A/B of the saved baseline library against the tree's, raster parity
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
EXTRA_FLAGS=512 timeout 300 python tools/bwd_probe.py > $OUT/r04d_bwd_probe.txt 2>&1; echo "probe rc=$?"; grep -E "walk:|phase time|view" $OUT/r04d_bwd_probe.txt
tools/gpu_ab.sh "${1:-base cur}" ${2:-test}
