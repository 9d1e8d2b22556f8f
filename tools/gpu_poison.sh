#!/bin/bash
# The whole GPU suite and 200 headline steps under LARA2DGS_POISON_BUFFERS=1 (every state / scratch buffer of the rasteriser
# 0xFF-filled between guard zones; the guards are checked after every test / step).  -> gpurun_out/<tag>_poison_*
set -u
TAG=${1:-r04a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
LARA2DGS_POISON_BUFFERS=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short -rs > $OUT/${TAG}_poison_suite.log 2>&1
echo "poison suite rc=$?"; tail -6 $OUT/${TAG}_poison_suite.log
LARA2DGS_POISON_BUFFERS=1 timeout 900 python bench.py --steps 200 --warmup 2 --no-side-legs --no-cpu-baseline --no-roofline > $OUT/${TAG}_poison_bench200.json 2> $OUT/${TAG}_poison_bench200.err
echo "poison bench rc=$?"; cut -c1-220 $OUT/${TAG}_poison_bench200.json; tail -2 $OUT/${TAG}_poison_bench200.err
