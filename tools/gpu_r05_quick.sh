#!/bin/bash
# round 5: one-stream kernel stats of the step with the MS-SSIM term + the glue probe.  tools/gpu_r05_quick.sh <tag>
set -u
TAG=${1:-r05q}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_pipe1ms -o stats -- \
    python $REPO/bench.py --steps 2 --warmup 1 --streams 1 --ms-ssim --no-cpu-baseline --no-side-legs --no-roofline > $OUT/prof_${TAG}_pipe1ms.log 2>&1
find $OUT/prof_${TAG}_pipe1ms -type f -size +8M -delete
cd $REPO
timeout 300 python tools/glue_probe.py --steps 2 --top 60 > $OUT/${TAG}_glue.txt 2>&1
head -12 $OUT/${TAG}_glue.txt | tail -8
