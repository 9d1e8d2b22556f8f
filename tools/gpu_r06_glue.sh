#!/bin/bash
# round 6: one-stream kernel stats of the whole step (every kernel of the step, torch's included, serialised) + the glue probe.
# tools/gpu_r06_glue.sh <tag>
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_pipe1 -o stats -- \
    python $REPO/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-side-legs --no-roofline > $OUT/prof_${TAG}_pipe1.log 2>&1
echo "pipe1 rc=$?"
find $OUT/prof_${TAG}_pipe1 -type f -size +8M -delete
cp $(find $OUT/prof_${TAG}_pipe1 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_pipeline_one_stream.csv
cd $REPO
timeout 300 python tools/glue_probe.py --steps 2 --top 80 > $OUT/${TAG}_glue.txt 2>&1
head -14 $OUT/${TAG}_glue.txt
