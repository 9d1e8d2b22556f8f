#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + a PMC pass of a short bench run.
# Usage: tools/gpu_profile.sh <tag>      -> gpurun_out/prof_<tag>_{stats,pmc}/
set -u
TAG=${1:-r01}
EXTRA=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -o stats -- \
    python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/prof_${TAG}_stats.log 2>&1
# counters in their own pass (no stats / other trace domains)
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d $OUT/prof_${TAG}_pmc -o pmc -- \
    python $REPO/bench.py --steps 1 --warmup 0 --scenes 1 --no-cpu-baseline --no-roofline $EXTRA > $OUT/prof_${TAG}_pmc.log 2>&1
# keep only what is small enough to travel back
find $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_pmc -type f -size +8M -delete
find $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_pmc -type f | head -50
du -sh $OUT
