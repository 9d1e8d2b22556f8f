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
# (a) the headline command (the whole pipeline step, multi-view calls, 2 scene streams); ONLY_RASTER=1 skips (a) and (a'):
#     the pipeline step has no `--regime` (its Gaussians come out of the random-init network)
if [ "${ONLY_RASTER:-0}" != "1" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats_train -o stats -- \
    python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-legs --no-roofline $EXTRA > $OUT/prof_${TAG}_stats_train.log 2>&1
# (a') the pipeline step on ONE stream: every kernel of the step, torch's included, serialised
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats_pipe1 -o stats -- \
    python $REPO/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-side-legs --no-roofline $EXTRA > $OUT/prof_${TAG}_stats_pipe1.log 2>&1
fi
# (b) the raster kernels one at a time (one call per view, one stream, ONLY the timed steps: --no-roofline keeps the
#     bench's two-stream / two-lane side legs out of the averages): the durations the roofline object is built from
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -o stats -- \
    python $REPO/bench.py --steps 2 --warmup 1 --step raster --raster-api loop --streams 1 --no-fine --no-cpu-baseline --no-side-legs --no-roofline $EXTRA > $OUT/prof_${TAG}_stats.log 2>&1
# counters in their own passes (no stats / other trace domains); the raster alone, one call per view, one scene
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d $OUT/prof_${TAG}_pmc -o pmc -- \
    python $REPO/bench.py --steps 1 --warmup 0 --scenes 1 --step raster --raster-api loop --streams 1 --no-fine --no-cpu-baseline --no-roofline $EXTRA > $OUT/prof_${TAG}_pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $OUT/prof_${TAG}_pmc2 -o pmc -- \
    python $REPO/bench.py --steps 1 --warmup 0 --scenes 1 --step raster --raster-api loop --streams 1 --no-fine --no-cpu-baseline --no-roofline $EXTRA > $OUT/prof_${TAG}_pmc2.log 2>&1
# (c) the same counters over the MULTI-VIEW launches (one scene, 8 views, every kernel once over the cameras, one stream):
#     what the batched launch does to the forward's idle issue slots
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $OUT/prof_${TAG}_vpmc -o pmc -- \
    python $REPO/bench.py --steps 1 --warmup 0 --scenes 1 --step raster --raster-api views --streams 1 --no-fine --no-cpu-baseline --no-roofline $EXTRA > $OUT/prof_${TAG}_vpmc.log 2>&1
find $OUT/prof_${TAG}_vpmc -type f -size +8M -delete
# keep only what is small enough to travel back
find $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_stats_train $OUT/prof_${TAG}_stats_pipe1 $OUT/prof_${TAG}_pmc $OUT/prof_${TAG}_pmc2 -type f -size +8M -delete
find $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_pmc $OUT/prof_${TAG}_pmc2 -type f | head -50
du -sh $OUT
