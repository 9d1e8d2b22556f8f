#!/bin/bash
# Round 5, after the ring kernels of the encoder: bench lines (both regimes), encoder counters (forward / backward), kernel stats of the
# trainable encoder step and of the whole step on one stream.  tools/gpu_r05c.sh [tag]
set -u
TAG=${1:-r05c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_init.json 2> $OUT/${TAG}_bench_init.err; echo "bench init rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --regime trained > $OUT/${TAG}_bench_trained.json 2> $OUT/${TAG}_bench_trained.err; echo "bench trained rc=$?"
bash tools/gpu_pmc_enc.sh $TAG > $OUT/${TAG}_pmc_enc.log 2>&1; echo "pmc enc rc=$?"
bash tools/gpu_pmc_train.sh $TAG > $OUT/${TAG}_encoder_train_pmc.txt 2>&1; echo "pmc train rc=$?"
bash tools/gpu_r05_enc_stats.sh $TAG cur > $OUT/${TAG}_enc_stats.log 2>&1; echo "enc stats rc=$?"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_pipe1 -o stats -- \
      python $REPO/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-side-legs --no-roofline > $OUT/prof_${TAG}_pipe1.log 2>&1
  echo "pipe1 rc=$?"
  find $OUT/prof_${TAG}_pipe1 -type f -size +8M -delete
  cp $(find $OUT/prof_${TAG}_pipe1 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_pipeline_one_stream.csv )
du -sh $OUT
