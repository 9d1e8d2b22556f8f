#!/bin/bash
# Round 4: the two committed bench lines (profiles/<tag>_bench_{init,trained}.json).
set -u
TAG=${1:-r04a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python bench.py --steps 20 --warmup 3 > $OUT/bench_${TAG}_init.json 2> $OUT/bench_${TAG}_init.err
echo "bench init rc=$?"; cut -c1-260 $OUT/bench_${TAG}_init.json; tail -2 $OUT/bench_${TAG}_init.err
timeout 900 python bench.py --step train --regime trained --steps 20 --warmup 3 > $OUT/bench_${TAG}_trained.json 2> $OUT/bench_${TAG}_trained.err
echo "bench trained rc=$?"; cut -c1-260 $OUT/bench_${TAG}_trained.json
