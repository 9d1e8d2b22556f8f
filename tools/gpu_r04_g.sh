#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 300 python tools/attn_bench.py --reps 20 > $OUT/r04_attn_bench.txt 2>&1; echo "attn rc=$?"; grep -v "^ga_\|Warning" $OUT/r04_attn_bench.txt; grep "ga_fused" $OUT/r04_attn_bench.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/r04_tests_g.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/r04_tests_g.log
