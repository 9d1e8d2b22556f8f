#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in 1 2; do
for V in $1; do
  L=$REPO/lara_amd/liblara2dgs_$V.so
  LARA2DGS_LIB=$L timeout 300 python tools/encoder_train_bench.py --reps 10 2>&1 | grep trainable | cut -c1-130
  LARA2DGS_LIB=$L timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'])"
done
done
