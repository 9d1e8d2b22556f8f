#!/bin/bash
# whole-step + encoder A/B of library variants on one box: tools/gpu_r06_ab_step.sh "<variant> ..."
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in ${1:-cur}; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/encoder_train_bench.py --reps 3 > $OUT/ab_enc_$V.txt 2>&1
  echo "== $V: $(grep trainable $OUT/ab_enc_$V.txt | cut -c1-140)"
  LARA2DGS_LIB=$L timeout 400 python bench.py --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', d['value'], d['ms_per_step'])"
done
