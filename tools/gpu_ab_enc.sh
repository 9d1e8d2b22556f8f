#!/bin/bash
# encoder-training A/B on one box: tools/gpu_ab_enc.sh "<variant> ..."  (see gpu_ab.sh)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in ${1:-cur}; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/encoder_train_bench.py --reps 3 > $OUT/ab_enc_$V.txt 2>&1
  echo "== encoder_train $V rc=$?"; grep -E "gbb_dw|trainable" $OUT/ab_enc_$V.txt
done
[ "${2:-}" = "test" ] && timeout 900 python -m pytest tests/test_voltrans_train.py -m gpu -q --tb=short 2>&1 | tail -3
