#!/bin/bash
# A/B with per-kernel-group lines of the encoder benches: tools/gpu_r05_ab2.sh "<variant> ..." [test]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
for V in ${1:-cur}; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/encoder_bench.py --layers 4 --reps 5 > $OUT/ab6_encf_${V}_$rep.txt 2>&1
  echo "== encoder fwd $V #$rep rc=$?"; grep -E " us x |VolTransformer" $OUT/ab6_encf_${V}_$rep.txt
  LARA2DGS_LIB=$L timeout 300 python tools/encoder_train_bench.py --reps 3 > $OUT/ab6_enct_${V}_$rep.txt 2>&1
  echo "== encoder train $V #$rep rc=$?"; grep -E " us x |trainable" $OUT/ab6_enct_${V}_$rep.txt
done
done
if [ "${2:-}" = "test" ]; then
  for V in ${1:-cur}; do
    [ $V = cur ] && continue
    LARA2DGS_LIB=$REPO/lara_amd/liblara2dgs_$V.so timeout 900 python -m pytest tests/test_voltrans.py tests/test_voltrans_train.py tests/test_groupatt.py -m gpu -q --tb=short 2>&1 | tail -4
  done
fi
