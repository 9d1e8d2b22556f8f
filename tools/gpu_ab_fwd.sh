#!/bin/bash
# encoder-forward A/B on one box (per-kernel HIP-event times): tools/gpu_ab_fwd.sh "<variant> ..." [pattern]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in ${1:-cur}; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/encoder_bench.py --reps 3 > $OUT/ab_fwd_$V.txt 2>&1
  echo "== encoder forward $V rc=$?"; grep -E "${2:-gb_mlp|VolTransformer}" $OUT/ab_fwd_$V.txt
done
