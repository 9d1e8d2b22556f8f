#!/bin/bash
# timing-only runs of tools/encoder_bench.py over library variants (results of the diagnostic builds are NOT valid):
#   tools/gpu_r05_diag.sh "<variant> ..."
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in ${1:-cur}; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 200 python tools/encoder_bench.py --layers 4 --reps 5 > $OUT/diag_${V}.txt 2>&1
  echo "== $V rc=$?"; grep -E "gb_conv3d|ga_gemm_kv" $OUT/diag_${V}.txt
done
