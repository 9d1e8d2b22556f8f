#!/bin/bash
# kernel A/B on one box: tools/gpu_ab.sh "<variant> <variant> ..."   (variant `cur` = lara_amd/liblara2dgs.so, else liblara2dgs_<variant>.so)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in ${1:-cur}; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/kbench.py --reps 5 > $OUT/ab_kbench_$V.txt 2>&1
  echo "== kbench $V rc=$?"; grep -E "^\[|composite" $OUT/ab_kbench_$V.txt
  if [ "${3:-}" = "color" ]; then
    LARA2DGS_LIB=$L timeout 300 python tools/kbench.py --reps 5 --color-only > $OUT/ab_kbench_${V}_color.txt 2>&1
    echo "== kbench $V, no gradient on the maps rc=$?"; grep -E "composite_bwd|preprocess_bwd" $OUT/ab_kbench_${V}_color.txt
  fi
done
[ "${2:-}" = "test" ] && timeout 600 python -m pytest tests/test_raster_parity_gpu.py tests/test_views_gpu.py -m gpu -q --tb=short 2>&1 | tail -3
