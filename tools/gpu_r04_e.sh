#!/bin/bash
# Round 4, GPU call E: composite_bwd slot-pool size A/B (384 slots / 3 workgroups per CU vs 640 / 2).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in cur prio cur prio; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/kbench.py --reps 5 > $OUT/r04_kbench_$V.txt 2>&1
  echo "== kbench $V rc=$?"; grep -E "^\[|composite" $OUT/r04_kbench_$V.txt
done
