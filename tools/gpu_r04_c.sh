#!/bin/bash
# Round 4, GPU call C: composite_bwd A/B (round-3 kernels vs this tree vs this tree without the all-inactive shortcut), the
# regroup probe with the static per-segment deal, the whole GPU suite (plain, full logs of failures) and under poison mode.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for V in r03 cur noskip; do
  L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
  LARA2DGS_LIB=$L timeout 300 python tools/kbench.py --reps 5 > $OUT/r04_kbench_$V.txt 2>&1
  echo "== kbench $V rc=$?"; grep -E "^\[|composite|preprocess" $OUT/r04_kbench_$V.txt
done
timeout 600 python tools/regroup_probe.py > $OUT/r04_regroup_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/r04_regroup_probe.txt | grep "^\["
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s > $OUT/r04_suite.log 2>&1
echo "suite rc=$?"; tail -6 $OUT/r04_suite.log; grep -n "unexplained_detail\|two-stream vs one-stream" $OUT/r04_suite.log | cut -c1-900 | head
LARA2DGS_POISON_BUFFERS=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short > $OUT/r04_poison_suite.log 2>&1
echo "poison suite rc=$?"; tail -5 $OUT/r04_poison_suite.log
