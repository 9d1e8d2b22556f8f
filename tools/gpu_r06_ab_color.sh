#!/bin/bash
# whole-step A/B on one box: the fine pass's backward with None for its maps' gradient (colour-only composite_bwd) against
# seven planes of zeros (bench.py --dense-map-grads, the behaviour up to round 5): tools/gpu_r06_ab_color.sh [pairs]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for i in $(seq 1 ${1:-3}); do
  for F in "--dense-map-grads" ""; do
    timeout 400 python bench.py --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline --no-roofline $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ${F:-colour-only fine backward}:', d['value'], 'frames/s', d['ms_per_step'], 'ms')"
  done
done
