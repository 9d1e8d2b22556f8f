#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of a short trainable-encoder step.
# Usage: tools/gpu_profile_train.sh <tag>   -> gpurun_out/prof_<tag>_train_stats/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_train_stats -o stats -- \
    python $REPO/tools/encoder_train_bench.py --layers 2 --reps 1 > $OUT/prof_${TAG}_train_stats.log 2>&1
find $OUT/prof_${TAG}_train_stats -type f -size +8M -delete
f=$(find $OUT/prof_${TAG}_train_stats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
PY
