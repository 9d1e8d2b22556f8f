#!/bin/bash
# kernel-trace stats of the trainable encoder step (tools/encoder_train_bench.py): tools/gpu_r05_enc_stats.sh <tag> [variant]
set -u
TAG=${1:-r05b}; V=${2:-cur}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
L=$REPO/lara_amd/liblara2dgs_$V.so; [ $V = cur ] && L=$REPO/lara_amd/liblara2dgs.so
cd /tmp && export TMPDIR=/tmp
LARA2DGS_LIB=$L timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_enct -o stats -- \
    python $REPO/tools/encoder_train_bench.py --reps 2 > $OUT/prof_${TAG}_enct.log 2>&1
echo "rc=$?"
find $OUT/prof_${TAG}_enct -type f -size +8M -delete
F=$(find $OUT/prof_${TAG}_enct -name "*kernel_stats.csv" | head -1)
cp $F $OUT/${TAG}_encoder_train_kernel_stats.csv
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.2f} %")
PY
