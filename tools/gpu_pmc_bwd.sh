#!/bin/bash
# Instruction-mix / stall counters of the composite kernels (own PMC pass, no other trace domains).
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/pmcb_${TAG}_$i -o pmc -- \
      python $REPO/bench.py --steps 1 --warmup 0 --scenes 1 --no-cpu-baseline --no-roofline > $OUT/pmcb_${TAG}_$i.log 2>&1
done
find $OUT/pmcb_${TAG}_* -type f -size +8M -delete
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/pmcb_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVES","SQ_ACTIVE_INST_VALU","SQ_LDS_BANK_CONFLICT"): cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "composite" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        n = max(cnt[(k,"SQ_WAVES")], 1)
        print(f"   {c:24s} {v/n:14.0f} per launch")
PY
