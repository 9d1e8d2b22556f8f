#!/bin/bash
# Counters of the encoder-backward kernels (own PMC passes, no other trace domains); prints per-kernel means.
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/pmct_${TAG}_$i -o pmc -- \
      python $REPO/tools/encoder_train_bench.py --layers 1 --reps 1 > $OUT/pmct_${TAG}_$i.log 2>&1
done
find $OUT/pmct_${TAG}_* -type f -size +8M -delete
python - $OUT/pmct_${TAG}_ <<'PY'
import csv, glob, re, sys, collections
for d in sorted(glob.glob(sys.argv[1] + "*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            m = re.search(r"(gemm_bf16_tn_kernel<\w+>|gemm_bf16_tn_lds_kernel<\w+>|gemm_ring2?_kernel<\d, \d>|gemm_tn_ring_kernel<\w+>|ln_bwd_kernel|accum_partials_kernel)", r["Kernel_Name"])
            if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
