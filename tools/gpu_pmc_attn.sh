#!/bin/bash
# Where the attention step's time goes: SQ wait / issue counters, VMEM instructions and L1 / L2 traffic of every kernel of
# tools/attn_bench.py (modes in $MODES, default 2 = the fused kernel).  Prints the per-kernel means.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/pmca_r04_$i -o pmc -- \
      python $REPO/tools/attn_bench.py --reps 2 --modes ${MODES:-2} > $OUT/pmca_r04_$i.log 2>&1
  echo "pass $i rc=$?"
done
find $OUT/pmca_r04_* -type f -size +8M -delete
python - $OUT/pmca_r04_ <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(group_attn_fused2_kernel|group_attn_kernel|gemm_ring2?_kernel<\d, \d>|gemm_tn_ring_kernel<\w+>|gemm_bf16_nt_kernel<\d, \d|ln_cast_kernel)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
