#!/bin/bash
# Round 4, GPU call D: the attention step's three modes, the changed tests, a bench line with the new side objects.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 300 python tools/attn_bench.py --reps 20 > $OUT/r04_attn_bench.txt 2>&1; echo "attn rc=$?"; cat $OUT/r04_attn_bench.txt
timeout 900 python -m pytest tests/test_groupatt.py tests/test_full_size_gpu.py tests/test_pipeline.py tests/test_ddp_encoder_gpu.py tests/test_bench_launch_gpu.py -m gpu -q --tb=short -s > $OUT/r04_tests_d.log 2>&1
echo "tests rc=$?"; tail -8 $OUT/r04_tests_d.log; grep -n "two-stream vs one-stream" $OUT/r04_tests_d.log | cut -c1-400
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/r04_bench_d.json 2> $OUT/r04_bench_d.err
echo "bench rc=$?"; tail -3 $OUT/r04_bench_d.err; python - <<PY
import json
d=json.load(open("$OUT/r04_bench_d.json"))
print({k:d[k] for k in ("value","ms_per_step")})
for k in ("step_with_ms_ssim","drop_in_step","trained_like_step","independent_tensors_step"):
    print(k, {a:b for a,b in (d.get(k) or {}).items() if a!="what" and a!="workload"})
PY
