#!/bin/bash
# Round 4, GPU call B: reproducer + new tests, instruction-rate and traffic-calibration micro-benchmarks, the suite (plain and
# under poison mode), 200 headline steps under poison mode, a quick bench with the MS-SSIM side object.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 120 python tools/repro_stream_reuse.py > $OUT/r04_repro_stream.json 2> $OUT/r04_repro_stream.err
echo "repro rc=$?"; cat $OUT/r04_repro_stream.json
timeout 120 tools/ubench/valu_rate4 > $OUT/r04_valu_rate4.txt 2>&1; echo "valu_rate4 rc=$?"; cat $OUT/r04_valu_rate4.txt
( cd /tmp && export TMPDIR=/tmp
  timeout 120 $REPO/tools/ubench/traffic_calib > $OUT/r04_traffic_calib_known.json 2> $OUT/r04_traffic_calib.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/calib_r04_$C -o pmc -- $REPO/tools/ubench/traffic_calib > $OUT/calib_r04_$C.log 2>&1
  done
  find $OUT/calib_r04_* -type f -size +4M -delete )
cat $OUT/r04_traffic_calib_known.json
timeout 1200 python -m pytest tests -m gpu -q -x -s 2>&1 | tail -25 > $OUT/r04_suite.log
echo "suite rc=${PIPESTATUS[0]}"; tail -8 $OUT/r04_suite.log
LARA2DGS_POISON_BUFFERS=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/r04_poison_suite.log
echo "poison suite rc=${PIPESTATUS[0]}"; tail -5 $OUT/r04_poison_suite.log
LARA2DGS_POISON_BUFFERS=1 timeout 900 python bench.py --steps 200 --warmup 2 --no-side-legs --no-cpu-baseline --no-roofline > $OUT/r04_poison_bench200.json 2> $OUT/r04_poison_bench200.err
echo "poison bench rc=$?"; cut -c1-300 $OUT/r04_poison_bench200.json; tail -3 $OUT/r04_poison_bench200.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/r04_bench_quick.json 2> $OUT/r04_bench_quick.err
echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/r04_bench_quick.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("step_with_ms_ssim"))
PY
