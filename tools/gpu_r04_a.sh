#!/bin/bash
# Round 4, GPU call A: the new stream-safety / full-size tests, the suite + 200 headline steps under poison mode, a quick bench.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 120 python tools/repro_stream_reuse.py > $OUT/r04_repro_stream.json 2> $OUT/r04_repro_stream.err
echo "repro rc=$?"; cat $OUT/r04_repro_stream.json
timeout 900 python -m pytest tests/test_stream_safety_gpu.py tests/test_full_size_gpu.py tests/test_coarsedec.py tests/test_pipeline.py -m gpu -q -x -s 2>&1 | tail -25 > $OUT/r04_new_tests.log
echo "new tests rc=${PIPESTATUS[0]}"; tail -12 $OUT/r04_new_tests.log
LARA2DGS_POISON_BUFFERS=1 timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/r04_poison_suite.log
echo "poison suite rc=${PIPESTATUS[0]}"; tail -5 $OUT/r04_poison_suite.log
LARA2DGS_POISON_BUFFERS=1 timeout 600 python bench.py --steps 200 --warmup 2 --no-side-legs --no-cpu-baseline --no-roofline > $OUT/r04_poison_bench200.json 2> $OUT/r04_poison_bench200.err
echo "poison bench rc=$?"; cut -c1-400 $OUT/r04_poison_bench200.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline --no-roofline > $OUT/r04_bench_quick.json 2> $OUT/r04_bench_quick.err
echo "bench rc=$?"; cut -c1-300 $OUT/r04_bench_quick.json
