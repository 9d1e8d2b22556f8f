#!/bin/bash
# Round 6: every rocprofv3 pass behind profiles/r06_* -- bench lines (both regimes), kernel stats of the headline step / the one-stream
# step / the serialised raster, SQ counters, calibrated HBM traffic incl. the forward-only (inference) instantiation, both regimes;
# encoder forward / backward counters and kernel stats; the steady-state glue summary.  tools/gpu_r06_profiles.sh [tag]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 700 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_init.json 2> $OUT/${TAG}_bench_init.err; echo "bench init rc=$?"
timeout 700 python bench.py --steps 20 --warmup 5 --regime trained > $OUT/${TAG}_bench_trained.json 2> $OUT/${TAG}_bench_trained.err; echo "bench trained rc=$?"
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile.log 2>&1; echo "profile rc=$?"
bash tools/gpu_traffic.sh $TAG > $OUT/${TAG}_traffic.log 2>&1; echo "traffic rc=$?"
ONLY_RASTER=1 bash tools/gpu_profile.sh ${TAG}_trained "--regime trained" > $OUT/${TAG}_trained_profile.log 2>&1; echo "profile trained rc=$?"
bash tools/gpu_traffic.sh ${TAG}_trained "--regime trained" > $OUT/${TAG}_trained_traffic.log 2>&1; echo "traffic trained rc=$?"
bash tools/gpu_pmc_enc.sh $TAG > $OUT/${TAG}_pmc_enc.log 2>&1; echo "pmc enc rc=$?"
bash tools/gpu_pmc_train.sh $TAG > $OUT/${TAG}_encoder_train_pmc.txt 2>&1; echo "pmc train rc=$?"
bash tools/gpu_r05_enc_stats.sh $TAG cur > $OUT/${TAG}_enc_stats.log 2>&1; echo "enc stats rc=$?"
bash tools/gpu_r06_glue.sh $TAG > $OUT/${TAG}_glue.log 2>&1; echo "glue rc=$?"
python tools/steady_state_glue.py $OUT/prof_${TAG}_pipe1/stats_kernel_trace.csv --top 40 > $OUT/${TAG}_steady_state_glue.txt 2>&1
python tools/fwdonly_probe.py --out $OUT/${TAG}_fwdonly_probe.json > $OUT/${TAG}_fwdonly_probe.log 2>&1; echo "fwdonly rc=$?"
timeout 300 python tools/attn_bench.py --reps 20 > $OUT/${TAG}_attn_bench.txt 2>&1; echo "attn rc=$?"
find $OUT -name "*kernel_trace.csv" -size +6M -delete
du -sh $OUT
