#!/bin/bash
# Round 5: every rocprofv3 pass behind profiles/r05_* (kernel stats of the headline step / the one-stream step / the one-stream step
# with the MS-SSIM term / the serialised raster, SQ counters, calibrated HBM traffic, both regimes; encoder forward / backward
# counters; traffic calibration incl. the AoS input patterns).  tools/gpu_r05_profiles.sh [tag]
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile.log 2>&1; echo "profile rc=$?"
bash tools/gpu_traffic.sh $TAG > $OUT/${TAG}_traffic.log 2>&1; echo "traffic rc=$?"
ONLY_RASTER=1 bash tools/gpu_profile.sh ${TAG}_trained "--regime trained" > $OUT/${TAG}_trained_profile.log 2>&1; echo "profile trained rc=$?"
bash tools/gpu_traffic.sh ${TAG}_trained "--regime trained" > $OUT/${TAG}_trained_traffic.log 2>&1; echo "traffic trained rc=$?"
bash tools/gpu_pmc_enc.sh $TAG > $OUT/${TAG}_pmc_enc.log 2>&1; echo "pmc enc rc=$?"
bash tools/gpu_pmc_train.sh $TAG > $OUT/${TAG}_encoder_train_pmc.txt 2>&1; echo "pmc train rc=$?"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_pipe1ms -o stats -- \
      python $REPO/bench.py --steps 2 --warmup 1 --streams 1 --ms-ssim --no-cpu-baseline --no-side-legs --no-roofline > $OUT/prof_${TAG}_pipe1ms.log 2>&1
  find $OUT/prof_${TAG}_pipe1ms -type f -size +8M -delete
  timeout 120 $REPO/tools/ubench/traffic_calib > $OUT/${TAG}_traffic_calib_known.json 2> $OUT/${TAG}_traffic_calib.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/calib_${TAG}_$C -o pmc -- $REPO/tools/ubench/traffic_calib > $OUT/calib_${TAG}_$C.log 2>&1
  done
  find $OUT/calib_${TAG}_* -type f -size +4M -delete )
echo "calib done"
timeout 300 python tools/glue_probe.py --steps 2 --top 60 > $OUT/${TAG}_glue.txt 2>&1; echo "glue rc=$?"
timeout 300 python tools/attn_bench.py --reps 20 > $OUT/${TAG}_attn_bench.txt 2>&1; echo "attn rc=$?"
timeout 300 python tools/kbench.py --reps 5 > $OUT/${TAG}_kbench.txt 2>&1; echo "kbench rc=$?"; grep -E "^\[|composite" $OUT/${TAG}_kbench.txt
du -sh $OUT
