#!/bin/bash
# HBM traffic counters (separate PMC passes, as MI355X_MICROARCH.md section HBM prescribes):
#   tools/gpu_traffic.sh <tag> [bench args]  -> gpurun_out/traffic_<tag>_{fetch,write}/
set -u
TAG=${1:-r01}
EXTRA=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/traffic_${TAG}_$C -o pmc -- \
      python $REPO/bench.py --steps 1 --warmup 0 --scenes 1 --step raster --raster-api loop --streams 1 --no-fine --no-cpu-baseline --no-roofline $EXTRA > $OUT/traffic_${TAG}_$C.log 2>&1
  # the forward-only (inference) instantiation of the composite: per-view calls under no_grad
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/traffic_${TAG}_${C}_fwdonly -o pmc -- \
      python $REPO/tools/fwdonly_probe.py --steps 1 --scenes 1 --only per_view --out $OUT/traffic_${TAG}_${C}_fwdonly.json > $OUT/traffic_${TAG}_${C}_fwdonly.log 2>&1
done
find $OUT/traffic_${TAG}_* -type f -size +8M -delete
find $OUT/traffic_${TAG}_* -type f | head
