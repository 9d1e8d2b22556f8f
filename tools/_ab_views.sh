cd $GRAFT_REPO_ROOT
for F in 0 256 0 256; do
  LARA2DGS_DEBUG_FLAGS=$F timeout 600 python bench.py --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline --no-roofline > gpurun_out/ab_views_$F.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ab_views_$F.json')); print('flags $F', d['value'], d['ms_per_step'])"
  LARA2DGS_DEBUG_FLAGS=$F timeout 600 python bench.py --steps 10 --warmup 3 --step raster --no-side-legs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  raster views flags $F', d['value'], d['ms_per_step'])"
  LARA2DGS_DEBUG_FLAGS=$F timeout 600 python bench.py --steps 10 --warmup 3 --step raster --regime trained --no-side-legs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  raster trained flags $F', d['value'], d['ms_per_step'])"
done
