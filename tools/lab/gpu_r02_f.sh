#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1   # build + cache the scene
for v in "RAYHIP_REFINE_LEAVES=0" "RAYHIP_REFINE_LEAVES=1" "RAYHIP_REFINE_LEAVES=2" "RAYHIP_REFINE_LEAVES=3" "RAYHIP_REFINE_LEAVES=4" "RAYHIP_REBUILD_BVH=2" "RAYHIP_REBUILD_BVH=4"; do
  env $v RAYHIP_TRACE_UPLOAD=1 timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_$v.json')); a=d['roofline']['algorithmic']; print('$v', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), 'ms/spp', {k: round(v) for k,v in d['stage_us_per_step'].items()}, 'nodes4/ray', round(a['nodes4_per_ray'],2), 'tris/ray', round(a['tris_per_ray'],2))"
  grep "rayhip_scene_upload" $OUT/bench_$v.err | tail -3
done
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/gputest.log
