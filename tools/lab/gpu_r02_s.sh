#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02s
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for v in "X=0" "RAYHIP_PRIMARY_WAVES=5" "RAYHIP_PRIMARY_WAVES=4" "RAYHIP_SHADOW_WAVES=5" "RAYHIP_SHADOW_WAVES=4" "X=0"; do
  env $v timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); s=d['stage_us_per_step']; print('$v', round(d['value'],1), 'Msamples/s primary_trace', s['primary_trace'], 'shadow', s['primary_shadow'] + s['secondary_shadow'])"
done
