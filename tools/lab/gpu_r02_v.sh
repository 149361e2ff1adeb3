#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02v
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  RAYHIP_TRACE_LAUNCH=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b$i.json 2> $OUT/b$i.err
  python3 -c "
import json; d=json.load(open('$OUT/b$i.json')); s=d['stage_us_per_step']; print('run $i', round(d['value'],1), 'gen us/step', round(s['primary_ray_gen']), 'ms_per_step', round(d['ms_per_step'],3))"
  grep "pass start" $OUT/b$i.err | tail -3 | tr '\n' ';'; echo
done
