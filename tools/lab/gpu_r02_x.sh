#!/bin/bash
# is the slow start of a timed pass the previous process's memory being released?  back-to-back runs vs runs after a pause
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02x
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for pause in 0 0 0 10 10 10 0 0; do
  sleep $pause
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); s=d['stage_us_per_step']; print('pause $pause s:', round(d['value'],1), 'Msamples/s  gen', round(s['primary_ray_gen']), 'ptrace', round(s['primary_trace']))"
done
