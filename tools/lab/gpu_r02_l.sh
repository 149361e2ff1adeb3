#!/bin/bash
# round 2, run L: persistent refill kernel for the secondary bounces only
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02l
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for v in "RAYHIP_REFILL=0" "RAYHIP_REFILL=2" "RAYHIP_REFILL=2 RAYHIP_REFILL_MULT=2" "RAYHIP_REFILL=2 RAYHIP_REFILL_MULT=4" "RAYHIP_REFILL=2 RAYHIP_REFILL_MULT=16" "RAYHIP_REFILL=0"; do
  env $v timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$v', round(d['value'],1), 'Msamples/s', d['stage_us_per_step']['secondary_trace'], d['stage_us_per_step']['primary_trace'])"
done
for w in sponza; do for v in "RAYHIP_REFILL=0" "RAYHIP_REFILL=2 RAYHIP_REFILL_MULT=4"; do
  env $v timeout 600 python bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$w $v', round(d['value'],1), 'Msamples/s', d['stage_us_per_step']['secondary_trace'], d['stage_us_per_step']['primary_trace'])"
done; done
