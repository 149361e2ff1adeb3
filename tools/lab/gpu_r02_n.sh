#!/bin/bash
# round 2, run N: refill default on -- full suite, small scenes with and without, driver command
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02n
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -9 $OUT/gputest.log
for w in cornell principled sponza; do for v in "RAYHIP_REFILL=0" "RAYHIP_REFILL=2" "RAYHIP_REFILL=2 RAYHIP_REFILL_SMALL=1"; do
  env $v timeout 600 python bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$w $v', round(d['value'],1), 'Msamples/s', d['stage_us_per_step']['secondary_trace'], d['stage_us_per_step']['primary_trace'])"
done; done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err
python3 -c "
import json; d=json.load(open('$OUT/bench_20_5.json')); print('driver cmd', round(d['value'],1), 'Msamples/s', d['stage_us_per_step'], d['parity'])"
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_64_64.json 2> $OUT/b.err
python3 -c "
import json; d=json.load(open('$OUT/bench_64_64.json')); print('64/64', round(d['value'],1), 'Msamples/s', d['stage_us_per_step'])"
