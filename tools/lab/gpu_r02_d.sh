#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/gputest.log; grep -E "1080p|1024\^2|2048\^2" $OUT/gputest.log | cut -c1-220
for mode in 0 1; do
  RAYHIP_EXACT_SHADE=$mode timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench64_exact$mode.json 2> $OUT/bench64_exact$mode.err
  python3 -c "
import json; d=json.load(open('$OUT/bench64_exact$mode.json')); print('exact=$mode', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), 'ms/spp', {k: round(v) for k,v in d['stage_us_per_step'].items()})"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bistro -- python $REPO/bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
for f in $(find $OUT/prof_stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/kernel_stats_steps64.csv; done
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats_steps64.csv')))
for r in rows[:12]:
    n=r['Name'].split('(')[0].replace('void rt::','').replace('rt::','')
    print(f"{n[:44]:44s} calls {int(r['Calls']):4d} per-spp(128) {int(r['TotalDurationNs'])/1e6/128:6.3f} ms  {r['Percentage']}%")
PY
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
