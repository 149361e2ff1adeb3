#!/bin/bash
# tests, the driver's bench command as the first process on the box, the 64-spp headline, kernel stats + PMC traffic of both
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench20 exit $?"
timeout 600 python bench.py --steps 64 --warmup 64 > $OUT/bench_steps64.json 2> $OUT/bench_steps64.err; echo "bench64 exit $?"
for f in bench_steps20 bench_steps64; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), 'ms/spp', {k: round(v) for k,v in d['stage_us_per_step'].items()}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"; done
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/gputest.log
cd /tmp
for cfg in "20 5" "64 64"; do
  set -- $cfg; K=$1; W=$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/stats_$K.log 2>&1
  cp $(find $OUT/stats_$K -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_steps$K.csv
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${pmc}_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/pmc_${pmc}_$K.log 2>&1
  done
  IPP=$(python3 -c "import json; print(json.load(open('$OUT/bench_steps$K.json'))['config']['iterations_per_pass'])")
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro $K $W $IPP $OUT/pmc_FETCH_SIZE_$K $OUT/pmc_WRITE_SIZE_$K $OUT/kernel_stats_steps$K.csv
done
python3 - <<PY
import csv
for K in (20, 64):
    rows=list(csv.DictReader(open('$OUT/kernel_stats_steps%d.csv' % K)))
    print("steps", K)
    for r in rows[:9]:
        n=r['Name'].split('(')[0].replace('void rt::','').replace('rt::','')
        print(f"  {n[:44]:44s} calls {int(r['Calls']):4d} total {int(r['TotalDurationNs'])/1e6:8.2f} ms avg {int(r['TotalDurationNs'])/1e6/int(r['Calls']):8.3f} ms {r['Percentage']}%")
PY
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
