#!/bin/bash
# round 2, final measurements: the driver's bench command as the first process on the box, the 64-spp headline, the full GPU
# suite, kernel stats + FETCH/WRITE counters of both commands (per-kernel HBM tables), shard emulation
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench20 exit $?"
timeout 600 python bench.py --steps 64 --warmup 64 > $OUT/bench_steps64.json 2> $OUT/bench_steps64.err; echo "bench64 exit $?"
for f in bench_steps20 bench_steps64; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), 'ms/spp', {k: round(v) for k,v in d['stage_us_per_step'].items()}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"; done
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/gputest.log | head -3
cd /tmp
for cfg in "20 5" "64 64"; do
  set -- $cfg; K=$1; W=$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/stats_$K.log 2>&1
  cp $(find $OUT/stats_$K -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps${K}_warmup$W.csv
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${pmc}_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/pmc_${pmc}_$K.log 2>&1
  done
  IPP=$(python3 -c "import json; print(json.load(open('$OUT/bench_steps$K.json'))['config']['iterations_per_pass'])")
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro $K $W $IPP $OUT/pmc_FETCH_SIZE_$K $OUT/pmc_WRITE_SIZE_$K $OUT/kernel_hbm_steps$K.txt > $OUT/k2_traffic_$K.log
  head -1 $OUT/k2_traffic_$K.log
done
cd $REPO
for w in bistro_tex sponza cornell principled; do
  timeout 600 python bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_${w}_steps64.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_${w}_steps64.json')); print('$w', round(d['value'],1), 'Msamples/s')"
done
python tools/shard_emulation.py bistro 64 20 > $OUT/shard_emulation.txt 2>&1; tail -10 $OUT/shard_emulation.txt
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
