#!/bin/bash
# round 3, final measurements: the driver's bench command as the first process on the box, the 64-spp headline, the full GPU
# suite, kernel stats + FETCH / WRITE / VALU counters of both commands (per-kernel HBM tables, k2_traffic.json), the other
# workloads, shard emulation, two emulated ranks, the VALU issue-rate table
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20.err; echo "bench20 exit $?"
timeout 600 python bench.py --steps 64 --warmup 64 > $OUT/bench_steps64_warmup64.json 2> $OUT/bench_steps64.err; echo "bench64 exit $?"
for f in bench_steps20_warmup5 bench_steps64_warmup64; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), 'ms/spp', {k: round(v) for k,v in d['stage_us_per_step'].items()}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"; done
timeout 600 python -m pytest tests -m gpu -q --durations=8 > $OUT/gputest_final.log 2>&1
echo "pytest exit $?"; grep "passed\|failed" $OUT/gputest_final.log | tail -2
cd /tmp
for cfg in "20 5" "64 64"; do
  set -- $cfg; K=$1; W=$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/stats_$K.log 2>&1
  cp $(find $OUT/stats_$K -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps${K}_warmup$W.csv
  python3 $REPO/tools/pass_timeline.py $OUT/stats_$K > $OUT/pass_timeline_steps$K.txt
  for pmc in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${tag}_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/pmc_${tag}_$K.log 2>&1
  done
  IPP=$(python3 -c "import json; print(json.load(open('$OUT/bench_steps${K}_warmup$W.json'))['config']['iterations_per_pass'])")
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro $K $W $IPP $OUT/pmc_FETCH_SIZE_$K $OUT/pmc_WRITE_SIZE_$K $OUT/kernel_hbm_bench_steps${K}_warmup$W.txt $OUT/pmc_SQ_INSTS_VALU_$K > $OUT/k2_traffic_$K.log 2>&1
  head -1 $OUT/k2_traffic_$K.log | cut -c1-600
done
cat $OUT/kernel_hbm_bench_steps20_warmup5.txt
cd $REPO
# the bench line again, now that the traffic table of THESE kernels exists (roofline.traffic exact, not stale)
mkdir -p profiles/r03 && cp $OUT/k2_traffic.json profiles/r03/k2_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20.err; echo "bench20 (with table) exit $?"
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_steps64_warmup64.json 2> $OUT/bench_steps64.err
python3 -c "
import json
for f in ('bench_steps20_warmup5', 'bench_steps64_warmup64'):
    d=json.load(open('$OUT/%s.json' % f)); r=d['roofline']; print(f, round(d['value'],1), 'frac', round(r['frac'],3), 'stale', r.get('traffic_is_stale'), 'valu', (r.get('valu_issue') or {}).get('frac_of_half_rate_class'))"
for w in bistro_tex sponza cornell principled; do
  timeout 600 python bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_${w}_steps64_warmup64.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_${w}_steps64_warmup64.json')); print('$w', round(d['value'],1), 'Msamples/s')"
done
timeout 200 python tools/shard_emulation.py bistro 64 20 > $OUT/shard_emulation.txt 2>&1; tail -12 $OUT/shard_emulation.txt
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2ranks_emulated_steps20.json 2> $OUT/b2.err; echo "2 ranks exit $?"
timeout 300 tools/_build/valu_bench > $OUT/valu_bench.txt 2>&1
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
