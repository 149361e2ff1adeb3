#!/bin/bash
# round 5, final measurements: the driver's bench command first, the whole GPU suite, kernel stats of the driver's command, FETCH / WRITE (/ VALU)
# counters for EVERY BASELINE workload (k2_traffic.json with K2 and K3, per-kernel HBM tables), the bench lines again with the fresh table, a rank
# of 8 (emulation + per-kernel ratio), the UNet's kernel stats, two emulated ranks
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_first_process.json 2> $OUT/bench_first.err; echo "bench (first process) exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench_first_process.json')); print('first', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms/frame', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $OUT/gputest_final.log 2>&1
echo "pytest exit $?"; grep "passed\|failed" $OUT/gputest_final.log | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bistro -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps20_warmup5.csv
python3 $REPO/tools/pass_timeline.py $OUT/stats > $OUT/pass_timeline.txt 2>&1; tail -1 $OUT/pass_timeline.txt
rm -rf $OUT/stats
cp $REPO/profiles/r05/k2_traffic.json $OUT/k2_traffic.json 2>/dev/null
for w in bistro sponza cornell principled bistro_tex bistro12m; do
  steps=4; [ $w = bistro12m ] && steps=2
  timeout 600 python $REPO/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/build_$w.err   # scene built + cached
  passes="FETCH_SIZE WRITE_SIZE"; [ $w = bistro ] && passes="FETCH_SIZE WRITE_SIZE VALU"
  for tag in $passes; do
    pmc=$tag; [ $tag = VALU ] && pmc="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${w}_${tag} -o $w -- python $REPO/bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline > $OUT/pmc_${w}_${tag}.log 2>&1
  done
  valu=""; [ $w = bistro ] && valu=$OUT/pmc_${w}_VALU
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json $w $steps 1 64 64 $OUT/pmc_${w}_FETCH_SIZE $OUT/pmc_${w}_WRITE_SIZE $OUT/kernel_hbm_${w}.txt $valu > $OUT/k2_traffic_$w.log 2>&1
  head -1 $OUT/k2_traffic_$w.log | cut -c1-400
  head -14 $OUT/kernel_hbm_${w}.txt | cut -c1-170
  find $OUT -name '*.csv' -size +4M -delete; find $OUT -name '*.db' -delete
done
cd $REPO
mkdir -p profiles/r05 && cp $OUT/k2_traffic.json profiles/r05/k2_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20.err; echo "bench (with table) exit $?"
python3 -c "
import json
d=json.load(open('$OUT/bench_steps20_warmup5.json')); r=d['roofline']; t=r['traversal']; print('final', round(d['value'],1), 'K2 frac', round(r['frac'],3), 'stale', r.get('traffic_is_stale'), 'K2+K3 frac', t.get('frac'), 'valu', {k: (round(v,3) if isinstance(v,float) else v) for k,v in (r.get('valu_issue') or {}).items() if k in ('frac','frac_paired_model','active_lanes_of_64','mix_is_stale')}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
for w in bistro_tex sponza cornell principled bistro12m; do
  timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_${w}.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_${w}.json')); r=d['roofline']; print('$w', round(d['value'],1), 'Msamples/s', 'K2 frac', r['frac'], 'K2+K3 frac', r['traversal'].get('frac'))"
done
for cfg in "default:RAYHIP_CENSUS=1" "round4_schedule:RAYHIP_CENSUS=0 RAYHIP_OVERLAP_SHADOW=0"; do
  name=${cfg%%:*}; env=${cfg#*:}
  echo "== $name" >> $OUT/shard_emulation.txt
  env $env timeout 300 python tools/shard_emulation.py bistro 64 20 2>&1 | grep -v amdgpu.ids >> $OUT/shard_emulation.txt
done
grep "N=8\|^==" $OUT/shard_emulation.txt
cd /tmp
for world in 1 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard_w$world -o shard -- python $REPO/tools/shard_profile.py $world 64 > $OUT/shard_w$world.log 2>&1
  cp $(find $OUT/shard_w$world -name '*kernel_stats.csv' | head -1) $OUT/shard_kernel_stats_w$world.csv
  [ $world = 8 ] && python3 $REPO/tools/pass_timeline.py $OUT/shard_w$world > $OUT/timeline_rank0_of_8.txt 2>&1
  rm -rf $OUT/shard_w$world
done
for form in f16 f32; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unet_$form -o unet -- python $REPO/tools/unet_bench.py 6 $form > $OUT/unet_bench_$form.log 2>&1; tail -1 $OUT/unet_bench_$form.log
  cp $(find $OUT/unet_$form -name '*kernel_stats.csv' | head -1) $OUT/unet_kernel_stats_$form.csv; rm -rf $OUT/unet_$form
done
cd $REPO
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_2ranks_emulated.json 2> $OUT/b2.err; echo "2 ranks exit $?"
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -delete; du -sh $OUT
