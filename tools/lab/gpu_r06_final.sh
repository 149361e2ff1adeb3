#!/bin/bash
# round 6, final measurements (-> profiles/r06/): the driver's bench command as the first process on the box, smoke, the whole GPU suite, kernel
# stats + per-launch timeline of the driver's command, FETCH / WRITE (/ VALU) counter profiles for EVERY workload of the bench table (k2_traffic.json
# with K2 and K3, per-kernel HBM tables), the bench lines again with the fresh table, a rank of 8 on one GPU, the UNet's kernel stats, two
# emulated ranks, the material ball
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_first_process.json 2> $OUT/bench_first.err; echo "bench (first process) exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench_first_process.json')); print('first', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms/frame', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, 'stage sum', round(d['stage_sum_over_step'],3), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/gputest_final.log 2>&1
echo "pytest exit $?"; grep "passed\|failed" $OUT/gputest_final.log | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bistro -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps20_warmup5.csv
python3 $REPO/tools/pass_timeline.py $OUT/stats > $OUT/pass_timeline.txt 2>&1; tail -1 $OUT/pass_timeline.txt
rm -rf $OUT/stats
cp $REPO/profiles/r06/k2_traffic.json $OUT/k2_traffic.json 2>/dev/null
for w in bistro bistro_assets bistro_assets_inst sponza cornell principled bistro_tex bistro12m; do
  steps=4; [ $w = bistro12m ] && steps=2
  timeout 900 python $REPO/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/build_$w.err   # scene built + cached
  passes="FETCH_SIZE WRITE_SIZE"; case $w in bistro|bistro_assets|bistro_assets_inst) passes="FETCH_SIZE WRITE_SIZE VALU";; esac
  for tag in $passes; do
    pmc=$tag; [ $tag = VALU ] && pmc="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${w}_${tag} -o $w -- python $REPO/bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline > $OUT/pmc_${w}_${tag}.log 2>&1
  done
  valu=""; [ -d $OUT/pmc_${w}_VALU ] && valu=$OUT/pmc_${w}_VALU
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json $w $steps 1 64 64 $OUT/pmc_${w}_FETCH_SIZE $OUT/pmc_${w}_WRITE_SIZE $OUT/kernel_hbm_${w}.txt $valu > $OUT/k2_traffic_$w.log 2>&1
  head -1 $OUT/k2_traffic_$w.log | cut -c1-300
  head -14 $OUT/kernel_hbm_${w}.txt | cut -c1-170
  find $OUT -name '*.csv' -size +4M -delete; find $OUT -name '*.db' -delete
done
cd $REPO
mkdir -p profiles/r06 && cp $OUT/k2_traffic.json profiles/r06/k2_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20.err; echo "bench (with table) exit $?"
python3 -c "
import json
d=json.load(open('$OUT/bench_steps20_warmup5.json')); r=d['roofline']; t=r['traversal']; print('final', round(d['value'],1), 'K2 frac', round(r['frac'],3), 'stale', r.get('traffic_is_stale'), 'K2+K3 frac', t.get('frac'), 'valu', {k: (round(v,3) if isinstance(v,float) else v) for k,v in (r.get('valu_issue') or {}).items() if k in ('frac','frac_paired_model','active_lanes_of_64','mix_is_stale')}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
for w in bistro_assets bistro_assets_inst bistro_tex sponza cornell principled bistro12m; do
  extra="--no-cpu-baseline"; [ $w = bistro_assets ] && extra=""
  timeout 900 python bench.py --workload $w --steps 4 --warmup 1 $extra > $OUT/bench_${w}.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_${w}.json')); r=d['roofline']; a=r['algorithmic']; print('$w', round(d['value'],1), 'Msamples/s', 'K2 frac', r['frac'], 'K2+K3 frac', r['traversal'].get('frac'), '| per ray: tlas', round(a['tlas_nodes_per_ray'],2), 'wide', round(a['wide_nodes_per_ray'],2), 'tris', round(a['tris_per_ray'],2), 'inst', round(a['instances_per_ray'],2), 'lanes', (r.get('valu_issue') or {}).get('active_lanes_of_64'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"
done
echo "== default" >> $OUT/shard_emulation.txt
timeout 400 python tools/shard_emulation.py bistro 64 20 2>&1 | grep -v amdgpu.ids >> $OUT/shard_emulation.txt
grep "N=8\|^==" $OUT/shard_emulation.txt
cd /tmp
for form in f16 f32; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unet_$form -o unet -- python $REPO/tools/unet_bench.py 6 $form > $OUT/unet_bench_$form.log 2>&1; tail -1 $OUT/unet_bench_$form.log
  cp $(find $OUT/unet_$form -name '*kernel_stats.csv' | head -1) $OUT/unet_kernel_stats_$form.csv; rm -rf $OUT/unet_$form
done
cd $REPO
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_2ranks_emulated.json 2> $OUT/b2.err; echo "2 ranks exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench_2ranks_emulated.json')); print('2 ranks', round(d['value'],1), d.get('transport'), d.get('ncclCommCount'), d.get('render_ms_per_rank'))"
RAY_AMD_FORCE_DIST=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_force_dist_1rank.json 2> $OUT/b3.err; echo "force-dist exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench_force_dist_1rank.json')); print('1 rank through the N>1 path', round(d['value'],1), d.get('transport'), d.get('ncclCommCount'), d.get('exchange'))"
# random scenes on the final tree, DIRECTLY against the live reference: the form the census picks, and round 6's form pinned
timeout 1200 python tools/gpu_fuzz.py 14000 60 oracle 2>&1 | grep -v amdgpu.ids | tail -2 > $OUT/gpu_fuzz.txt
RAYHIP_SHADE_SPLIT=29 timeout 1200 python tools/gpu_fuzz.py 15000 60 oracle 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT/gpu_fuzz.txt
cat $OUT/gpu_fuzz.txt | cut -c1-250
timeout 900 python tools/material_ball_bench.py complex_mat5 64 > $OUT/material_ball_bench.txt 2>&1; tail -5 $OUT/material_ball_bench.txt | cut -c1-300
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -delete; du -sh $OUT
