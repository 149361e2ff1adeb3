#!/bin/bash
# round 2, run U: block-compressed textures on the device -- suite, textured workloads (plain / blocks / decoded at export),
# and the two bench commands with the final k2_traffic.json in place
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02u
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench20 exit $?"
timeout 600 python bench.py --steps 64 --warmup 64 > $OUT/bench_steps64.json 2> $OUT/bench_steps64.err; echo "bench64 exit $?"
for f in bench_steps20 bench_steps64; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); r=d['roofline']; print('$f', round(d['value'],1), 'Msamples/s', 'frac', round(r['frac'],3), 'achieved', round(r['achieved']), 'launch ms', round(r['avg_launch_ms'],2), 'profiled', r['traffic_detail']['profiled_avg_launch_ms'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"; done
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/gputest.log | head -2
for v in "bistro_tex X=0" "bistro_texc X=0" "bistro_texc RAY_HIP_DECODE_BC=1"; do
  set -- $v
  env $2 timeout 900 python bench.py --workload $1 --steps 64 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); s=d['stage_us_per_step']; print('$v', round(d['value'],1), 'Msamples/s shade', round(s['primary_shade']+s['secondary_shade']), 'us; scene build', round(d['scene_build_s'],1), 's')"
  cp $OUT/b.json "$OUT/bench_$1_$2.json"
done
