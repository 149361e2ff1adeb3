#!/bin/bash
# round 4: the out-of-cache data point of the roofline -- the same kernels on a 12 M-triangle atrium (~0.9 GB of nodes + triangle records)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04g
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 900 python bench.py --workload bistro12m --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_bistro12m.json 2> $OUT/bench_bistro12m.err; echo "bench12m exit $?"; tail -3 $OUT/bench_bistro12m.err
python3 -c "
import json; d=json.load(open('$OUT/bench_bistro12m.json')); r=d['roofline']; print('12m', round(d['value'],1), 'Msamples/s', d['config']['unique_tris'], 'tris', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, 'alg', {k: (round(v,2) if isinstance(v,float) else v) for k,v in r['algorithmic'].items() if k!='reference_bvh2'}, 'build', round(d['scene_build_s'],1))"
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${tag} -o b12 -- python $REPO/bench.py --workload bistro12m --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_${tag}.log 2>&1
done
python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic_12m.json bistro12m 2 0 64 64 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/kernel_hbm_bistro12m.txt $OUT/pmc_SQ_INSTS_VALU > $OUT/k2_traffic.log 2>&1; head -1 $OUT/k2_traffic.log | cut -c1-700
cat $OUT/kernel_hbm_bistro12m.txt | cut -c1-170
python3 $REPO/tools/summarize_pmc.py $OUT pmc_TCC x 2>/dev/null | grep "TCC" | grep "k_trace\|k_surface" | cut -c1-200
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
