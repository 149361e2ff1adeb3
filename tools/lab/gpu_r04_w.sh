#!/bin/bash
# round 4, final kernels: the out-of-cache data point again (12 M triangles) -- counters, table entry next to the headline scene's, bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04w
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --workload bistro12m --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/build.err; echo "scene built $?"
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${tag} -o b12 -- python $REPO/bench.py --workload bistro12m --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_${tag}.log 2>&1
done
cp $REPO/profiles/r04/k2_traffic.json $OUT/k2_traffic.json
python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro12m 2 0 64 64 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/kernel_hbm_bistro12m.txt $OUT/pmc_SQ_INSTS_VALU > $OUT/k2_traffic.log 2>&1; head -1 $OUT/k2_traffic.log | cut -c1-600
head -12 $OUT/kernel_hbm_bistro12m.txt | cut -c1-150
python3 $REPO/tools/summarize_pmc.py $OUT pmc_TCC x 2>/dev/null | grep "TCC" | grep "refill" | cut -c1-160 > $OUT/tcc_bistro12m.txt; cat $OUT/tcc_bistro12m.txt
cd $REPO
cp $OUT/k2_traffic.json profiles/r04/k2_traffic.json
timeout 600 python bench.py --workload bistro12m --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_bistro12m.json 2> $OUT/b.err
python3 -c "
import json; d=json.load(open('$OUT/bench_bistro12m.json')); r=d['roofline']; print('12m', round(d['value'],1), 'frac', round(r['frac'],3), 'stale', r.get('traffic_is_stale'), r.get('achieved_is'), (r.get('valu_issue') or {}).get('frac'))"
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
