#!/bin/bash
# round 2: FETCH/WRITE counters of the other workloads at 64 / 64 -> profiles/r02/k2_traffic.json entries + per-kernel HBM tables
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02ac
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
cp profiles/r02/k2_traffic.json $OUT/k2_traffic.json
cd /tmp
for w in sponza cornell principled bistro_tex; do
  python $REPO/bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_$w.json 2>/dev/null
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${pmc}_$w -o b -- python $REPO/bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline > $OUT/pmc_${pmc}_$w.log 2>&1
  done
  IPP=$(python3 -c "import json; print(json.load(open('$OUT/bench_$w.json'))['config']['iterations_per_pass'])")
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json $w 64 64 $IPP $OUT/pmc_FETCH_SIZE_$w $OUT/pmc_WRITE_SIZE_$w $OUT/kernel_hbm_${w}_steps64.txt | head -1 | cut -c1-400
done
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
