#!/bin/bash
# round 2, run O: kernel stats + FETCH/WRITE counters of the 64-spp command, per-kernel HBM table
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02o
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_steps64.json 2> $OUT/bench_steps64.err; echo "bench64 exit $?"
cd /tmp
for cfg in "64 64"; do
  set -- $cfg; K=$1; W=$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/stats_$K.log 2>&1
  cp $(find $OUT/stats_$K -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_steps$K.csv
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${pmc}_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/pmc_${pmc}_$K.log 2>&1
  done
  IPP=$(python3 -c "import json; print(json.load(open('$OUT/bench_steps$K.json'))['config']['iterations_per_pass'])")
  python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro $K $W $IPP $OUT/pmc_FETCH_SIZE_$K $OUT/pmc_WRITE_SIZE_$K $OUT/kernel_hbm_steps$K.txt
done
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
