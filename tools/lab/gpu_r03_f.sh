#!/bin/bash
# round 3: kernel statistics + per-launch timeline + HBM counters of the driver's command (profiles/r03)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03f
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 exit $?"
cd /tmp
K=20; W=5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/stats_$K.log 2>&1
cp $(find $OUT/stats_$K -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps${K}_warmup$W.csv
python3 $REPO/tools/pass_timeline.py $OUT/stats_$K > $OUT/pass_timeline_steps$K.txt; tail -75 $OUT/pass_timeline_steps$K.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${pmc}_$K -o bistro -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/pmc_${pmc}_$K.log 2>&1
done
IPP=$(python3 -c "import json; print(json.load(open('$OUT/bench20.json'))['config']['iterations_per_pass'])")
python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro $K $W $IPP $OUT/pmc_FETCH_SIZE_$K $OUT/pmc_WRITE_SIZE_$K $OUT/kernel_hbm_bench_steps${K}_warmup$W.txt > $OUT/k2_traffic_$K.log 2>&1
head -3 $OUT/k2_traffic_$K.log; cat $OUT/kernel_hbm_bench_steps${K}_warmup$W.txt | head -30
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
