#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the default bench command + HBM/L2 counter passes.
# Usage: tools/gpu_profile.sh <tag> [workload]
TAG=${1:-r01}
WL=${2:-bistro}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export RAY_AMD_CACHE=/tmp/ray_amd_cache
cd /tmp
echo "== rocprofv3 --kernel-trace --stats (same command as the default bench, CPU baseline skipped)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_stats -o $WL -- python $REPO/bench.py --workload $WL --no-cpu-baseline > $OUT/${TAG}_prof_stats.log 2>&1
grep '"metric"' $OUT/${TAG}_prof_stats.log | tail -c 1200
for f in $(find $OUT/${TAG}_prof_stats -name '*kernel_stats.csv' | head -1); do echo $f; head -12 $f; done
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  echo "== rocprofv3 --pmc $pmc (8 steps)"
  timeout 400 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/${TAG}_prof_$name -o $WL -- python $REPO/bench.py --workload $WL --no-cpu-baseline --steps 8 --warmup 1 > $OUT/${TAG}_prof_$name.log 2>&1
  ls $OUT/${TAG}_prof_$name/* | head -5
done
python3 $REPO/tools/summarize_pmc.py $OUT $TAG $WL | tee $OUT/${TAG}_pmc_summary.txt
find $OUT -name '*.csv' -size +6M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
