#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines for the workloads + rocprofv3 kernel stats and HBM counters.
# Usage: tools/gpu_bench_profile.sh <tag>     -> writes gpurun_out/<tag>_*.{json,log,csv}
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
export RAY_AMD_CACHE=/tmp/ray_amd_cache

echo "== bench bistro (default)"; timeout 600 python bench.py > $OUT/${TAG}_bench_bistro.json 2> $OUT/${TAG}_bench_bistro.err; tail -c 3000 $OUT/${TAG}_bench_bistro.json
echo "== bench sponza"; timeout 300 python bench.py --workload sponza > $OUT/${TAG}_bench_sponza.json 2> $OUT/${TAG}_bench_sponza.err; tail -c 3000 $OUT/${TAG}_bench_sponza.json
echo "== bench cornell"; timeout 300 python bench.py --workload cornell --steps 256 > $OUT/${TAG}_bench_cornell.json 2> $OUT/${TAG}_bench_cornell.err; tail -c 2500 $OUT/${TAG}_bench_cornell.json

cd /tmp
echo "== rocprofv3 kernel stats (bistro, same command as the default bench minus the CPU baseline)"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_stats -o bistro -- python $REPO/bench.py --no-cpu-baseline > $OUT/${TAG}_prof_stats.log 2>&1
tail -c 1500 $OUT/${TAG}_prof_stats.log
find $OUT/${TAG}_prof_stats -name '*kernel_stats*' | head -3
for f in $(find $OUT/${TAG}_prof_stats -name '*kernel_stats.csv' | head -1); do head -20 $f; done

echo "== rocprofv3 PMC passes (bistro, 8 steps)"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_prof_fetch -o bistro -- python $REPO/bench.py --no-cpu-baseline --steps 8 --warmup 1 > $OUT/${TAG}_prof_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_prof_write -o bistro -- python $REPO/bench.py --no-cpu-baseline --steps 8 --warmup 1 > $OUT/${TAG}_prof_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/${TAG}_prof_l2 -o bistro -- python $REPO/bench.py --no-cpu-baseline --steps 8 --warmup 1 > $OUT/${TAG}_prof_l2.log 2>&1
ls -la $OUT/${TAG}_prof_fetch/* 2>/dev/null | head
# keep only the small CSVs (merge limit 64 MiB): drop per-dispatch traces bigger than 8 MB
find $OUT -name '*.csv' -size +8M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
