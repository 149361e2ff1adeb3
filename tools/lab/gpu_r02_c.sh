#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/gputest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bistro -- python $REPO/bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
grep '"metric"' $OUT/prof_stats.log | cut -c1-300
for f in $(find $OUT/prof_stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/kernel_stats_steps64.csv; head -16 $f | cut -c1-200; done
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
