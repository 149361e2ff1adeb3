#!/bin/bash
# Round 2, first GPU call: host probe, whole -m gpu suite, the driver's bench command, the 64-spp headline, kernel stats.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
{
  echo "nproc: $(nproc)  cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)"
  python3 -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
  lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\(s\)|NUMA node\(s\)'
  free -g | head -2
} > $OUT/host.txt 2>&1
cat $OUT/host.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 -s > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench20 exit $?"
cut -c1-1500 $OUT/bench_steps20.json
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_steps64.json 2> $OUT/bench_steps64.err; echo "bench64 exit $?"
cut -c1-600 $OUT/bench_steps64.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bistro -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
for f in $(find $OUT/prof_stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/kernel_stats_steps20.csv; head -14 $f; done
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
