#!/bin/bash
# Runs on the GPU box (via gpurun): the numbers and profiles that get committed under profiles/<tag>/.
# Usage: tools/gpu_final.sh <tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
echo "== parity report"; python tools/parity_report.py 2>&1 | grep -v Warning | tee $OUT/parity_report.txt
echo "== shard emulation"; timeout 300 python tools/shard_emulation.py bistro 480 2>&1 | grep "^N=" | tee $OUT/shard_emulation.txt
for wl in bistro sponza cornell principled; do
  echo "== bench $wl"
  timeout 400 python bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || tail -3 $OUT/bench_$wl.err
  tail -c 400 $OUT/bench_$wl.json
done
cd /tmp
echo "== rocprofv3 --kernel-trace --stats of the default bench command (CPU baseline skipped)"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bistro -- python $REPO/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
grep '"metric"' $OUT/stats.log | tail -c 600
for f in $(find $OUT/stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/bistro_kernel_stats.csv; head -14 $f; done
echo "== PMC: FETCH_SIZE, WRITE_SIZE (separate passes), the default bench command"
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -o bistro -- python $REPO/bench.py --no-cpu-baseline > $OUT/pmc_$pmc.log 2>&1
done
echo "== PMC calibration on a known gather: tools/gather_bench (64-byte random node reads)"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_gather -o gather -- $REPO/tools/_build/gather_bench > $OUT/gather_bench.txt 2>&1
python3 $REPO/tools/summarize_pmc.py $OUT pmc_ x > $OUT/pmc_summary.txt 2>&1; grep "k_trace\|k_shade\|k_lane\|k_quad" $OUT/pmc_summary.txt | head -30
find $OUT -name '*.csv' -size +4M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
