#!/bin/bash
# Usage: tools/gpu_pmc.sh <tag> <workload> "<pmc pass 1>" "<pmc pass 2>" ...   (runs on the GPU box)
TAG=$1; WL=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd /tmp
python $REPO/bench.py --workload $WL --no-cpu-baseline --steps ${PMC_STEPS:-60} --warmup 0 > /dev/null 2>&1   # build + cache the scene
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/${TAG}_pmc$i -o $WL -- python $REPO/bench.py --workload $WL --no-cpu-baseline --steps ${PMC_STEPS:-60} --warmup 0 > $OUT/${TAG}_pmc$i.log 2>&1 || tail -5 $OUT/${TAG}_pmc$i.log
done
python3 $REPO/tools/summarize_pmc.py $OUT ${TAG}_pmc x | grep -v 'rocclr\|k_fill\|k_accum\|k_raygen\|ILb1E' | tee $OUT/${TAG}_pmc_summary.txt
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
