#!/bin/bash
# round 3: where the stages stand in 64-layer passes (the verdict's targets are quoted there), all workloads
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03x
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS='{"base": []}'
python tools/variants.py build > /dev/null 2>&1
for wl in bistro sponza cornell principled; do
  python bench.py --workload $wl --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
  RT_BATCH=64 timeout 600 python tools/variants.py run $wl 64 2>&1 | grep -v "^    " | tail -1 | tee -a $OUT/stages_64.txt
done
