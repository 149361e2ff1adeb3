#!/bin/bash
# round 3: blocks per resident wave slot of the persistent closest-hit kernel (the per-block drain against the end of the launch)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03ad
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_mult_r03.json)"
for K in 20 64; do
  RT_BATCH=$K timeout 900 python tools/variants.py run bistro $K > $OUT/variants_mult_$K.txt 2>&1; echo "variants $K exit $?"
  grep -v "^    " $OUT/variants_mult_$K.txt | tail -6
done
