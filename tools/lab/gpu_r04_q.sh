#!/bin/bash
# round 4: thresholds / grids / register budgets once more with the final kernels
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04q
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_retune_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_retune_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_retune_bistro64.txt | cut -c1-200
