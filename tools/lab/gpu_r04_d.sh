#!/bin/bash
# round 4: what does a lane-address (a 16-byte quarter of a node) cost the closest-hit kernel?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04d
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_touch_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_touch_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_touch_bistro64.txt | cut -c1-200
