#!/bin/bash
# round 3: does per-XCD spatial affinity (sorted rays + each XCD taking a contiguous part of the sorted list) raise the L2 hit
# rate of the closest-hit kernel?  (K2 time is reported apart from the sort's.)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03g
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_xcd_sort_r03.json)"
RT_BATCH=8 timeout 900 python tools/variants.py run bistro 8 > $OUT/variants_xcd_sort.txt 2>&1; echo "variants exit $?"
grep -v "^  " $OUT/variants_xcd_sort.txt | tail -8
