#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02ad
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_xcd_r02.json)"
for w in bistro sponza; do RT_BATCH=64 timeout 1200 python tools/variants.py run $w 64 2>&1 | grep -v "^  " | tee -a $OUT/variants_xcd.txt; done
RT_BATCH=20 timeout 1200 python tools/variants.py run bistro 20 2>&1 | grep -v "^  " | tee -a $OUT/variants_xcd.txt
