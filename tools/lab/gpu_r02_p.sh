#!/bin/bash
# round 2, run P: what would ray ordering buy K2 today?  (sort keys of different coarseness, unbatched passes)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02p
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_sort_r02.json)"
timeout 1200 python tools/variants.py run bistro 16 2>&1 | grep -v "^  " | tee $OUT/variants_sort.txt
