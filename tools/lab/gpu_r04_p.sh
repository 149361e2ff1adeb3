#!/bin/bash
# round 4: the compiler's SLP vectoriser (v_pk_mul / v_pk_add + the moves that pair their operands) on / off per translation unit
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04p
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_slp_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_slp_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_slp_bistro64.txt | cut -c1-200
for w in sponza principled; do
RT_BATCH=64 timeout 300 python tools/variants.py run $w 64 > $OUT/variants_slp_${w}64.txt 2>&1
grep -v "^    " $OUT/variants_slp_${w}64.txt | cut -c1-200
done
