#!/bin/bash
# round 4: four samples per primary wavefront and the refill threshold 48, twice each in one process sequence (run-to-run spread)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04s
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_s4_r04.json)"
for w in bistro sponza; do
RT_BATCH=64 timeout 600 python tools/variants.py run $w 64 > $OUT/variants_s4_${w}64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_s4_${w}64.txt | grep Msamples | cut -c1-140
done
