#!/bin/bash
# round 3: lane census of the scatter stage
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03u
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS='{"shade_prof": ["+shade:-DRT_PROFILE_SHADE"]}'
RT_BATCH=20 timeout 900 python tools/variants.py run bistro 20 > $OUT/shade_census.txt 2>&1; echo "variants exit $?"
cat $OUT/shade_census.txt | tail -30
