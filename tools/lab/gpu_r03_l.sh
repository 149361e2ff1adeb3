#!/bin/bash
# round 3: VALU issue cost per instruction kind (extended), lane utilisation of K2 from the profile build
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03l
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 tools/_build/valu_bench > $OUT/valu_bench.txt 2>&1; echo "valu_bench exit $?"; cat $OUT/valu_bench.txt
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS='{"prof": ["+trace:-DRT_PROFILE_TRACE"]}'
RT_BATCH=20 RT_PROF_RAW=1 timeout 600 python tools/variants.py run bistro 20 > $OUT/prof.txt 2>&1; echo "prof exit $?"; cat $OUT/prof.txt | tail -50
