#!/bin/bash
# round 3: compare-and-swap forms (VALU microbenchmark); 8-wide walk with the pinned fetch; 6 waves
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03p
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 tools/_build/valu_bench > $OUT/valu_bench.txt 2>&1; echo "valu_bench exit $?"; tail -8 $OUT/valu_bench.txt
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_fold2_r03.json)"
RT_BATCH=20 timeout 900 python tools/variants.py run bistro 20 > $OUT/variants_fold2.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_fold2.txt | tail -16
