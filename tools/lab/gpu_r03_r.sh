#!/bin/bash
# round 3: compare-and-swap forms (VALU microbenchmark); 8-wide walk with the pinned fetch; 6 waves
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03r
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_prefetch_r03.json)"
RT_BATCH=20 timeout 900 python tools/variants.py run bistro 20 > $OUT/variants_prefetch.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_prefetch.txt | tail -16
unset RT_VARIANTS; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
