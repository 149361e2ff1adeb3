#!/bin/bash
# round 3: the 4-wide slab test with the error bound folded into the plane padding; service rounds of the persistent kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03o
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_fold_r03.json)"
RT_BATCH=20 timeout 900 python tools/variants.py run bistro 20 > $OUT/variants_fold.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_fold.txt | tail -16
unset RT_VARIANTS
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
