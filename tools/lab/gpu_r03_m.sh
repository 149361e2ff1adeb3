#!/bin/bash
# round 3: postponed leaves in the persistent closest-hit kernel (A/B, threshold sweep, lane utilisation), parity tests
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03m
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_postpone_r03.json)"
RT_BATCH=20 timeout 900 python tools/variants.py run bistro 20 > $OUT/variants_postpone.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_postpone.txt | tail -16
unset RT_VARIANTS
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_build.py tests/test_gpu_baseline_configs.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
