#!/bin/bash
# round 4: the flat any-hit kernel: parity of both forms, wait thresholds / waves per SIMD on the headline scene (64-layer passes)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04l
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "flat_shadow or shadow or baseline or config" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
export RT_VARIANTS="$(cat tools/variants_shadow_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_shadow_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_shadow_bistro64.txt | cut -c1-260
