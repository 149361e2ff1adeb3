#!/bin/bash
# round 3: triangle records at a pitch of 64 bytes (one sector each) against the reference's 48-byte array
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03j
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_tripitch_r03.json)"
RT_BATCH=32 timeout 900 python tools/variants.py run bistro 32 > $OUT/variants_tripitch.txt 2>&1; echo "variants exit $?"
grep -v "^  " $OUT/variants_tripitch.txt | tail -8
unset RT_VARIANTS
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_build.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
