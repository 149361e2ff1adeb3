#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02t
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_pick_r02.json)"
RT_BATCH=64 timeout 1200 python tools/variants.py run bistro 64 2>&1 | grep -v "^  " | tee $OUT/variants_pick.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_shade_kernel.py -m gpu -q -x > $OUT/gputest.log 2>&1; tail -3 $OUT/gputest.log
