#!/bin/bash
# Round 2, GPU call B: -m gpu suite on the staged shade kernels, then the shade-stage variant sweep
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/gputest.log
export RT_VARIANTS="$(cat tools/variants_shade_r02.json)"
RT_BATCH=32 timeout 1500 python tools/variants.py run bistro 32 > $OUT/variants_bistro.txt 2>&1
cat $OUT/variants_bistro.txt | cut -c1-230
