#!/bin/bash
# round 4: the pooled kernel, second form (swap as a third step kind inside the walk loop, index prefetch, slow-path state in a slab)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04c
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refill or wide_forms or closest_hit or deterministic" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/gputest.log
export RT_VARIANTS="$(cat tools/variants_pool4_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_pool4_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_pool4_bistro64.txt | cut -c1-200
du -sh $OUT
