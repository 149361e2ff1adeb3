#!/bin/bash
# round 4: the light pick with lane refill: parity, wait thresholds on the headline scene (64-layer passes), the other workloads
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04n
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_shade_kernel.py -m gpu -q -x -k "pick or golden or frame or deterministic or shade" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
export RT_VARIANTS="$(cat tools/variants_pick_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_pick_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_pick_bistro64.txt | cut -c1-200
export RT_VARIANTS='{"base": [], "chunked": ["+env:RAYHIP_SHADE_SPLIT=5"]}'
for w in sponza principled cornell; do
RT_BATCH=64 timeout 300 python tools/variants.py run $w 64 > $OUT/variants_pick_${w}64.txt 2>&1
grep -v "^    " $OUT/variants_pick_${w}64.txt | cut -c1-200
done
