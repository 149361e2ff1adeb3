#!/bin/bash
# round 3, third GPU call: whole-chunk refill for the primary rays, noinline rare paths, 8- vs 4-wide on the other workloads,
# new tests (RendererHIP over several ranks, the drop-in sample, wide forms agree)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03c
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_rare_r03.json)"
RT_BATCH=32 timeout 900 python tools/variants.py run bistro 32 > $OUT/variants_rare.txt 2>&1; echo "variants exit $?"
grep -v "^  " $OUT/variants_rare.txt | tail -8
export RT_VARIANTS='{"w4": ["+env:RAYHIP_BVH_WIDTH=4"], "w8": ["+build:w4"]}'
for wl in sponza cornell principled; do
  RT_BATCH=32 timeout 600 python tools/variants.py run $wl 32 > $OUT/variants_width_$wl.txt 2>&1
  echo "== $wl"; grep -v "^  " $OUT/variants_width_$wl.txt | tail -2
done
unset RT_VARIANTS
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_renderer_devices.py tests/test_gpu_dropin.py tests/test_gpu_comm.py tests/test_gpu_instance_update.py -m gpu -q --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/gputest.log
