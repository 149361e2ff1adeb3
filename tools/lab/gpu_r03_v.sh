#!/bin/bash
# round 3: next-event estimation over the compacted queue of points that got a light
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03v
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_nee_r03.json)"
for wl in bistro sponza principled cornell bistro_tex; do
  [ $wl = bistro ] || python bench.py --workload $wl --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
  RT_BATCH=20 timeout 900 python tools/variants.py run $wl 20 > $OUT/variants_nee_$wl.txt 2>&1; echo "variants $wl exit $?"
  grep -v "^    " $OUT/variants_nee_$wl.txt | tail -3
done
unset RT_VARIANTS
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/gputest.log
