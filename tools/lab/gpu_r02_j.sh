#!/bin/bash
# round 2, run J: 2.5-ulp division / sqrt in the shade stage -- speed and parity
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_div_r02.json)"
RT_BATCH=32 timeout 900 python tools/variants.py run bistro 32 2>&1 | grep -v "^  " | tee $OUT/variants.txt
cp ray_amd/csrc/_build/librayhip.so /tmp/librayhip_base.so
for v in shade_div; do
  cp ray_amd/csrc/_build/variants/$v/librayhip.so ray_amd/csrc/_build/librayhip.so
  timeout 1500 python -m pytest tests -m gpu -q --durations=3 > $OUT/gputest_$v.log 2>&1
  echo "$v pytest exit $?"; tail -15 $OUT/gputest_$v.log
done
cp /tmp/librayhip_base.so ray_amd/csrc/_build/librayhip.so
