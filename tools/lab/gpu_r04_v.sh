#!/bin/bash
# round 4: the surface stage must not read an ior plane the passes no longer write
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04v
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stale_ior or ior_plane or golden or frame" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
# the same test against a library built BEFORE the fix (tools/variants.py kept one): it must fail there, or it pins nothing
if [ -f ray_amd/csrc/_build/variants/base/librayhip.so ]; then
  cp ray_amd/csrc/_build/librayhip.so /tmp/fixed.so; cp ray_amd/csrc/_build/variants/base/librayhip.so ray_amd/csrc/_build/librayhip.so
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stale_ior" > $OUT/gputest_before_fix.log 2>&1; echo "before the fix: pytest exit $? (1 = the test catches the bug)"; tail -3 $OUT/gputest_before_fix.log | cut -c1-200
  cp /tmp/fixed.so ray_amd/csrc/_build/librayhip.so
fi
