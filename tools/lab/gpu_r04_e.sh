#!/bin/bash
# round 4: BASELINE configs 2 and 5 at their stated 256 / 512 spp against RendererRef, once; the new GPU tests of the round so far
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04e
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_build.py -m gpu -q -x -k "tie_pixels or falls_back or refill" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/gputest.log
timeout 900 python tools/full_spp_parity.py > $OUT/full_spp_parity.txt 2> $OUT/full_spp_parity.err; echo "full spp exit $?"
cat $OUT/full_spp_parity.txt
