#!/bin/bash
# round 3: the parity tests added after the final run (shade launch forms, sparse lights, triangle pitch)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03af
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shade_forms or sparse_lights or triangle_pitch" > $OUT/gputest_new.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/gputest_new.log
