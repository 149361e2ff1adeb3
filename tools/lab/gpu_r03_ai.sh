#!/bin/bash
# round 3: K4's stack in LDS -- the parity tests that reach the analytic-light kernels
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03ai
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
timeout 240 python -m pytest tests/test_gpu_parity.py tests/test_shade_kernel.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
