#!/bin/bash
# round 3: random scenes on the device with the final kernels against the host build (render + NLM filter)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03fuzz
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
timeout 150 python tools/gpu_fuzz.py 5000 40 > $OUT/gpu_fuzz_a.txt 2>&1; echo "fuzz a exit $?"; tail -1 $OUT/gpu_fuzz_a.txt
timeout 150 python tools/gpu_fuzz.py 6000 40 > $OUT/gpu_fuzz_b.txt 2>&1; echo "fuzz b exit $?"; tail -1 $OUT/gpu_fuzz_b.txt
