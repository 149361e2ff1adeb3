#!/bin/bash
# round 4: random scenes on the device against the host build with the final kernels (flat any-hit kernel, light pick with lane refill,
# passes without the ior plane where nothing refracts): new seeds (final tree: 9000-9039, 10000-10039; earlier in the round 7000-7039, 8000-8039), the three fuzzers
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04fuzz
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 200 python tools/gpu_fuzz.py 9000 40 > $OUT/gpu_fuzz_a.txt 2>&1; echo "fuzz a exit $?"; tail -1 $OUT/gpu_fuzz_a.txt
timeout 200 python tools/gpu_fuzz.py 10000 40 > $OUT/gpu_fuzz_b.txt 2>&1; echo "fuzz b exit $?"; tail -1 $OUT/gpu_fuzz_b.txt
cat $OUT/gpu_fuzz_a.txt $OUT/gpu_fuzz_b.txt > $OUT/gpu_fuzz.txt
grep -c . $OUT/gpu_fuzz.txt
