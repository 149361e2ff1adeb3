#!/bin/bash
# round 3: is K2 bound by vector-ALU issue?  N extra full-rate instructions per node visit against the kernel's time
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03n
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python bench.py --no-cpu-baseline --steps 4 --warmup 0 > /dev/null 2>&1
export RT_VARIANTS="$(cat tools/variants_dummy_r03.json)"
RT_BATCH=20 timeout 900 python tools/variants.py run bistro 20 > $OUT/variants_dummy.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_dummy.txt | tail -8
