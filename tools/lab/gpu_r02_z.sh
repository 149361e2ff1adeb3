#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02z
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
export RT_VARIANTS="$(cat tools/variants_shade2_r02.json)"
RT_BATCH=64 timeout 1200 python tools/variants.py run bistro 64 2>&1 | grep -v "^  " | tee $OUT/variants_shade2.txt
