#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_refill_r02.json)"
RT_BATCH=64 timeout 1200 python tools/variants.py run bistro 64 2>&1 | grep -v "^  " | tee $OUT/variants_refill.txt
