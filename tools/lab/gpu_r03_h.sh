#!/bin/bash
# round 3: does the host-side layout pass (bvh_layout.h: 186 ms of the upload) still buy anything with the 4-wide BLAS?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03h
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_layout_r03.json)"
for wl in bistro sponza; do
RT_BATCH=32 timeout 900 python tools/variants.py run $wl 32 > $OUT/variants_layout_$wl.txt 2>&1
grep -v "^  " $OUT/variants_layout_$wl.txt | tail -2
done
unset RT_VARIANTS
RAYHIP_TRACE_UPLOAD=1 timeout 300 python - <<'PY' 2>&1 | grep rayhip_scene_upload
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from ray_amd import hip, api
blob, info = bench.get_scene_blob("bistro", bench.WORKLOADS["bistro"], 0, 1, lambda: None)
ctx = hip.Context(0)
ctx.upload_static(api.pmj_table()); ctx.resize(64, 64)
ctx.upload_scene_blob(blob)
PY
