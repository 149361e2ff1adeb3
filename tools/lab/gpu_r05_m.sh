#!/bin/bash
# round 5, call M: the reference's own material test matrix on the device -- under pytest (96 x 96, 8 spp) and as the full table (256 x 256, the tests' own
# sample counts up to 64)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
timeout 900 python -m pytest tests/test_material_matrix.py -m gpu -q > $O/gputest_matrix.log 2>&1; echo "pytest exit $?"; grep -v "^Extends\|^Spatial" $O/gputest_matrix.log | tail -15
timeout 1200 python tools/material_matrix.py gpu 256 64 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids" > $O/material_matrix_gpu.txt; echo "table exit $?"
grep "^X \|^#" $O/material_matrix_gpu.txt
