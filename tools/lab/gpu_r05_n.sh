#!/bin/bash
# round 5, call N: the material matrix driven like the reference's harness (RendererHIP behind the Ray API), the 91st entry, and the table again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
timeout 900 python -m pytest tests/test_material_matrix.py -m gpu -q -s > $O/gputest_matrix.log 2>&1; echo "pytest exit $?"; grep -v "^Extends\|^Spatial" $O/gputest_matrix.log | tail -12
timeout 1200 python tools/material_matrix.py gpu 256 64 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids" > $O/material_matrix_gpu.txt; echo "table exit $?"
grep "^X \|^#" $O/material_matrix_gpu.txt
