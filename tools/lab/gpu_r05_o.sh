#!/bin/bash
# round 5, call O: pytest -m gpu on the tree as committed (with the material matrix in it) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05o; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest exit $?"; grep "passed\|failed" $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
