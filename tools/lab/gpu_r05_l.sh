#!/bin/bash
# round 5, call L: the last check of the tree as committed -- pytest -m gpu, smoke(), the driver's bench line, and 120 more fuzz scenes directly against the live reference
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; cut -c1-400 $O/bench_default.json
{ echo "# seeds 13000-13119 DIRECTLY against the live reference, the tree as committed at the end of round 5"
  timeout 900 python tools/gpu_fuzz.py 13000 120 oracle 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids"; } > $O/gpu_fuzz_more.txt
tail -5 $O/gpu_fuzz_more.txt
