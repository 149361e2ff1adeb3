#!/bin/bash
# round 5, call J: the scene fuzz on the final tree -- against the host build (as round 4) and DIRECTLY against the live reference
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05final; mkdir -p $O
{ echo "# seeds 11000-11059 against the host build of the kernel sources, 12000-12059 and the two worst scenes of round 4's fuzz (7017, 10017 and their neighbours)"
  echo "# DIRECTLY against the live reference (RendererRef from oracle/_ref: render 4 spp + NLM filter on both sides); the final tree of round 5"
  timeout 900 python tools/gpu_fuzz.py 11000 60 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids"
  timeout 900 python tools/gpu_fuzz.py 12000 60 oracle 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids"
  timeout 600 python tools/gpu_fuzz.py 7010 12 oracle 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids"
  timeout 600 python tools/gpu_fuzz.py 10010 12 oracle 2>&1 | grep -v "^Extends\|^Spatial\|amdgpu.ids"
} > $O/gpu_fuzz.txt
cat $O/gpu_fuzz.txt
for k in 1 2 3; do python -m pytest tests/test_gpu_comm.py -q -m gpu 2>&1 | grep "passed\|failed"; done
