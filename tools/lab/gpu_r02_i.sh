#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02i
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_instance_update.py -m gpu -q -x -s > $OUT/instance_update.log 2>&1
grep -E "rayhip_scene|update of|pixels differing|passed|failed|Error|assert" $OUT/instance_update.log | tail -60
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_gpu_baseline_configs.py > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/gputest.log
