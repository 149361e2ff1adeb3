#!/bin/bash
# round 2, run H: full GPU suite with the device-side table fill + instance update; upload timeline; bench at both shapes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02h
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/gputest.log
timeout 300 python -m pytest tests/test_gpu_instance_update.py -m gpu -q -x -s > $OUT/instance_update.log 2>&1
grep -E "rayhip_scene|update of|pixels differing|passed|failed" $OUT/instance_update.log | tail -40
RAYHIP_TRACE_UPLOAD=1 timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err
grep "rayhip_scene_upload" $OUT/bench_20_5.err | cut -c1-100
cat $OUT/bench_20_5.json
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_64_64.json 2> $OUT/bench_64_64.err
cat $OUT/bench_64_64.json
