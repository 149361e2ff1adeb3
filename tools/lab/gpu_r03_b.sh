#!/bin/bash
# round 3, second GPU call: the 8-wide BLAS (rt_bvh8.h) -- variants against the 4-wide form, parity tests, upload trace, bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03b
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_bvh8_r03.json)"
RT_BATCH=32 timeout 900 python tools/variants.py run bistro 32 > $OUT/variants_bvh8.txt 2>&1; echo "variants exit $?"
grep -v "^  " $OUT/variants_bvh8.txt | tail -14
unset RT_VARIANTS
RAYHIP_TRACE_UPLOAD=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 exit $?"
grep "rayhip_scene_upload" $OUT/bench20.err | head -20
python3 -c "
import json; d=json.load(open('$OUT/bench20.json')); print('bench20', round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_step'].items()}, d['roofline']['algorithmic'])" || tail -5 $OUT/bench20.err
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_instance_update.py -m gpu -q -x --durations=5 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/gputest.log
