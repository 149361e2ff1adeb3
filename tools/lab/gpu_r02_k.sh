#!/bin/bash
# round 2, run K: textured variant of the headline scene (parity + rate); K2 launch-shape re-check after leaf refinement
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 900 python -m pytest "tests/test_gpu_baseline_configs.py::test_atrium_1080p_against_renderer_ref[bistro_tex]" -m gpu -q -x -s > $OUT/tex_parity.log 2>&1
echo "tex parity exit $?"; grep -E "spp|passed|failed|Error" $OUT/tex_parity.log | tail -8
timeout 600 python bench.py --workload bistro_tex --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_tex_64_64.json 2> $OUT/bench_tex.err
python3 -c "
import json; d=json.load(open('$OUT/bench_tex_64_64.json')); print('bistro_tex', round(d['value'],1), 'Msamples/s', d['stage_us_per_step'])"
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_64_64.json 2> $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench_64_64.json')); print('bistro', round(d['value'],1), 'Msamples/s', d['stage_us_per_step'])"
for v in "RAYHIP_REFILL=1" "RAYHIP_REFILL=1 RAYHIP_REFILL_MULT=2"; do
  env $v timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench_refill.json 2> $OUT/bench_refill.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_refill.json')); print('$v', round(d['value'],1), 'Msamples/s', d['stage_us_per_step'])"
done
export RT_VARIANTS='{"w6": [], "w5": ["-DRT_TRACE_MIN_WAVES=5"], "w4": ["-DRT_TRACE_MIN_WAVES=4"]}'
RT_BATCH=32 timeout 900 python tools/variants.py run bistro 32 2>&1 | grep -v "^  " | tee $OUT/variants_waves.txt
