#!/bin/bash
# round 6, call C: the shade stage without the point queue (pick first, surface + continuation fused, NEE over dense records): bit-identity first, then time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "shade_forms or sparse_lights or lane_refill or frame_vs_reference or batching" > $O/parity.log 2>&1; echo "pytest exit $?"; tail -3 $O/parity.log
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/build.err
RT_BATCH=64 RT_VARIANTS="$(cat tools/lab/variants_fused_r06.json)" timeout 900 python tools/variants.py run bistro 64 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $O/variants_fused.txt
cat $O/variants_fused.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$O/bench_fused.json')); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
