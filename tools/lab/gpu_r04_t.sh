#!/bin/bash
# round 4: layered accumulation with the pixel's state in registers: parity of the batched passes, the bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04t
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_shade_kernel.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "golden or frame or principled or zoo or refract or shade or ior" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
for i in 1 2; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench$i.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench$i.json')); print(round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, d.get('parity'))"
done
cd /tmp
