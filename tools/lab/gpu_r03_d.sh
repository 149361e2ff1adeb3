#!/bin/bash
# round 3, fourth GPU call: the UNet denoiser on the matrix cores (parity per pass, timing), the new defaults, bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03d
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -s > $OUT/gputest_unet.log 2>&1
echo "pytest unet exit $?"; grep -E "worst|UNet|passed|failed|Error|assert" $OUT/gputest_unet.log | head -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 exit $?"
timeout 600 python bench.py --steps 64 --warmup 64 --no-cpu-baseline > $OUT/bench64.json 2> $OUT/bench64.err; echo "bench64 exit $?"
for f in bench20 bench64; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), {k: round(v) for k,v in d['stage_us_per_step'].items()})" || tail -5 $OUT/$f.err; done
