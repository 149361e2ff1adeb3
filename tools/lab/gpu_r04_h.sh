#!/bin/bash
# round 4: the physical sky on the device + the other new GPU tests of the round; the bench line (no regressions from the SKY template)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04h
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_build.py -m gpu -q -x -s -k "sky or tie_pixels or falls_back or refill or frame_vs_reference" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; grep -E "cornell_sky|passed|failed|Error" $OUT/gputest.log | cut -c1-300 | tail -8
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
