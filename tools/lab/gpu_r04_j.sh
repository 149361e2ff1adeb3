#!/bin/bash
# round 4: the continuation class by class (shade_split bit 3): parity suite of the frames, then the bench line with and without it
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04j
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed" $OUT/gputest.log | tail -2
for split in 13 5 13; do
RAYHIP_SHADE_SPLIT=$split timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench$split.json 2> $OUT/bench$split.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench$split.json')); print($split, round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
done
