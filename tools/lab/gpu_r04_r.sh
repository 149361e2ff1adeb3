#!/bin/bash
# round 4: four samples of a pixel in one primary wavefront (RAYHIP_RAYGEN_SAMPLES=4): parity, bench A/B
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04r
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "four_samples" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
for v in 1 4 16 64 1 16; do
RAYHIP_RAYGEN_SAMPLES=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench$v.json 2> $OUT/bench$v.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench$v.json')); print('samples per wave $v:', round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
done
