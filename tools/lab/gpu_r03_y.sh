#!/bin/bash
# round 3: pass size against stage times with the new kernels
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03y
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
for K in 8 20 32 48 64; do
  timeout 600 python bench.py --steps $K --warmup $K --no-cpu-baseline > $OUT/bench$K.json 2> $OUT/bench$K.err
  python3 -c "
import json; d=json.load(open('$OUT/bench$K.json')); print($K, round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), d['config']['iterations_per_pass'], {k: round(v) for k,v in d['stage_us_per_step'].items()})" || tail -3 $OUT/bench$K.err
done
