#!/bin/bash
# round 3: where does the timed region's time outside the stages go?  bench three times in a row (first process on the box first)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03e
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20_$i.json 2> $OUT/bench20_$i.err; echo "bench20 exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench20_$i.json')); print('bench20 run $i', round(d['value'],1), 'Msamples/s step', round(d['ms_per_step'],3), 'render_ms', round(d['render_ms'],2), 'readback_ms', round(d['readback_ms'],2), 'stages', round(sum(d['stage_us_per_step'].values())*20/1e3,2))"
done
RAY_AMD_BENCH_REPEAT=2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20_rep.json 2> $OUT/bench20_rep.err; grep repeat $OUT/bench20_rep.err
