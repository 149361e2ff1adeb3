#!/bin/bash
# round 4: the whole GPU suite after the physical sky + the bench line twice (run-to-run spread of the stage times)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04i
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/gputest.log | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench$i.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench$i.json')); print(round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
done
