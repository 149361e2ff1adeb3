#!/bin/bash
# round 4: the bench line again (box-to-box spread of the secondary trace stage)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04u
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
for i in 1 2 3; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench$i.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench$i.json')); print(round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms', {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
