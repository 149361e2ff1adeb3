#!/bin/bash
# round 4: passes over scenes without refractive surfaces leave the rays' ior plane alone: parity, then the bench line with and without
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04m
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ior_plane or golden or frame or deterministic or batching" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/gputest.log
for v in 0 1 0 1; do
if [ $v = 1 ]; then export RAYHIP_NO_PLAIN_IOR=1; else unset RAYHIP_NO_PLAIN_IOR; fi
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench$v.json 2> $OUT/bench$v.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench$v.json')); print('no_plain_ior=$v', round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
done
