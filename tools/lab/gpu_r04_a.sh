#!/bin/bash
# round 4, first call: the pooled closest-hit kernel -- parity of every form, swap thresholds / stack depths / waves on the headline
# scene (64-layer passes), lane census, and the bench line with a step = one 64-spp frame
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04a
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refill or wide_forms or closest_hit or deterministic" > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/gputest.log
export RT_VARIANTS="$(cat tools/variants_pool_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_pool_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_pool_bistro64.txt | cut -c1-260
for m in 3 4; do
  RAYHIP_REFILL=$m timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_refill$m.json 2> $OUT/bench_refill$m.err; echo "bench mode $m exit $?"
  python3 -c "
import json; d=json.load(open('$OUT/bench_refill$m.json')); print('mode $m', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms/frame', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, d['config']['spp'], d['render_ms'], d.get('readback_ms'))"
done
du -sh $OUT
