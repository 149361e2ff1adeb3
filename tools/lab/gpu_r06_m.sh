#!/bin/bash
# round 6, call M: the shade form chosen per pass from the census -- every BASELINE workload, default settings; parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06m; rm -rf $O; mkdir -p $O
for w in cornell principled sponza bistro; do
  timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_${w}.json 2> $O/b.err
  python3 -c "
import json; d=json.load(open('$O/bench_${w}.json')); print('$w auto', round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})" | tee -a $O/forms_auto.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "shade_forms or sparse_lights or frame_vs_reference or batching or partition or sharding or lane_refill" > $O/parity.log 2>&1; echo "pytest exit $?"; tail -3 $O/parity.log
