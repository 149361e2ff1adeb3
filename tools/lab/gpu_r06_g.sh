#!/bin/bash
# round 6, call G: regression run -- the driver's bench command, the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06g; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v) for k,v in d['stage_us_per_spp'].items()}, 'stage sum / step', round(d['stage_sum_over_step'],3), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/gputest.log 2>&1; echo "pytest exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/gputest.log | cut -c1-220 | tail -30
