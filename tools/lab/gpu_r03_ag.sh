#!/bin/bash
# round 3: the driver's round-end sequence in small: smoke(), then the bench command
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03ag
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['metric'], round(d['value'],1), d['unit'], d['n_gpus'], d['steps'], d['warmup'], 'frac', round(d['roofline']['frac'],3), 'stale', d['roofline'].get('traffic_is_stale'), 'cpu', round(d['cpu_baseline']['value'],2), d['cpu_baseline']['cores'])"
