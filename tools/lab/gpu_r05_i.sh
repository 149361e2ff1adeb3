#!/bin/bash
# round 5, call I: after the last test fix and the regenerated VALU mix -- the comm tests 5 x, the whole suite once more, the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05final; mkdir -p $O
for k in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_comm.py -q -m gpu 2>&1 | tail -1; done > $O/gputest_comm_after_fix.log; cat $O/gputest_comm_after_fix.log
timeout 1200 python -m pytest tests -m gpu -q > $O/gputest_final2.log 2>&1; grep "passed\|failed" $O/gputest_final2.log | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20.err
python3 -c "
import json
d=json.load(open('$O/bench_steps20_warmup5.json')); r=d['roofline']; print('final', round(d['value'],1), 'K2 frac', round(r['frac'],3), 'stale', r.get('traffic_is_stale'), 'trav', r['traversal'].get('frac'), 'valu', (r.get('valu_issue') or {}).get('frac'), (r.get('valu_issue') or {}).get('mix_is_stale'))"
