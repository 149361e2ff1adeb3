#!/bin/bash
# round 3, first GPU call: the state before the kernel work -- new tests (64-spp headline parity, device builder, packed-tile
# exchange), bench.py as a plain process with 2 ranks on a 1-GPU box (emulated), the one-rank RCCL path, the driver's command
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03a
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 exit $?"
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20_2ranks.json 2> $OUT/bench20_2ranks.err; echo "bench 2 ranks exit $?"
RAY_AMD_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20_forcedist.json 2> $OUT/bench20_forcedist.err; echo "bench forcedist exit $?"
for f in bench20 bench20_2ranks bench20_forcedist; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value'],1), 'Msamples/s n_gpus', d['n_gpus'], 'exchange_ms', d.get('exchange_ms'), 'emulated', d.get('emulated_ranks'), 'stale', d['roofline'].get('traffic_is_stale'), {k: round(v) for k,v in d['stage_us_per_step'].items()})" || tail -5 $OUT/$f.err; done
timeout 1500 python -m pytest tests/test_gpu_bvh_build.py tests/test_gpu_comm.py tests/test_gpu_baseline_configs.py -m gpu -q -x --durations=8 > $OUT/gputest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/gputest.log
