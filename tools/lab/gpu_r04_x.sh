#!/bin/bash
# round 4: the driver's N > 1 command line, as the driver launches it (torch.distributed.run from outside), on a 1-GPU box: the ranks share
# device 0 and exchange over gloo (emulated_ranks) -- a plumbing check of the launch path with the 64-spp-frame step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04x
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_torchrun_2ranks.json 2> $OUT/bench_torchrun_2ranks.err; echo "torchrun 2 ranks exit $?"
tail -c 1500 $OUT/bench_torchrun_2ranks.json | python3 -c "
import sys, json
lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
print(len(lines), 'JSON line(s)')
d=json.loads(lines[-1]) if lines else {}
print({k: d.get(k) for k in ('value','n_gpus','steps','warmup','ms_per_step','scaling','emulated_ranks','exchange_ms','render_ms')})" 2>&1 | tail -3
grep -c "RCCL communicator up" $OUT/bench_torchrun_2ranks.err
