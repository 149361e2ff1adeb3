#!/bin/bash
# round 5, call K: bench.py's transport fall-back (N > 1 code path with one rank: the in-library communicator, then the same with an RCCL that
# cannot be loaded -> torch.distributed's gather), and BASELINE configs 2 and 5 at their stated 256 / 512 spp on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
RAY_AMD_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force_dist exit $?"
RAY_AMD_FORCE_DIST=1 RAYHIP_RCCL_LIB=libm.so.6 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_force_dist_no_rccl.json 2> $O/bench_force_dist_no_rccl.err; echo "no_rccl exit $?"
grep -h "rayhip_comm unavailable" $O/*.err
python - <<'PY'
import json
for n in ("bench_force_dist", "bench_force_dist_no_rccl"):
    try:
        d = json.loads(open(f"gpurun_out/r05k/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d.get("exchange_ms"), d.get("exchange"))
    except Exception as e:
        print(n, "unreadable", e)
PY
timeout 1000 python tools/full_spp_parity.py > $O/full_spp_parity.txt 2> $O/full_spp_parity.err; echo "full spp exit $?"
cat $O/full_spp_parity.txt
