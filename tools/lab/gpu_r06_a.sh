#!/bin/bash
# round 6, call A: the shade stage compiled with -ffp-contract=fast against the STATED bar: time, then the whole GPU suite on that build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06a; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; echo "bench base exit $?"
RT_BATCH=64 RT_VARIANTS="$(cat tools/lab/variants_contract_r06.json)" timeout 900 python tools/variants.py run bistro 64 2>&1 | grep -v amdgpu.ids > $O/variants_contract.txt
cat $O/variants_contract.txt | cut -c1-260
cp ray_amd/csrc/_build/librayhip.so /tmp/librayhip_base.so
cp ray_amd/csrc/_build/variants/shade_contract/librayhip.so ray_amd/csrc/_build/librayhip.so
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_contract.json 2> $O/bench_contract.err; echo "bench contract exit $?"
python3 - <<PY
import json
for n in ("base","contract"):
    d=json.load(open("$O/bench_%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_step"],2), {k: round(v) for k,v in d["stage_us_per_spp"].items()}, d.get("parity"))
PY
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest_contract.log 2>&1; echo "pytest exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/gputest_contract.log | cut -c1-220 | tail -60
