#!/bin/bash
# round 5, call B: chunks handed out dynamically (ChunkWalk with a work counter): exactness, then the headline, then a rank of 8
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sized_launches or batching or refill or flat_shadow or tile_sharding or frame_vs_reference or physical_sky or full_size" > $O/b_pytest.log 2>&1
tail -5 $O/b_pytest.log
for cfg in "default:RAYHIP_CENSUS=1" "no_overlap:RAYHIP_OVERLAP_SHADOW=0" "old:RAYHIP_OVERLAP_SHADOW=0 RAYHIP_CENSUS=0"; do
  name=${cfg%%:*}; env=${cfg#*:}
  env $env python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/b_bench_$name.json 2> $O/b_bench_$name.err
  python - <<PY
import json
d=json.load(open("$O/b_bench_$name.json"))
print("$name", round(d["value"],1), "Msamples/s", {k: round(v/1000,2) for k,v in d["stage_us_per_step"].items()})
PY
done
for cfg in "default:RAYHIP_CENSUS=1" "no_overlap:RAYHIP_OVERLAP_SHADOW=0"; do
  name=${cfg%%:*}; env=${cfg#*:}
  echo "== shard emulation $name"
  env $env timeout 600 python tools/shard_emulation.py bistro 64 20 2>&1 | grep -v amdgpu.ids | tee $O/b_shard_$name.txt
done
