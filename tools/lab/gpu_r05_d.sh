#!/bin/bash
# round 5, call D: the new GPU tests (stand-in RCCL transport with 2 and 3 ranks on one device, builder flags, sized launches) + the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rccl_processes.py tests/test_builder_flags.py -x -q -m gpu > $O/d_pytest_new.log 2>&1; tail -5 $O/d_pytest_new.log
timeout 1200 python -m pytest tests -x -q -m gpu > $O/d_pytest_gpu.log 2>&1; tail -4 $O/d_pytest_gpu.log
python bench.py --steps 6 --warmup 2 > $O/d_bench.json 2> $O/d_bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05/d_bench.json"))
print(round(d["value"],1), d["roofline"]["frac"], d["roofline"]["traversal"], d["parity"]["pass"], d["cpu_baseline"]["value"])
PY
