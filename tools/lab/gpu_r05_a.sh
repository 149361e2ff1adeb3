#!/bin/bash
# round 5, call A: does running two passes side by side on one GPU pay?  + this box's baseline line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r05/bench_base_a.json 2> gpurun_out/r05/bench_base_a.err
tail -c 600 gpurun_out/r05/bench_base_a.json | head -c 300; echo
timeout 600 python tools/lab/overlap_probe.py bistro > gpurun_out/r05/overlap_probe.txt 2>&1
cat gpurun_out/r05/overlap_probe.txt
