#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02aa
mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_shade_kernel.py -m gpu -q -x > $OUT/gputest.log 2>&1; tail -2 $OUT/gputest.log | head -1
python bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for i in 1 2; do
python bench.py --steps 64 --warmup 64 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_us_per_step']; print(64, round(d['value'],1), 'shade', round(s['primary_shade']+s['secondary_shade']))"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $REPO/bench.py --steps 64 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:10]:
    n=r["Name"].split("(")[0].replace("void rt::","").replace("rt::","")
    print(n[:44].ljust(44), "calls", r["Calls"], "avg ms", round(float(r["AverageNs"])/1e6,3), r["Percentage"])
PY
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
