#!/bin/bash
# round 4: flat any-hit kernel parity; per-kernel times of the shade stage with / without the class-by-class continuation, K3 both forms
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04k
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
echo skip-tests
python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
cd /tmp
for cfg in "5 0" "13 1"; do
set -- $cfg
RAYHIP_SHADE_SPLIT=$1 RAYHIP_SHADOW_REFILL=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1_$2 -o b -- python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/prof_$1_$2.log 2>&1
f=$(find $OUT/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
echo "== split $1 shadow_refill $2"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.2f' % (float(r['TotalDurationNs']) / 1e6)).rjust(9), 'ms')
PY
cp $f $OUT/kernel_stats_$1_$2.csv
done
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
