#!/bin/bash
# round 4: where does a rank of 8 lose against an eighth of the unsharded frame?  kernel stats of one rank's share at N = 1 and N = 8; smoke()
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04o
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
cd /tmp
for n in 1 8; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard$n -o shard -- python $REPO/tools/shard_profile.py $n 64 > $OUT/shard$n.log 2>&1
cp $(find $OUT/shard$n -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_shard$n.csv
python3 $REPO/tools/pass_timeline.py $OUT/shard$n > $OUT/pass_timeline_shard$n.txt 2>&1
done
python3 - $OUT <<'PY'
import csv, sys
out = sys.argv[1]
t = {}
for n in (1, 8):
    t[n] = {r['Name']: (float(r['TotalDurationNs']) / 1e6, int(r['Calls'])) for r in csv.DictReader(open(f'{out}/kernel_stats_shard{n}.csv'))}
tot1 = tot8 = 0.0
print(f"{'kernel':70s} {'N=1 ms':>9s} {'N=8 ms':>9s} {'x8 / N=1':>9s}")
for k, (ms1, c1) in sorted(t[1].items(), key=lambda kv: -kv[1][0])[:18]:
    ms8 = t[8].get(k, (0.0, 0))[0]
    if 'lbvh' in k or 'collapse' in k or 'fill_tri' in k:
        continue
    tot1 += ms1; tot8 += ms8
    print(f"{k[:70]:70s} {ms1:9.2f} {ms8:9.2f} {8 * ms8 / ms1 if ms1 else 0:9.2f}")
print('sum', round(tot1, 1), round(tot8, 1), round(8 * tot8 / tot1, 3))
PY
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
