#!/bin/bash
# round 4, final measurements: the driver's bench command as the first process on the box (a step = one 64-spp frame), the full GPU suite,
# kernel stats + FETCH / WRITE / VALU / L2 counters of the same frame shape (per-kernel HBM table, k2_traffic.json, active lanes per kernel),
# the bench line again with the fresh table, the other workloads, shard emulation, two emulated ranks
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_first_process.json 2> $OUT/bench_first.err; echo "bench (first process) exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench_first_process.json')); print('first', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms/frame', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
timeout 900 python -m pytest tests -m gpu -q --durations=8 > $OUT/gputest_final.log 2>&1
echo "pytest exit $?"; grep "passed\|failed" $OUT/gputest_final.log | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bistro -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps20_warmup5.csv
python3 $REPO/tools/pass_timeline.py $OUT/stats > $OUT/pass_timeline.txt 2>&1
for pmc in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${tag} -o bistro -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_${tag}.log 2>&1
done
python3 $REPO/tools/k2_traffic.py $OUT/k2_traffic.json bistro 4 1 64 64 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/kernel_hbm_bench_64spp_frames.txt $OUT/pmc_SQ_INSTS_VALU > $OUT/k2_traffic.log 2>&1
head -1 $OUT/k2_traffic.log | cut -c1-700
cat $OUT/kernel_hbm_bench_64spp_frames.txt | cut -c1-170
python3 - $OUT <<'PY' > $OUT/lanes_l2_per_kernel.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
for d in ("pmc_SQ_INSTS_VALU", "pmc_TCC_HIT_sum"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
print("# every launch of `bench.py --steps 4 --warmup 1 --no-cpu-baseline` (64-spp frames + the instrumented passes behind them), per kernel:")
print("# vector instructions (wave level), the lanes they ran with (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU), L2 hit rate (TCC_HIT / (HIT + MISS))")
print(f"# {'kernel':60s} {'SQ_INSTS_VALU':>14s} {'lanes of 64':>12s} {'L2 hit':>8s}")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    a = acc[k]
    if a.get("SQ_INSTS_VALU", 0) < 1e6:
        continue
    hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
    print(f"  {k[:60]:60s} {a['SQ_INSTS_VALU']:14.4g} {a.get('SQ_THREAD_CYCLES_VALU', 0) / a['SQ_INSTS_VALU']:12.1f} {(hit / (hit + miss) if hit + miss else 0):8.2f}")
PY
head -16 $OUT/lanes_l2_per_kernel.txt | cut -c1-120
cd $REPO
# the bench line again, now that the traffic table of THESE kernels exists (roofline.traffic exact, not stale)
mkdir -p profiles/r04 && cp $OUT/k2_traffic.json profiles/r04/k2_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20.err; echo "bench (with table) exit $?"
python3 -c "
import json
d=json.load(open('$OUT/bench_steps20_warmup5.json')); r=d['roofline']; print('final', round(d['value'],1), 'frac', round(r['frac'],3), 'stale', r.get('traffic_is_stale'), 'valu', {k: (round(v,3) if isinstance(v,float) else v) for k,v in (r.get('valu_issue') or {}).items() if k in ('frac','frac_paired_model','active_lanes_of_64','mix_is_stale')}, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}))"
for w in bistro_tex sponza cornell principled bistro12m; do
  timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_${w}.json 2> $OUT/b.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_${w}.json')); print('$w', round(d['value'],1), 'Msamples/s', d['config'].get('spp'), 'spp')"
done
timeout 300 python tools/shard_emulation.py bistro 64 20 > $OUT/shard_emulation.txt 2>&1; tail -14 $OUT/shard_emulation.txt
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_2ranks_emulated.json 2> $OUT/b2.err; echo "2 ranks exit $?"
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
du -sh $OUT
