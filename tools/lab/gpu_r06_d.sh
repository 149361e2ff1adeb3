#!/bin/bash
# round 6, call D: per-launch times of the fused form (3 waves) against the three-kernel form, first two bounces of a 64-layer pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; rm -rf $O; mkdir -p $O
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/build.err
export RT_BATCH=64 RT_VARIANTS="$(cat $1)"
shift
for v in "$@"; do
  rm -rf /tmp/prof_$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -o v -- python $GRAFT_REPO_ROOT/tools/variants.py run1 $v bistro 64 > $O/run_$v.log 2>&1)
  grep Msamples $O/run_$v.log | cut -c1-200
  python3 - /tmp/prof_$v $v <<'PY' >> $O/launches.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
gens = [i for i, r in enumerate(rows) if "k_raygen" in r[2] and r[1] - r[0] > 1e6]
i0 = gens[1] if len(gens) > 1 else gens[0]
i1 = next(i for i in range(i0, len(rows)) if "k_accumulate" in rows[i][2])
seen = {}; tot = {}
print("==", sys.argv[2])
for s, e, k in rows[i0:i1 + 1]:
    k = k.split("(")[0].replace("void ", "").replace("rt::", "")
    n = seen.get(k, 0); seen[k] = n + 1; tot[k] = tot.get(k, 0) + (e - s)
    if n < 2 and "shade::" in k and (e - s) > 2e5:
        print(f"  {k[:50]:50s} launch {n}: {(e - s) / 1e6:7.3f} ms")
print("  -- whole pass, by kernel:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    if v > 2e5: print(f"  {k[:50]:50s} {seen[k]:3d} launches {v / 1e6:8.3f} ms")
print(f"  pass wall {(rows[i1][1] - rows[i0][0]) / 1e6:.3f} ms; shade kernels {sum(v for k, v in tot.items() if 'shade::' in k) / 1e6:.3f} ms")
PY
done
cat $O/launches.txt
