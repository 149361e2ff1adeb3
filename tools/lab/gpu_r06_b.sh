#!/bin/bash
# round 6, call B: what the classes of the continuation cost (bounce 0 of a 64-layer pass: the same points in every variant): the kernel with the
# microfacet draws dropped / with the diffuse draw dropped (wrong images, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; rm -rf $O; mkdir -p $O
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/build.err
export RT_BATCH=64 RT_VARIANTS="$(cat tools/lab/variants_classcost_r06.json)"
for v in base drop_heavy drop_light; do
  rm -rf /tmp/prof_$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -o v -- python $GRAFT_REPO_ROOT/tools/variants.py run1 $v bistro 64 > $O/run_$v.log 2>&1)
  python3 - /tmp/prof_$v $v <<'PY' >> $O/classcost.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# passes of 64 layers: the k_raygen launches that take > 1 ms
gens = [i for i, r in enumerate(rows) if "k_raygen" in r[2] and r[1] - r[0] > 1e6]
i0 = gens[1] if len(gens) > 1 else gens[0]
seen = {}
print("==", sys.argv[2])
for s, e, k in rows[i0:i0 + 40]:
    k = k.split("(")[0].replace("void ", "").replace("rt::", "")
    n = seen.get(k, 0); seen[k] = n + 1
    if n < 2 and ("k_scatter" in k or "k_surface" in k or "k_light_pick" in k):
        print(f"  {k[:50]:50s} launch {n}: {(e - s) / 1e6:7.3f} ms")
PY
done
cat $O/classcost.txt
