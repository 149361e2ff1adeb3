"""Reduce rocprofv3 counter CSVs to per-kernel per-dispatch averages (runs on the GPU box right after profiling)."""
import csv, glob, os, sys
from collections import defaultdict
out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
dirs = set(glob.glob(os.path.join(out, f"{tag}_prof_*")) + glob.glob(os.path.join(out, f"{tag}[0-9]*")) + glob.glob(os.path.join(out, f"{tag}*")))
for d in sorted(x for x in dirs if os.path.isdir(x)):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?").split("(")[0][:60]
                c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
                acc[k][c] += v; cnt[k][c] += 1
        print("##", os.path.relpath(f, out))
        for k in sorted(acc):
            for c in sorted(acc[k]):
                print(f"{k:62s} {c:22s} dispatches {cnt[k][c]:6d} sum {acc[k][c]:.6g} avg/dispatch {acc[k][c]/max(cnt[k][c],1):.6g}")
