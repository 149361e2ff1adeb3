#!/usr/bin/env python3
"""Per-launch timeline of the LAST product pass in a rocprofv3 --kernel-trace run of bench.py: kernel, start offset, duration,
gap to the previous launch.  Shows what the late (thin) bounces of a wavefront pass cost.

    python tools/pass_timeline.py <rocprofv3 output dir> [--pass-index -1]
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("rt::", "")
    return re.sub(r"\s+", " ", name).strip()


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_raygen")]
    counted = [i for i, r in enumerate(rows) if r[2].startswith("k_trace_closest<true")]
    # the last pass that is not an instrumented (counting) one: a pass's ray generation comes before its first traversal launch,
    # so the counting passes start with the last k_raygen in front of the first counting kernel
    first_counting = max(i for i in starts if i < min(counted)) if counted else len(rows)
    last = max(i for i in starts if i < first_counting)
    end = next(i for i in range(last, len(rows)) if rows[i][2].startswith("k_accumulate"))
    t0 = rows[last][0]
    prev_end = t0
    total_busy = 0
    print(f"{'kernel':44s} {'start ms':>9s} {'dur ms':>8s} {'gap us':>7s}")
    for s, e, k in rows[last:end + 1]:
        print(f"{k[:44]:44s} {(s - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {(s - prev_end) / 1e3:7.1f}")
        total_busy += e - s
        prev_end = e
    print(f"pass: {(rows[end][1] - t0) / 1e6:.3f} ms wall, {total_busy / 1e6:.3f} ms in kernels, {len(rows[last:end + 1])} launches")


if __name__ == "__main__":
    main()
