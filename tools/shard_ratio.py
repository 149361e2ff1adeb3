#!/usr/bin/env python3
"""Per-kernel strong-scaling ratio of a rank of N from two `rocprofv3 --kernel-trace --stats` kernel_stats.csv files of tools/shard_profile.py
(world 1 and world N): N x (time at N) / (time at 1) per kernel.

    python tools/shard_ratio.py <stats_w1.csv> <stats_wN.csv> [N=8] > profiles/<round>/shard_kernel_ratio.txt
"""
import csv
import re
import sys


def load(path):
    d = {}
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\s+", " ", r["Name"].split("(")[0].replace("void ", "").replace("rt::", ""))
        d[k] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return d


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    print(f"# one rank's share of the headline frame (64 spp, three frames) under rocprofv3 --kernel-trace --stats: the unsharded frame (N = 1) against rank 0 of {n}")
    print(f"# (tools/shard_profile.py, tools/shard_ratio.py).  Last column: {n} x the N = {n} time / the N = 1 time -- 1.00 would be perfect strong scaling of that kernel.")
    print(f"# {'kernel':58s} {'calls':>6s} {'N=1 ms':>9s} {f'N={n} ms':>9s} {f'x{n} / N=1':>9s}")
    s1 = s8 = 0.0
    for k, (c, ms) in sorted(a.items(), key=lambda kv: -kv[1][1]):
        if k not in b or ms <= 0.3 or k.startswith("__amd") or k.startswith("rayhip_") or "k_fill" in k or "queue_totals" in k:
            continue  # (upload-time builders and fills are not part of a pass)
        note = ""
        if "k_trace_shadow" in k:
            note = "   <- elapsed on the low-priority second stream, next to K2: not its cost (not in the sum)"
        else:
            s1, s8 = s1 + ms, s8 + b[k][1]
        print(f"  {k[:58]:58s} {c:6d} {ms:9.2f} {b[k][1]:9.2f} {n * b[k][1] / ms:9.2f}{note}")
    print(f"  {'sum without K3':58s} {'':6s} {s1:9.2f} {s8:9.2f} {n * s8 / s1:9.2f}")


if __name__ == "__main__":
    main()
