#!/usr/bin/env python3
"""rocprofv3 counter CSVs -> profiles/<round>/k2_traffic.json: HBM-side bytes per launch of the closest-hit kernel.

    python tools/k2_traffic.py <out.json> <workload> <steps> <warmup> <iterations_per_pass> <fetch_dir> <write_dir> [<stats_csv>]

fetch_dir / write_dir: output directories of two `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of
`bench.py --steps <steps> --warmup <warmup> --no-cpu-baseline` (separate passes: the two counters do not fit one).  Only
launches of the timed shape count: the warm-up passes (different iteration count per pass) and the instrumented kernels that
bench.py runs after the timed region are told apart by name / by being the largest launches.  Unit of both counters: KiB.
Appends / replaces the entry for this (workload, steps, iterations_per_pass)."""
import csv, glob, json, os, sys

KERNEL = "k_trace_closest<false, true, 6>"


def per_launch(d, counter):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"]) * 1024.0))
    vals.sort()
    return [v for _, v in vals]


def main():
    out, workload, steps, warmup, ipp, fetch_dir, write_dir = sys.argv[1:8]
    steps, warmup, ipp = int(steps), int(warmup), int(ipp)
    fetch, write = per_launch(fetch_dir, "FETCH_SIZE"), per_launch(write_dir, "WRITE_SIZE")
    # launches per pass: primary + one per bounce; the timed passes are the LAST ceil(steps / ipp) passes of the product kernel
    passes = -(-steps // ipp)
    warm_passes = -(-warmup // ipp) if warmup else 0
    per_pass = len(fetch) // max(passes + warm_passes, 1)
    take = per_pass * passes
    f, w = fetch[-take:], write[-take:]
    entry = {"workload": workload, "steps": steps, "warmup": warmup, "iterations_per_pass": ipp,
             "launches_sampled": len(f), "launches_per_pass": per_pass,
             "fetch_bytes_per_launch": sum(f) / max(len(f), 1), "write_bytes_per_launch": sum(w) / max(len(w), 1),
             "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) -- python bench.py --steps {steps} --warmup {warmup} --no-cpu-baseline"}
    # launch durations of the same launches, from the kernel trace the FETCH_SIZE pass wrote (under the profiler; bench.py
    # measures its own with HIP events in the unprofiled run)
    durs = []
    for f in glob.glob(os.path.join(fetch_dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get("Kernel_Name", ""):
                    durs.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
    durs = [d for _, d in sorted(durs)][-take:]
    if durs:
        entry["avg_launch_ms"] = sum(durs) / len(durs)
        entry["hbm_GBps_under_profiler"] = (entry["fetch_bytes_per_launch"] + entry["write_bytes_per_launch"]) / 1e9 / (entry["avg_launch_ms"] / 1e3)
    table = {"_comment": "HBM-side traffic of k_trace_closest<false,true,6> per launch of the timed passes; unit bytes (counters are KiB). "
                         "FETCH_SIZE on this access pattern (random 64-byte gathers) was calibrated at 0.96-0.98 x the missed bytes "
                         "(profiles/r01/gather_bench.txt), so no x2 correction is applied; Infinity-Cache hits are included in both counters "
                         "(MI355X_MICROARCH.md), i.e. this is an upper bound of what reached HBM.", "runs": []}
    if os.path.exists(out):
        with open(out) as fh:
            table = json.load(fh)
    table["runs"] = [e for e in table.get("runs", []) if not (e["workload"] == workload and e["steps"] == steps and e["iterations_per_pass"] == ipp)]
    table["runs"].append(entry)
    with open(out, "w") as fh:
        json.dump(table, fh, indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
