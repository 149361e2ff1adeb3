#!/usr/bin/env python3
"""rocprofv3 counter CSVs -> profiles/<round>/k2_traffic.json (HBM-side bytes per launch of the closest-hit kernel) and a
per-kernel HBM table of the whole timed region.

    python tools/k2_traffic.py <out.json> <workload> <steps> <warmup> <spp> <iterations_per_pass> <fetch_dir> <write_dir> [<table.txt> [<valu_dir>]]

fetch_dir / write_dir: output directories of two `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of
`bench.py --steps <steps> --warmup <warmup> --spp <spp> --no-cpu-baseline` (separate passes: the two counters do not fit one).
A step is one frame of <spp> iterations (bench.py, round 4), i.e. ceil(spp / iterations_per_pass) passes.

The closest-hit kernel K2 has two forms (rayhip.hip): k_trace_closest<false,true,N> for the primary rays and
k_trace_closest_refill / k_trace_closest_pool for the secondary bounces; "a K2 launch" is a launch of any.  Which launches belong to
the timed region: bench.py runs, in this order, one priming frame of the timed shape, the warm-up frames, the timed frames and then
the instrumented passes (other kernels: k_trace_closest<true,...>) and the single-iteration parity render; every product pass has the
same number of K2 launches (primary + one per bounce), so the timed ones are the LAST steps x passes-per-frame x launches_per_pass
product launches before the first instrumented one.
Unit of both counters: KiB.  Appends / replaces the entry for this (workload, spp, iterations_per_pass)."""
import csv, glob, json, os, re, sys

K2_FORMS = ("k_trace_closest<false, 8", "k_trace_closest<false, 4", "k_trace_closest<false, true", "k_trace_closest_refill", "k_trace_closest_pool")
K3_FORMS = ("k_trace_shadow_refill", "k_trace_shadow<false")  # the any-hit kernel: SURVEY 8d's "traversal" is K2 + K3


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("rt::", "")
    return re.sub(r"\s+", " ", name).strip()


def counters(d, counter):
    """[(dispatch id, kernel, bytes)] in dispatch order"""
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    vals.append((int(row.get("Dispatch_Id", 0)), short(row.get("Kernel_Name", "")), float(row["Counter_Value"]) * 1024.0))
    vals.sort()
    return vals


def durations(d):
    """[(start, kernel, ms)] in start order"""
    out = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                out.append((int(row["Start_Timestamp"]), short(row.get("Kernel_Name", "")), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
    out.sort()
    return out


def is_k2(kernel):
    return any(kernel.startswith(k) for k in K2_FORMS)


def is_k3(kernel):
    return any(kernel.startswith(k) for k in K3_FORMS)


def main():
    out, workload, steps, warmup, spp, ipp, fetch_dir, write_dir = sys.argv[1:9]
    table_path = sys.argv[9] if len(sys.argv) > 9 else None
    steps, warmup, spp, ipp = int(steps), int(warmup), int(spp), int(ipp)
    fetch_all, write_all, dur_all = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE"), durations(fetch_dir)

    def product_part(rows, which=is_k2):
        """the K2 (K3) launches in front of the first instrumented kernel (what follows it -- the counting passes, a parity render -- is not the timed region)"""
        cut = next((i for i, (_, k, _) in enumerate(rows) if k.startswith("k_trace_closest<true")), len(rows))
        return [r for r in rows[:cut] if which(r[1])]

    fetch, write, durs = product_part(fetch_all), product_part(write_all), product_part(dur_all)
    per_frame = -(-spp // ipp)
    passes = steps * per_frame
    warm_passes = warmup * per_frame
    per_pass = len(fetch) // max(passes + warm_passes + per_frame, 1)  # (+ the priming frame)
    take = per_pass * passes
    f, w, dd = fetch[-take:], write[-take:], durs[-take:]
    fsum, wsum = sum(v for _, _, v in f), sum(v for _, _, v in w)
    entry = {"workload": workload, "steps": steps, "warmup": warmup, "spp": spp, "iterations_per_pass": ipp,
             "kernels": sorted({k for _, k, _ in f}),
             "launches_sampled": len(f), "launches_per_pass": per_pass,
             "fetch_bytes_per_launch": fsum / max(len(f), 1), "write_bytes_per_launch": wsum / max(len(w), 1),
             "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) -- python bench.py --steps {steps} --warmup {warmup} --spp {spp} --no-cpu-baseline"}
    # optional third profile, `--pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU`: vector instructions per launch and the lanes they ran with --
    # the kernel is bound by vector-ALU issue, this is its other roofline
    valu_dir = sys.argv[10] if len(sys.argv) > 10 else None
    if valu_dir:
        insts = [(i, k, v / 1024.0) for i, k, v in product_part(counters(valu_dir, "SQ_INSTS_VALU"))][-take:]
        lanes = [(i, k, v / 1024.0) for i, k, v in product_part(counters(valu_dir, "SQ_THREAD_CYCLES_VALU"))][-take:]
        if insts:
            entry["valu_wave_instructions_per_launch"] = sum(v for _, _, v in insts) / len(insts)
            if lanes:
                entry["valu_active_lanes"] = sum(v for _, _, v in lanes) / max(sum(v for _, _, v in insts), 1.0)
    # the any-hit kernel K3 of the same passes (round 5: bench.py's roofline block covers the traversal, K2 + K3)
    f3, w3, d3 = product_part(fetch_all, is_k3), product_part(write_all, is_k3), product_part(dur_all, is_k3)
    per_pass3 = len(f3) // max(passes + warm_passes + per_frame, 1)
    take3 = per_pass3 * passes
    if take3:
        f3, w3, d3 = f3[-take3:], w3[-take3:], d3[-take3:]
        entry["shadow"] = {"kernels": sorted({k for _, k, _ in f3}), "launches_sampled": len(f3), "launches_per_pass": per_pass3,
                           "fetch_bytes_per_launch": sum(v for _, _, v in f3) / len(f3), "write_bytes_per_launch": sum(v for _, _, v in w3) / max(len(w3), 1),
                           "avg_launch_ms": sum(v for _, _, v in d3) / max(len(d3), 1)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    entry["csrc_hash"] = bench.csrc_hash()  # the kernels this profile belongs to (bench.py marks it stale for any other)
    if dd:
        entry["avg_launch_ms"] = sum(v for _, _, v in dd) / len(dd)
        entry["hbm_GBps_under_profiler"] = (entry["fetch_bytes_per_launch"] + entry["write_bytes_per_launch"]) / 1e9 / (entry["avg_launch_ms"] / 1e3)
    table = {"_comment": "HBM-side traffic of the closest-hit kernel (k_trace_closest<false,true,N> for primary rays, k_trace_closest_refill for "
                         "the secondary bounces) per launch of the timed passes; unit bytes (counters are KiB). "
                         "FETCH_SIZE on this access pattern (random 64-byte gathers) was calibrated at 0.96-0.98 x the missed bytes "
                         "(profiles/r01/gather_bench.txt), so no x2 correction is applied; Infinity-Cache hits are included in both counters "
                         "(MI355X_MICROARCH.md), i.e. this is an upper bound of what reached HBM.", "runs": []}
    if os.path.exists(out):
        with open(out) as fh:
            table = json.load(fh)
        table["_comment"] = table.get("_comment", "")
    table["runs"] = [e for e in table.get("runs", []) if not (e["workload"] == workload and e.get("spp", e["steps"]) == spp and e["iterations_per_pass"] == ipp)]
    table["runs"].append(entry)
    with open(out, "w") as fh:
        json.dump(table, fh, indent=1)
    print(json.dumps(entry))

    if table_path:
        # every kernel of the timed region: the dispatches after the first timed K2 launch and before the first instrumented kernel
        first_id = f[0][0] if f else 0
        # (the pass starts with its ray generation, just before the first K2 launch)
        gen = [i for i, k, _ in fetch_all if k.startswith("k_raygen") and i < first_id]
        first_id = max(gen) if gen else first_id
        last_id = max(i for i, _, _ in f) if f else 0
        # (the stages that follow the last K2 launch of the pass -- shade, shadow, accumulate -- belong to it too: take every
        # product dispatch up to the first instrumented / counting kernel after last_id)
        end_id = None
        for i, k, _ in fetch_all:
            if i > last_id and k.startswith("k_trace_closest<true"):
                end_id = i
                break
        fb, wb, n = {}, {}, {}
        for i, k, v in fetch_all:
            if i >= first_id and (end_id is None or i < end_id):
                fb[k] = fb.get(k, 0.0) + v
                n[k] = n.get(k, 0) + 1
        for i, k, v in write_all:
            if i >= first_id and (end_id is None or i < end_id):
                wb[k] = wb.get(k, 0.0) + v
        # durations of the same kernels: the last n[k] launches of each kernel before the instrumented ones cannot be told
        # apart by id in the trace, so take time per kernel over the same count of launches counted back from the end of the
        # product launches
        t_first = dd[0][0] if dd else 0
        gen_t = [t for t, k, _ in dur_all if k.startswith("k_raygen") and t < t_first]
        t_first = max(gen_t) if gen_t else t_first
        t_end = None
        for t, k, _ in dur_all:
            if dd and t > dd[-1][0] and k.startswith("k_trace_closest<true"):
                t_end = t
                break
        ms = {}
        for t, k, v in dur_all:
            if t >= t_first and (t_end is None or t < t_end):
                ms[k] = ms.get(k, 0.0) + v
        total_ms = sum(ms.values())
        with open(table_path, "w") as fh:
            fh.write(f"# per-kernel HBM-side traffic of the timed passes: bench.py --workload {workload} --steps {steps} --warmup {warmup} --spp {spp}\n"
                     f"# (rocprofv3 --kernel-trace --pmc FETCH_SIZE, second run --pmc WRITE_SIZE; times from the FETCH_SIZE run, i.e. under the profiler)\n"
                     f"# {'kernel':58s} {'launches':>8s} {'ms':>9s} {'% time':>7s} {'fetch GB':>9s} {'write GB':>9s} {'TB/s':>6s} {'% of 8 TB/s':>11s}\n")
            for k in sorted(ms, key=lambda k: -ms[k]):
                gb_f, gb_w = fb.get(k, 0.0) / 1e9, wb.get(k, 0.0) / 1e9
                rate = (gb_f + gb_w) / ms[k] if ms[k] > 0 else 0.0  # GB / ms = TB/s
                fh.write(f"  {k[:58]:58s} {n.get(k, 0):8d} {ms[k]:9.2f} {100 * ms[k] / total_ms:7.1f} {gb_f:9.2f} {gb_w:9.2f} {rate:6.2f} {100 * rate / 8.0:11.1f}\n")
            tot_gb = (sum(fb.values()) + sum(wb.values())) / 1e9
            fh.write(f"  {'all kernels':58s} {sum(n.values()):8d} {total_ms:9.2f} {100.0:7.1f} {sum(fb.values()) / 1e9:9.2f} {sum(wb.values()) / 1e9:9.2f} "
                     f"{tot_gb / total_ms if total_ms else 0:6.2f} {100 * tot_gb / total_ms / 8.0 if total_ms else 0:11.1f}\n")
        print(open(table_path).read())


if __name__ == "__main__":
    main()
