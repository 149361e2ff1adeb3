"""Do two wavefront passes that share one GPU finish sooner side by side than one after the other?  (tuning tool, round 5)

    python tools/lab/overlap_probe.py [workload=bistro]

Each rayhip context owns a non-blocking stream, so two contexts on one device are two independent in-order queues.  Cases, all the
same total work (the 64-spp headline frame, or rank 0's share of it at N = 8):
  serial     one context, one 64-layer pass                                   (what bench.py times)
  layers a+b two contexts, iterations 1..a on one and a+1..64 on the other, enqueued back to back, both drained
  tiles K    K contexts, each the 64 layers of its 1/K of the tiles
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401

import bench
from ray_amd import api, hip


def make_ctx(blob, W, H, shard=None):
    ctx = hip.Context(0)
    ctx.upload_static(api.pmj_table())
    ctx.resize(W, H)
    ctx.upload_scene_blob(blob)
    if shard:
        ctx.set_shard(bench.TILE, *shard)
    return ctx


def timed(ctxs, jobs, reps=3):
    """jobs: per context (first_iteration, count).  Returns the best wall time in ms of `reps` runs (after one set-up run)."""
    best = 1e9
    for rep in range(reps + 1):
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for c, (first, n) in zip(ctxs, jobs):
            c.render_batch(first, n)
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) * 1e3
        if rep:
            best = min(best, dt)
    return best


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "bistro"
    wl = bench.WORKLOADS[workload]
    W, H = wl["w"], wl["h"]
    blob, _ = bench.get_scene_blob(workload, wl, 0, 1, lambda: None)
    SPP = 64
    for world in (1, 8):
        shard = (world, 0) if world > 1 else None
        print(f"== {workload} {W}x{H}, {SPP} spp, rank 0 of {world}", flush=True)
        c = make_ctx(blob, W, H, shard)
        c.reserve_batch(SPP)
        t_serial = timed([c], [(1, SPP)])
        print(f"serial: one 64-layer pass            {t_serial:8.2f} ms", flush=True)
        t_two = timed([c], [(1, 32)]) + timed([c], [(33, 32)])
        print(f"serial: two 32-layer passes          {t_two:8.2f} ms", flush=True)
        c.close()
        for a in (32, 40, 48):
            ctxs = [make_ctx(blob, W, H, shard), make_ctx(blob, W, H, shard)]
            ctxs[0].reserve_batch(a), ctxs[1].reserve_batch(SPP - a)
            t = timed(ctxs, [(1, a), (a + 1, SPP - a)])
            print(f"side by side: layers {a:2d} + {SPP - a:2d}          {t:8.2f} ms  ({t_serial / t:5.3f} x serial)", flush=True)
            for x in ctxs:
                x.close()
        for parts in ((16, 16, 16, 16), (22, 21, 21)):
            ctxs = [make_ctx(blob, W, H, shard) for _ in parts]
            jobs, first = [], 1
            for x, n in zip(ctxs, parts):
                x.reserve_batch(n)
                jobs.append((first, n))
                first += n
            t = timed(ctxs, jobs)
            print(f"side by side: layers {'+'.join(map(str, parts)):14s}  {t:8.2f} ms  ({t_serial / t:5.3f} x serial)", flush=True)
            for x in ctxs:
                x.close()
        if world == 1:
            for K in (2, 3):
                ctxs = [make_ctx(blob, W, H, (K, k)) for k in range(K)]
                for x in ctxs:
                    x.reserve_batch(SPP)
                t = timed(ctxs, [(1, SPP)] * K)
                print(f"side by side: tiles 1/{K} x {K} contexts     {t:8.2f} ms  ({t_serial / t:5.3f} x serial)", flush=True)
                for x in ctxs:
                    x.close()


if __name__ == "__main__":
    main()
