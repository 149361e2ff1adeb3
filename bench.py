#!/usr/bin/env python3
"""bench.py -- Msamples/s of the HIP path-tracer core loop on N MI355X (BASELINE.json metric), one JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bistro|sponza|cornell|principled]

A "step" is ONE RenderScene iteration (one sample per pixel over the whole frame: ray gen -> trace -> shade ->
shadow -> [trace -> shade -> shadow] x bounces -> accumulate).  K steps = K spp.  Msamples/s = W*H*K / time.

Workloads (BASELINE.json configs; no real Sponza/Bistro asset exists offline, see ray_amd/scenes.py):
  bistro      default.  synthetic atrium, ~3.0 M triangles, 1920x1080   (config 4: the scene the metric is quoted on)
  sponza      synthetic atrium, ~0.25 M triangles, 1920x1080            (config 3)
  cornell     samples/00_basic Cornell box, 1024x1024                   (config 2)
  principled  samples/03_principled, 2048x2048                          (config 5)

Multi-GPU (N > 1, launched by torch.distributed.run, one process per GPU): the scene is replicated, the frame's
64x64 tiles are dealt round-robin to the ranks (rayhip_set_shard), every rank renders all K samples of ITS tiles,
and ONE RCCL reduce (torch.distributed, backend nccl) of the raw fp32 framebuffer assembles the frame on rank 0
inside the timed region.  Total work is fixed -> "scaling": "strong".

Timed region: barrier + stream sync | K iterations (+ the reduce at N>1) | stream sync + barrier; max over ranks.
Inputs (scene, PMJ table) are resident in HBM before the region starts; nothing is copied to the host inside it.

roofline: for the dominant kernel k_trace_closest (BVH2 closest-hit traversal).  Its launches are bracketed by
HIP events on the context stream during the timed region (RAYHIP_FLAG_TIME_STAGES, no synchronisation); the
ALGORITHMIC bytes are the SURVEY.md section 8(d) formula  72+20+4 + 64*nodes + 48*tris + 144*instances  per ray,
with the visit counts taken from the instrumented kernel variant on iterations of the same workload right after
the timed region (counts per iteration are averaged over up to 4 iterations and scaled to K).
cpu_baseline: the reference's own AVX2 backend (oracle/_ref, kind "reference") on all host cores with the
documented tile/thread pattern, on a bounded number of spp of the same scene and resolution (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    "bistro": dict(kind="atrium", detail=4.3, w=1920, h=1080, label="synthetic Bistro-class atrium"),
    "sponza": dict(kind="atrium", detail=0.36, w=1920, h=1080, label="synthetic Sponza-class atrium"),
    "cornell": dict(kind="cornell_basic", w=1024, h=1024, label="samples/00_basic Cornell box"),
    "principled": dict(kind="cornell_principled", w=2048, h=2048, label="samples/03_principled Cornell box"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
TILE = 64


def build_scene(scene, wl):
    from ray_amd import scenes
    if wl["kind"] == "atrium":
        return scenes.atrium(scene, wl["detail"])
    scenes.SCENES[wl["kind"]](scene)
    return scene.triangle_count()


def get_scene_blob(name, wl, rank, world, barrier):
    """rank 0 builds (reference host-side SAH BVH + light tree through SceneHIP) and caches; the others read."""
    from ray_amd import api
    cache_dir = os.environ.get("RAY_AMD_CACHE", "/tmp/ray_amd_cache")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"{name}_{wl.get('detail', 0)}.rayscene")
    meta = path + ".json"
    t0 = time.time()
    if rank == 0 and not (os.path.exists(path) and os.path.exists(meta)):
        s = api.CreateSceneHIP()
        ntris = build_scene(s, wl)
        blob = api.export_scene_blob(s)
        with open(path + ".tmp", "wb") as f:
            f.write(blob)
        os.replace(path + ".tmp", path)
        with open(meta, "w") as f:
            json.dump({"tris": int(ntris), "bvh_tris": int(s.triangle_count()), "nodes": int(s.node_count())}, f)
    barrier()
    with open(path, "rb") as f:
        blob = f.read()
    with open(meta) as f:
        info = json.load(f)
    info["build_s"] = time.time() - t0
    return blob, info


def measured_traffic(workload, batch):
    """HBM bytes per K2 launch from the committed rocprofv3 PMC passes (profiles/r01/k2_traffic.json: FETCH_SIZE + WRITE_SIZE,
    collected in separate runs of this command under the profiler; scaled by the ray count if this run puts a different
    number of iterations into a pass than the profiled one); None for workloads that were not profiled"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "k2_traffic.json")) as f:
            t = json.load(f).get(workload)
        if t is None:
            return None
        return float(t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]) * batch / float(t.get("iterations_per_pass", batch))
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(wl, budget_s=12.0):
    """reference AVX2 backend on the host cores, bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import oracle_lib as O
        if not O.have_ref():
            return None
        threads = os.cpu_count() or 1
        kind = "AVX2"
        try:
            r = O.create_renderer(wl["w"], wl["h"], kind)
        except RuntimeError:
            kind = "REF"
            r = O.create_renderer(wl["w"], wl["h"], kind)
        s = r.CreateScene()
        build_scene(s, wl)
        spp_done, t_total = 0, 0.0
        step = 1
        while t_total < budget_s and spp_done < 64:
            t_total += r.render_tiled_mt(s, 32, step, threads)
            spp_done += step
            step = min(step * 2, 8)
        return {"value": wl["w"] * wl["h"] * spp_done / t_total / 1e6, "unit": "Msamples/s", "cores": threads,
                "kind": "reference",
                "sample": f"{kind} backend of the reference (oracle/_ref), {wl['w']}x{wl['h']}, {spp_done} spp, "
                          f"{threads} threads x 32x32 tiles, {t_total:.1f} s"}
    except Exception as e:  # the baseline is informational: never fail the bench because of it
        return {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)  # 4 passes of 120 iterations on one GPU, 1 pass of 480 on each of 8
    ap.add_argument("--warmup", type=int, default=120)  # one full pass of 120 iterations (1080p), the shape of the timed passes
    ap.add_argument("--workload", default=os.environ.get("RAY_AMD_WORKLOAD", "bistro"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner with the C library's printf (block-buffered
    # when piped, so it would land after anything Python printed): point file descriptor 1 at stderr for the duration of
    # the run and restore it on rank 0 for the JSON line only.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    # RAY_AMD_FORCE_DIST=1: run the N > 1 code path (process group, frame reduce over RCCL, re-tonemap) with one rank
    force_dist = world == 1 and os.environ.get("RAY_AMD_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    from ray_amd import api, hip, multigpu

    wl = WORKLOADS[args.workload]
    W, H, K, Wm = wl["w"], wl["h"], args.steps, args.warmup
    blob, info = get_scene_blob(args.workload, wl, rank, world, barrier)

    ctx = hip.Context(local_rank)
    ctx.upload_static(api.pmj_table())
    ctx.resize(W, H)
    cam = ctx.upload_scene_blob(blob)
    del blob
    ctx.set_shard(TILE, world, rank)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}") if dist is not None else None

    batch = int(os.environ.get("RAY_AMD_BATCH", "0")) or multigpu.batch_size(W * H // world, ctx.max_batch(), K)
    ctx.reserve_batch(batch)  # (the shard is set: a rank's buffers are sized for its share of the frame)
    it = 0
    if Wm > 0:  # untimed warm-up with the same pass shape (allocates the layered buffers)
        done = 0
        while done < Wm:
            n = min(batch, Wm - done)
            ctx.render_batch(it + 1, n)
            it, done = it + n, done + n
    if dist is not None:  # warm the communicator too
        ctx.readback_device(hip.BUF_RAW, frame.data_ptr())
        dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)
    ctx.sync()
    ctx.trav_timing(reset=True)
    ctx.stage_times(reset=True)

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # K iterations of this rank's tiles + (N>1) the one exchange step of the path: the frame reduce over RCCL/xGMI
    multigpu.render_sharded(ctx, range(it + 1, it + 1 + K), rank, world, dist=dist, frame=frame,
                            flags=hip.FLAG_TIME_STAGES, tile=TILE, batch=batch)
    it += K
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    (k2_ms, k2_launches), (k3_ms, k3_launches) = ctx.trav_timing(reset=True)
    stages = ctx.stage_times(reset=True)

    # algorithmic bytes of the traversal kernels: instrumented variant on the next iterations of the same workload
    n_count = max(1, min(K, 4))
    ctx.trav_counters(reset=True)
    for _ in range(n_count):
        it += 1
        ctx.render(it, flags=hip.FLAG_COUNT_TRAVERSAL)
    c2, c3 = ctx.trav_counters(reset=True)
    scale = K / n_count
    k2_bytes = ((72 + 20 + 4) * c2["rays"] + 64 * c2["nodes"] + 48 * c2["tris"] + 144 * c2["instances"]) * scale
    k3_bytes = ((48 + 32) * c3["rays"] + 64 * c3["nodes"] + 48 * c3["tris"] + 144 * c3["instances"]) * scale

    if dist is not None:  # whole-job traversal figures: sum over ranks
        v = torch.tensor([k2_bytes, k2_ms, k2_launches, k3_bytes, k3_ms], dtype=torch.float64, device=f"cuda:{local_rank}")
        vmax = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        k2_bytes_all, k2_ms_max = float(v[0]), float(vmax[1])
    else:
        k2_bytes_all, k2_ms_max = k2_bytes, k2_ms

    if rank == 0:
        samples = W * H * K
        achieved = (k2_bytes / 1e9) / (k2_ms / 1e3) if k2_ms > 0 else 0.0  # this rank's GPU: GB/s inside K2
        out = {
            "metric": "Msamples/sec (W*H*spp/time)",
            "value": samples / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": dt * 1e3 / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['label']}, {W}x{H}, {K} spp", "width": W, "height": H,
                       "spp": K, "unique_tris": info["tris"], "bvh_tris": info["bvh_tris"], "bvh2_nodes": info["nodes"],
                       "max_depth": int(cam.pass_settings.max_total_depth),
                       "parallelism": f"tile-shard x{world} (64x64 tiles round-robin, 1 RCCL reduce/frame)",
                       "iterations_per_pass": batch},
            "roofline": {
                "bound": "hbm", "kernel": "k_trace_closest<false,true> (closest-hit traversal, K2); algorithmic bytes = "
                                          "reference BVH2 visit counts on the same rays (SURVEY 8d)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_note": "algorithmic bytes are those of the reference's BVH2 walk (64 B/node visit, 48 B/triangle test); the "
                             "kernel walks a 4-wide quantised tree out of L1/L2, so frac > 1 means it finishes the reference's "
                             "traversal faster than HBM could stream it -- see traffic for what actually left L2",
                "traffic": measured_traffic(args.workload, batch),
                "alg_bytes_per_launch": k2_bytes / max(k2_launches, 1), "avg_launch_ms": k2_ms / max(k2_launches, 1),
                "launches": k2_launches,
                "alg_bytes_per_ray": k2_bytes / scale / max(c2["rays"], 1),
                "nodes_per_ray": c2["nodes"] / max(c2["rays"], 1), "tris_per_ray": c2["tris"] / max(c2["rays"], 1),
                "rays_per_sample": c2["rays"] / (n_count * W * H / world),
                "shadow_kernel": {"achieved": (k3_bytes / 1e9) / (k3_ms / 1e3) if k3_ms > 0 else 0.0,
                                  "avg_launch_ms": k3_ms / max(k3_launches, 1),
                                  "rays_per_sample": c3["rays"] / (n_count * W * H / world)},
            },
            "stage_us_per_step": {k: v / K for k, v in stages.items() if v},
            "scene_build_s": info["build_s"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    else:
        line = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # everything the libraries wrote to "stdout" went to stderr (see main's prologue); the real stdout gets the one line
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
