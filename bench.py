#!/usr/bin/env python3
"""bench.py -- Msamples/s of the HIP path-tracer core loop on N MI355X (BASELINE.json metric), one JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--workload bistro|sponza|cornell|principled]

A "step" is ONE FRAME of the BASELINE configuration: Clear, S = 64 samples per pixel over the whole frame (RenderScene
iterations 1..S: ray gen -> trace -> shade -> shadow -> [trace -> shade -> shadow] x bounces -> accumulate; the iterations of a
frame go through the device as one layered pass), and the finished frame leaving the GPU (N = 1: read-back
to the host; N > 1: the exchange that assembles it on rank 0).  K steps = K frames of S spp.  Msamples/s = W*H*S*K / time.
(Rounds 1-3 called one iteration a step, so the driver's `--steps 20` timed a 20-spp frame; since round 4 the driver's
command times twenty 64-spp frames -- the configuration BASELINE.json quotes the metric on.)

Workloads (BASELINE.json configs; no real Sponza/Bistro asset exists offline, see ray_amd/scenes.py):
  bistro      default.  synthetic atrium, ~3.0 M triangles, 1920x1080   (config 4: the scene the metric is quoted on)
  sponza      synthetic atrium, ~0.25 M triangles, 1920x1080            (config 3)
  cornell     samples/00_basic Cornell box, 1024x1024                   (config 2)
  principled  samples/03_principled, 2048x2048                          (config 5)

Multi-GPU (N > 1, one process per GPU: under torch.distributed.run, or -- started as a plain process -- bench.py launches
its N ranks itself): the scene is replicated, the frame's 64x64 tiles are dealt round-robin to the ranks
(rayhip_set_shard), every rank renders all K samples of ITS tiles, and ONE exchange inside the timed region assembles the
frame on rank 0: the product's own collective (rayhip_comm_reduce_framebuffers: every rank's owned tiles, densely packed,
point-to-point to the root over RCCL / xGMI; torch.distributed only carries the barrier and the communicator id).  Its
time is reported separately (`exchange_ms`).  Total work is fixed -> "scaling": "strong".  With more ranks than devices
(a 1-GPU box) the ranks share devices and the packed tiles travel through host memory over gloo -- RCCL cannot put two
ranks on one device -- and the line says `"emulated_ranks": true`: a plumbing check, not a scaling measurement.

Timed region: barrier + stream sync | K frames (each: clear, S iterations, read-back or exchange) | stream sync + barrier; max
over ranks.  Inputs (scene, PMJ table) are resident in HBM before the region starts; the only host traffic inside it is the
finished frame (33 MB once per frame at 1080p).

roofline (dominant kernel: the closest-hit traversal K2 -- the persistent kernel k_trace_closest_refill, lane-by-lane refill
for the secondary bounces, whole chunks for the primary rays; "a launch" is a launch of either).  Its launches are bracketed by HIP
events on the context stream during the timed region (RAYHIP_FLAG_TIME_STAGES, no synchronisation).  Three byte counts,
all per launch:
  traffic      HBM bytes that actually moved: rocprofv3 PMC FETCH_SIZE + WRITE_SIZE of this kernel, from a profiled run of
               THIS command line (profiles/r05/k2_traffic.json, keyed by workload / spp / iterations per pass; written by
               tools/k2_traffic.py from the rocprofv3 output and stamped with a hash of ray_amd/csrc); for another pass size
               or N > 1 the profiled run of the same workload with the nearest pass size, scaled by rays per launch
               (traffic_detail.exact = false says so); traffic_detail.stale = true when the kernel sources changed since the
               profile was taken; null when the workload was never profiled
  achieved     = traffic / launch time when traffic is known (the north star's figure: "achieved HBM GB/s from rocprof
               against the chip's memory roofline"), else the kernel's own algorithmic rate; frac = achieved / 8 TB/s, or null
               when no counter profile of the workload is committed (an algorithmic rate is an upper bound, not a fraction)
  traversal    (round 5) the same for K2 + K3 together -- SURVEY 8d's "traversal": the committed per-launch traffic of both kernels
               x their launches / the time the traversal took.  K3 of bounce b runs on a second stream next to K2 of bounce b + 1,
               so that time is the sum of the closest-hit intervals of the context stream (each ends when both launches are through)
  algorithmic  the kernel's OWN algorithmic bytes: 72+20+4 per ray + 64 per TLAS node + 64 per 4-wide node (80 per 8-wide
               node with RAYHIP_BVH_WIDTH=8) + 48 per triangle + 144 per instance, counted by the instrumented product kernel (RAYHIP_FLAG_COUNT_WIDE) on iterations
               of the same workload right after the timed region; next to it the same for the reference's BVH2 walk
               (SURVEY 8d's formula, RAYHIP_FLAG_COUNT_TRAVERSAL) -- what a cache-less machine would have to move
cpu_baseline: the reference's own AVX2 backend (oracle/_ref, kind "reference"), one persistent pool of worker threads (as
many as this process may run on: affinity mask and cgroup quota) pulling 32x32 tiles, ALL samples of a tile in one go, on a
bounded number of spp of the same scene and resolution (rank 0, N=1 only); per-stage split from RendererBase::GetStats.
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
    # the same atrium at four times the tessellation: ~12 M triangles, ~0.9 GB of nodes + triangle records -- 3.5 x the 256 MiB Infinity
    # Cache, so the walk is served by HBM (not a BASELINE.json config: the out-of-cache data point of the roofline)
    "bistro12m": dict(kind="atrium", detail=17.2, w=1920, h=1080, label="synthetic atrium, 12 M triangles (out of the last-level cache)"),
    # the same geometry with a texture set (11 mip-mapped 1024^2 maps: base colour on every large surface, a normal map and
    # roughness maps): not a BASELINE.json config -- the material -> texture gathers a textured asset set adds to the shade stage
    "bistro_tex": dict(kind="atrium", detail=4.3, textured=True, w=1920, h=1080, label="synthetic Bistro-class atrium, textured"),
    # ... and with settings_t::use_tex_compression (the reference's default): the maps live in the BC3 / BC4 / BC5 storages and
    # stay compressed on the device (RAYHIP_TEX_RAW_BC; RAY_HIP_DECODE_BC=1 expands them at export instead)
    # tuning experiment only: every surface Principled (what does material-type divergence inside a wavefront cost?)
    "bistro_1type": dict(kind="atrium", detail=4.3, one_material_type=True, w=1920, h=1080, label="synthetic Bistro-class atrium, one material type"),
    "bistro_texc": dict(kind="atrium", detail=4.3, textured=True, compressed=True, w=1920, h=1080,
                        label="synthetic Bistro-class atrium, textured, block-compressed"),
    # round 6 -- the Bistro-class scene as BASELINE.md 4.3 (4) specifies it: 38 jittered / rotated / scaled copies of the reference's own asset mesh
    # (tests/test_data/meshes/mat_test/model.bin, 77 762 triangles each; staged as a data file by tests/golden/stage_ref_assets.py) along a street
    # between two facades, ~200 emissive triangles; BAKED into one mesh (one bottom-level tree over 3.0 M triangles) ...
    "bistro_assets": dict(kind="street", copies=38, w=1920, h=1080, label="Bistro-class street: 38 baked copies of mat_test/model.bin"),
    # ... and with the copies as mesh INSTANCES (top-level tree over 41 instances: the two-level walk is on the clock)
    "bistro_assets_inst": dict(kind="street", copies=38, instanced=True, w=1920, h=1080,
                               label="Bistro-class street: 38 instances of mat_test/model.bin"),
    # the reference builder's spatial splits on the same two scenes (mesh_desc_t::allow_spatial_splits, BVHSplit.cpp:323-470): A/B only
    "bistro_assets_sbvh": dict(kind="street", copies=38, spatial_splits=True, w=1920, h=1080,
                               label="Bistro-class street: 38 baked copies of mat_test/model.bin, spatial splits"),
    "bistro_assets_inst_sbvh": dict(kind="street", copies=38, instanced=True, spatial_splits=True, w=1920, h=1080,
                                    label="Bistro-class street: 38 instances of mat_test/model.bin, spatial splits"),
    "cornell": dict(kind="cornell_basic", w=1024, h=1024, label="samples/00_basic Cornell box"),
    "principled": dict(kind="cornell_principled", w=2048, h=2048, label="samples/03_principled Cornell box"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
TILE = 64


def build_scene(scene, wl):
    from ray_amd import scenes
    if wl["kind"] == "atrium":
        return scenes.atrium(scene, wl["detail"], textured=wl.get("textured", False), compress=wl.get("compressed", False),
                             one_material_type=wl.get("one_material_type", False))
    if wl["kind"] == "street":
        return scenes.street_assets(scene, copies=wl["copies"], instanced=wl.get("instanced", False), spatial_splits=wl.get("spatial_splits", False))
    scenes.SCENES[wl["kind"]](scene)
    return scene.triangle_count()


def get_scene_blob(name, wl, rank, world, barrier):
    """rank 0 builds (reference host-side SAH BVH + light tree through SceneHIP) and caches; the others read."""
    from ray_amd import api
    cache_dir = os.environ.get("RAY_AMD_CACHE", "/tmp/ray_amd_cache")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"{name}_{wl.get('detail', 0)}{'_decoded' if os.environ.get('RAY_HIP_DECODE_BC') == '1' else ''}.rayscene")
    meta = path + ".json"
    t0 = time.time()
    if rank == 0 and not (os.path.exists(path) and os.path.exists(meta)):
        s = api.CreateSceneHIP(use_tex_compression=wl.get("compressed", False))
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


def csrc_hash():
    """hash of the kernel sources a committed PMC profile belongs to (tools/k2_traffic.py stamps its entries with it)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ray_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "ray_amd", "csrc", "*.hip"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_table_path():
    for r in ("r06", "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", r, "k2_traffic.json")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r05", "k2_traffic.json")


def measured_traffic(workload, spp, batch, world=1):
    """HBM bytes per K2 launch (FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3 PMC passes (profiles/r02/k2_traffic.json,
    written by tools/k2_traffic.py).  Exact when this very command line was profiled (same workload, steps, iterations per
    pass, one GPU); otherwise SCALED from the profiled run of the same workload with the nearest pass size -- K2's traffic
    per launch is proportional to the rays of the launch, i.e. to iterations per pass / ranks (64 vs 20 iterations per pass:
    0.5636 vs 0.5610 GB per iteration) -- and marked as such; None when the workload was never profiled."""
    try:
        with open(traffic_table_path()) as f:
            table = json.load(f)
        runs = [e for e in table.get("runs", []) if e["workload"] == workload]
        if not runs:
            return None
        # (a launch of a 64-layer pass is the same launch whatever the number of frames: the key is the shape of the pass.  Entries of
        # rounds 2-3 have no "spp": their frame was `steps` iterations long)
        exact = [e for e in runs if world == 1 and e.get("spp", e["steps"]) == spp and e["iterations_per_pass"] == batch]
        e = exact[0] if exact else min(runs, key=lambda r: abs(r["iterations_per_pass"] - batch))
        k = 1.0 if exact else (batch / e["iterations_per_pass"]) / world
        out = {"bytes_per_launch": float(e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]) * k,
               "fetch_bytes_per_launch": float(e["fetch_bytes_per_launch"]) * k,
               "write_bytes_per_launch": float(e["write_bytes_per_launch"]) * k,
               "profiled_avg_launch_ms": e.get("avg_launch_ms") if exact else None, "source": e.get("source"),
               "exact": bool(exact), "table": os.path.relpath(traffic_table_path(), ROOT),
               # the profile belongs to the kernels it was taken with: anything else is a stale figure
               "stale": e.get("csrc_hash") != csrc_hash(), "profiled_csrc_hash": e.get("csrc_hash")}
        if "shadow" in e:  # (round 5) the any-hit kernel of the same profiled passes: its launches scale like K2's
            sh = e["shadow"]
            out["shadow"] = {"bytes_per_launch": float(sh["fetch_bytes_per_launch"] + sh["write_bytes_per_launch"]) * k,
                             "fetch_bytes_per_launch": float(sh["fetch_bytes_per_launch"]) * k, "write_bytes_per_launch": float(sh["write_bytes_per_launch"]) * k,
                             "profiled_avg_launch_ms": sh.get("avg_launch_ms") if exact else None, "launches_per_pass": sh.get("launches_per_pass")}
        if "valu_wave_instructions_per_launch" in e:  # (the third profile of tools/k2_traffic.py: SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU)
            out["valu_wave_instructions_per_launch"] = float(e["valu_wave_instructions_per_launch"]) * k
            out["valu_active_lanes"] = e.get("valu_active_lanes")
        if not exact:
            out["scaled_by"] = k
            out["scaled_from"] = {"spp": e.get("spp", e["steps"]), "iterations_per_pass": e["iterations_per_pass"], "n_gpus": 1}
        return out
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass
    return None


def valu_issue_block(traffic, launches, k2_s):
    """the closest-hit kernel against the vector ALU's issue rate FOR ITS OWN INSTRUCTION MIX: vector instructions per launch from the
    committed PMC profile / the live launch time, against the mix-weighted peak of profiles/r04/k2_valu_mix.json (tools/valu_mix.py:
    static class mix of the walk loop x the per-class issue costs tools/valu_bench.hip measured).  `frac` = achieved / the serial-issue
    peak of the mix (every instruction at its class's cost: < 1 unless the classes overlap more than the model allows);
    `frac_paired_model` = against the peak if every full-rate instruction issued in the shadow of a half-rate one (an upper bound of
    the peak, so a lower bound of the fraction)"""
    if not (traffic and traffic.get("valu_wave_instructions_per_launch") and k2_s > 0):
        return None
    rate = traffic["valu_wave_instructions_per_launch"] * launches / k2_s
    out = {"wave_instructions_per_s": rate, "active_lanes_of_64": traffic.get("valu_active_lanes"),
           "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU of this command / live launch time"}
    try:
        mix_path = next(p for p in (os.path.join(ROOT, "profiles", r, "k2_valu_mix.json") for r in ("r06", "r05", "r04")) if os.path.exists(p))
        with open(mix_path) as f:
            mix = json.load(f)
        peak = mix["peak_wave_instructions_per_s"]
        out.update({"mix": mix["share"], "peak_mix_weighted": peak["serial"], "frac": rate / peak["serial"],
                    "peak_paired_model": peak["paired"], "frac_paired_model": rate / peak["paired"],
                    "mix_is_stale": mix.get("csrc_hash") != csrc_hash(),
                    "peaks_from": os.path.relpath(mix_path, ROOT) + " (tools/valu_mix.py) x profiles/r03/valu_bench.txt"})
    except (OSError, ValueError, KeyError, StopIteration):
        out.update({"peak_full_rate_class": 1.09e12, "peak_half_rate_class": 0.59e12})
    return out


def traversal_block(traffic, k2_launches, k3_launches, k2_ms, k3_ms, overlapped, algorithmic_bytes):
    """roofline of K2 + K3 together"""
    t_ms = k2_ms if overlapped else k2_ms + k3_ms
    out = {"kernels": "k_trace_closest_refill (both forms) + k_trace_shadow_refill", "ms_per_region": t_ms,
           "time_is": ("the trace stages of the context stream (a stage ends when the K2 launch AND the shadow launch beside it are through)" if overlapped
                       else "K2 intervals + K3 intervals"),
           "algorithmic_GBps": algorithmic_bytes / 1e9 / (t_ms / 1e3) if t_ms > 0 else None,
           "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
    if traffic and traffic.get("shadow") and t_ms > 0:
        total = traffic["bytes_per_launch"] * k2_launches + traffic["shadow"]["bytes_per_launch"] * k3_launches
        out.update({"traffic": total, "achieved": total / 1e9 / (t_ms / 1e3), "frac": total / 1e9 / (t_ms / 1e3) / HBM_PEAK_GBS})
    return out


def usable_cpus():
    """hardware threads this process may actually run on: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota|max> <period>"
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, period = float(f1.read()), float(f2.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(wl, budget_s=15.0):
    """reference AVX2 backend on the host cores, bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import oracle_lib as O
        if not O.have_ref():
            return None
        threads, quota = usable_cpus()
        kind = "AVX2"
        try:
            r = O.create_renderer(wl["w"], wl["h"], kind)
        except RuntimeError:
            kind = "REF"
            r = O.create_renderer(wl["w"], wl["h"], kind)
        s = r.CreateScene()
        build_scene(s, wl)
        # calibration: one sample per pixel (also pages the scene in and sizes the per-thread buffers) ...
        t1 = r.render_tiled_mt(s, 32, 1, threads)
        # ... then ONE call for all remaining samples: the pool lives for the whole call, every tile gets all its samples
        spp = int(max(1, min(63, budget_s / max(t1, 1e-3))))
        r.ResetStats()
        t = r.render_tiled_mt(s, 32, spp, threads)
        st = r.GetStats()
        tot = float(sum(st.values())) or 1.0
        return {"value": wl["w"] * wl["h"] * spp / t / 1e6, "unit": "Msamples/s", "cores": threads,
                "kind": "reference",
                "sample": f"{kind} backend of the reference (oracle/_ref), {wl['w']}x{wl['h']}, {spp} spp in one call after a "
                          f"1-spp warm-up ({t1:.2f} s), {threads} threads x 32x32 tiles, {t:.1f} s",
                "host": {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota},
                "first_spp_value": wl["w"] * wl["h"] / t1 / 1e6,
                "stage_share": {k: round(v / tot, 4) for k, v in st.items() if v}}
    except Exception as e:  # the baseline is informational: never fail the bench because of it
        return {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}


def parity_check(ctx, wl):
    """the benchmarked workload against the oracle, outside the timed region: iteration 1 of the full frame rendered by
    RendererRef (scalar reference backend, all usable cores, tiles) and by the HIP path, compared in the stated tolerance
    (tests/util.py: >= 99.5 % of pixels within 1e-3 * max(1, |ref|), PSNR >= 55 dB at 1 spp)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import oracle_lib as O
        import util
        from ray_amd import hip
        if not O.have_ref():
            return None
        threads, _ = usable_cpus()
        r = O.create_renderer(wl["w"], wl["h"], "REF")
        s = r.CreateScene()
        build_scene(s, wl)
        t = r.render_tiled_mt(s, 32, 1, threads)
        ref = r.get_raw_pixels_ref()
        ctx.clear()
        ctx.render(1)
        m = util.frame_metrics(ctx.readback(hip.BUF_RAW), ref)
        m.update({"against": "RendererRef (oracle/_ref), iteration 1, full frame", "ref_render_s": t,
                  "pass": bool(m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_1SPP)})
        return m
    except Exception as e:
        return {"pass": False, "error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)   # frames of --spp samples per pixel (8 x 0.2 s at 1080p / 64 spp on one GPU)
    ap.add_argument("--warmup", type=int, default=2)  # frames, untimed
    ap.add_argument("--spp", type=int, default=64)    # samples per pixel of a frame: BASELINE.json quotes the metric at 64
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
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as a plain process: launch the N ranks ourselves (one process per GPU) and pass rank 0's line through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.dup2(real_stdout, 1)
        raise SystemExit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    n_dev = torch.cuda.device_count()
    emulated = world > n_dev  # more ranks than devices: ranks share devices, the exchange goes through host memory (gloo)
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    # torch initialises its device context lazily, on the first call that needs it -- which would be the
    # torch.cuda.synchronize() that opens the timed region: the first timed pass of a process then started 15-30 ms late
    # (measured: 350 instead of 478 Msamples/s at --steps 20, repeats of the same region in the same process unaffected).
    # Do it here, with everything else that is set-up.
    torch.zeros(1, device=f"cuda:{local_rank}").add_(1.0)
    torch.cuda.synchronize()
    dist = None
    # RAY_AMD_FORCE_DIST=1: run the N > 1 code path (process group, frame reduce over RCCL, re-tonemap) with one rank
    force_dist = world == 1 and os.environ.get("RAY_AMD_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulated:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    from ray_amd import api, hip, multigpu

    wl = WORKLOADS[args.workload]
    W, H, K, Wm, SPP = wl["w"], wl["h"], args.steps, args.warmup, max(1, args.spp)
    blob, info = get_scene_blob(args.workload, wl, rank, world, barrier)

    ctx = hip.Context(local_rank)
    ctx.upload_static(api.pmj_table())
    ctx.resize(W, H)
    cam = ctx.upload_scene_blob(blob)
    ctx.set_shard(TILE, world, rank)
    # the exchange step: the product's collective (RCCL behind the C ABI); torch.distributed only hands the id around
    comm = None
    comm_error = None
    comm_info = None
    if dist is not None and not emulated:
        # Every rank must end up on the same transport, and NO rank may enter ncclCommInitRank (collective: it returns when all ranks are in it)
        # unless all of them will.  So: each rank probes RCCL locally (dlopen + symbols: rayhip_comm_probe; rank 0 also makes the id), the ranks
        # agree on the outcome by an all-reduce(MIN), and only then create their rank.  A job that cannot form the product's communicator is an
        # ERROR -- a SCALE line must never quietly measure another exchange -- unless RAY_AMD_ALLOW_FALLBACK=1 asks for torch.distributed's
        # gather of the same packed tiles (multigpu.exchange_frame); the line's top-level "transport" says which one ran.
        comm_error = hip.Comm.probe(ctx.L) or None
        ids = [None]
        if rank == 0 and comm_error is None:
            try:
                ids = [hip.Comm.unique_id(ctx.L)]
            except RuntimeError as e:
                comm_error = str(e)
        ok = torch.tensor([0 if comm_error else 1], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            dist.broadcast_object_list(ids, src=0)
            try:
                comm = hip.Comm.for_rank(ctx.L, ids[0], world, rank, ctx)
                comm_info = comm.info()
            except RuntimeError as e:  # (after the collective: every rank that got here has left it)
                comm_error = str(e)
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
                comm = None
            why = f"bench.py rank {rank}: rayhip_comm unavailable ({comm_error or 'on another rank'})"
            if os.environ.get("RAY_AMD_ALLOW_FALLBACK") != "1":
                dist.destroy_process_group()
                raise SystemExit(why + "; refusing to measure another transport (RAY_AMD_ALLOW_FALLBACK=1 moves the tiles with torch.distributed)")
            print(why + "; RAY_AMD_ALLOW_FALLBACK=1: the tiles go through torch.distributed", file=sys.stderr)

    def exchange():
        if dist is not None:
            multigpu.exchange_frame(ctx, rank, world, comm=comm, dist=dist, what=hip.REDUCE_RADIANCE, via_host=emulated)

    batch = int(os.environ.get("RAY_AMD_BATCH", "0")) or multigpu.batch_size(W * H // world, ctx.max_batch(), SPP)
    ctx.reserve_batch(batch)  # (the shard is set: a rank's buffers are sized for its share of the frame)
    # Host memory the HIP runtime pinned for a copy (the scene blob, the PMJ table, a read-back frame) must not be unmapped
    # while the GPU works: the kernel driver answers the unmap by stopping and restarting this process's GPU queues, and the
    # kernels running at that moment stand still for 15-30 ms (measured: 350 instead of 478 Msamples/s at --steps 20 whenever
    # a large numpy array happened to be freed around the start of the timed region).  So: the frame buffer of the read-back
    # exists -- and is touched -- before the first pass, the blob stays referenced until after it, and the garbage collector
    # rests.  (A C++ host has the same rule: allocate the read-back buffer once.)
    import gc
    # ... in page-locked memory: the device copies straight into it, and a pageable
    # np.zeros buffer is not even mapped before its first write (first process on a box: 7 ms of page faults inside the region)
    host_frame = torch.zeros((H, W, 4), dtype=torch.float32).pin_memory().numpy() if dist is None else None
    if host_frame is not None:
        host_frame.fill(0.0)
    gc.collect()
    gc.disable()
    t_render = t_out = 0.0

    def render_frame(flags, timed=False):
        """one step: Clear, iterations 1..SPP of this rank's tiles, the finished frame out of the GPU (read-back / exchange)"""
        nonlocal t_render, t_out
        ta = time.perf_counter()
        ctx.clear()
        multigpu.render_sharded(ctx, range(1, SPP + 1), rank, world, dist=None, frame=None, flags=flags, tile=TILE, batch=batch)
        ctx.sync()
        tb = time.perf_counter()
        if dist is None:  # what the exchange is at N > 1: the finished frame leaves the GPU once per image
            ctx.readback(hip.BUF_RAW, out=host_frame)
        else:
            exchange()
        ctx.sync()
        if timed:
            t_render, t_out = t_render + (tb - ta), t_out + (time.perf_counter() - tb)
            ctx.stage_times(reset=False)  # (resolves this frame's stage events into the running sums: the event pool stays small)

    # set-up, not warm-up: one frame of exactly the shape and flags of the timed ones, so that nothing in the timed region
    # is the first of its kind in this process (measured: the first timed pass of the first process on a fresh box spent
    # 15-30 ms before its first kernel finished when the warm-up passes were shorter than the timed ones)
    render_frame(hip.FLAG_TIME_STAGES)
    for _ in range(Wm):  # the W untimed warm-up steps
        render_frame(0)
    torch.cuda.synchronize()
    ctx.trav_timing(reset=True)
    ctx.stage_times(reset=True)

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        render_frame(hip.FLAG_TIME_STAGES, timed=True)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    del blob
    it = SPP
    rank_times = None
    if dist is not None:
        # (after the timed region) every rank's own render time and the exchange as rank 0 saw it -- the wait for the
        # slowest rank included -- and the maximum of the region over the ranks
        mine = [dt, t_render, t_out]
        every = [None] * world
        dist.all_gather_object(every, mine)
        dt = max(e[0] for e in every)
        rank_times = {"render_ms": [round(e[1] * 1e3 / K, 3) for e in every], "exchange_ms_rank0": round(every[0][2] * 1e3 / K, 3)}

    (k2_ms, k2_launches), (k3_ms, k3_launches) = ctx.trav_timing(reset=True)
    stages = ctx.stage_times(reset=True)
    for rep in range(int(os.environ.get("RAY_AMD_BENCH_REPEAT", "0"))):  # diagnostics: the same timed sequence again
        ctx.stage_times(reset=True)
        torch.cuda.synchronize()
        t_rep = time.perf_counter()
        for _ in range(K):
            render_frame(hip.FLAG_TIME_STAGES)
        ctx.sync()
        st_rep = ctx.stage_times(reset=True)
        ctx.trav_timing(reset=True)
        print(f"repeat {rep}: {W * H * SPP * K / (time.perf_counter() - t_rep) / 1e6:.1f} Msamples/s, ray gen {st_rep['primary_ray_gen'] / K:.0f} us/frame, "
              f"primary trace {st_rep['primary_trace'] / K:.0f} us/frame (first timed region of this process: {W * H * SPP * K / dt / 1e6:.1f}, "
              f"ray gen {stages['primary_ray_gen'] / K:.0f}, primary trace {stages['primary_trace'] / K:.0f})", file=sys.stderr)

    # algorithmic bytes of the traversal kernels: instrumented variants on the next iterations of the same workload --
    # first the product kernels with counters (their own bytes), then the reference's BVH2 walk on the same rays
    n_count = 2
    scale = K * SPP / n_count  # (counted iterations -> all iterations of the timed region)

    def count_pass(flag):
        nonlocal it
        ctx.trav_counters(reset=True)
        for _ in range(n_count):
            it += 1
            ctx.render(it, flags=flag)
        return ctx.trav_counters(reset=True)

    bvh_width = ctx.bvh_width()
    overlap_shadow = os.environ.get("RAYHIP_OVERLAP_SHADOW", "1") != "0" and bvh_width == 4 and os.environ.get("RAYHIP_SHADOW_REFILL", "1") != "0"
    wide_node_bytes = 80 if bvh_width == 8 else 64  # what a node visit reads: the 8-wide node uses 80 bytes of its 128-byte line

    def alg_bytes(c, per_ray):
        return (per_ray * c["rays"] + 64 * c["nodes"] + wide_node_bytes * c.get("nodes4", 0) + 48 * c["tris"] + 144 * c["instances"]) * scale

    w2, w3 = count_pass(hip.FLAG_COUNT_WIDE)
    c2, c3 = count_pass(hip.FLAG_COUNT_TRAVERSAL)
    k2_bytes, k3_bytes = alg_bytes(w2, 72 + 20 + 4), alg_bytes(w3, 48 + 32)          # the kernels' own
    k2_bytes_ref, k3_bytes_ref = alg_bytes(c2, 72 + 20 + 4), alg_bytes(c3, 48 + 32)  # the reference algorithm's

    if rank == 0:
        samples = W * H * SPP * K
        launches = max(k2_launches, 1)
        k2_s = k2_ms / 1e3
        alg_gbs = (k2_bytes / 1e9) / k2_s if k2_s > 0 else 0.0          # the kernel's own algorithmic bytes per second
        traffic = measured_traffic(args.workload, SPP, batch, world)
        hbm_gbs = (traffic["bytes_per_launch"] * launches / 1e9) / k2_s if (traffic and k2_s > 0) else None
        achieved = hbm_gbs if hbm_gbs is not None else alg_gbs
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
            "step": f"one {SPP}-spp frame: clear, {SPP} iterations ({-(-SPP // batch)} layered pass(es) of {batch}), " +
                    ("read-back of the frame to the host" if world == 1 else "exchange of the owned tiles to rank 0"),
            "config": {"workload": f"{args.workload}: {wl['label']}, {W}x{H}, {SPP} spp", "width": W, "height": H,
                       "spp": SPP, "frames": K, "unique_tris": info["tris"], "bvh_tris": info["bvh_tris"], "bvh2_nodes": info["nodes"],
                       "max_depth": int(cam.pass_settings.max_total_depth),
                       "parallelism": f"tile-shard x{world} (64x64 tiles round-robin, 1 gather of the owned tiles per frame over RCCL)",
                       "iterations_per_pass": batch},
            "roofline": {
                "bound": "hbm", "kernel": f"K2 closest-hit traversal over the {bvh_width}-wide BLAS: " +
                ("k_trace_closest_pool (secondary bounces: finished lanes take prepared rays from a per-wavefront pool in LDS)" if ctx.closest_hit_form() == 2
                 else f"k_trace_closest_refill<{bvh_width}, 40> (secondary bounces, lanes refilled one by one)") +
                f" + k_trace_closest_refill<{bvh_width}, 64> (primary rays, whole chunks)",
                # (no committed PMC profile of this workload: `achieved` is the kernel's own algorithmic rate -- an upper bound, every visit counted
                # as a miss -- and no fraction of the HBM peak is claimed for it: VERDICT round 4, weak 4)
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if hbm_gbs is not None else None,
                "achieved_is": (("HBM traffic (rocprofv3 FETCH_SIZE + WRITE_SIZE of a profiled run of this command) / launch time"
                                 if traffic.get("exact") else
                                 "HBM traffic scaled from the profiled run of this workload with the nearest pass size (traffic is proportional "
                                 "to the rays of a launch) / launch time of rank 0")
                                if hbm_gbs is not None else
                                "the kernel's own algorithmic bytes / launch time (no PMC profile of this workload is committed): an upper "
                                "bound of the HBM rate, every visit counted as a miss"),
                "traffic_is_stale": bool(traffic and traffic.get("stale")),  # profile taken with other kernel sources: re-profile (tools/k2_traffic.py)
                "traffic": traffic["bytes_per_launch"] if traffic else None,
                "traffic_detail": traffic,
                # the kernel's BINDING resource is the issue rate of the vector ALU: vector instructions per launch from the
                # committed PMC profile / the live launch time, against what tools/valu_bench.hip measures for one instruction class
                # (all full-rate: mul / add / logic; all half-rate: compares, selects, min / max, conversions, 3-operand integer)
                "valu_issue": valu_issue_block(traffic, launches, k2_s),
                "avg_launch_ms": k2_ms / launches, "launches": k2_launches,
                "algorithmic": {
                    "bytes_per_launch": k2_bytes / launches, "GBps": alg_gbs, "frac_of_peak": alg_gbs / HBM_PEAK_GBS,
                    "bytes_per_ray": k2_bytes / scale / max(w2["rays"], 1),
                    "bvh_width": bvh_width, "wide_node_bytes": wide_node_bytes,
                    "tlas_nodes_per_ray": w2["nodes"] / max(w2["rays"], 1), "wide_nodes_per_ray": w2["nodes4"] / max(w2["rays"], 1),
                    "tris_per_ray": w2["tris"] / max(w2["rays"], 1), "instances_per_ray": w2["instances"] / max(w2["rays"], 1),
                    "reference_bvh2": {"bytes_per_launch": k2_bytes_ref / launches,
                                       "bytes_per_ray": k2_bytes_ref / scale / max(c2["rays"], 1),
                                       "nodes_per_ray": c2["nodes"] / max(c2["rays"], 1), "tris_per_ray": c2["tris"] / max(c2["rays"], 1),
                                       "note": "what the reference's BVH2 walk would move for the same rays; not a rate of this kernel"},
                },
                "rays_per_sample": c2["rays"] / (n_count * W * H / world),
                "shadow_kernel": {"algorithmic_bytes_per_launch": k3_bytes / max(k3_launches, 1),
                                  "elapsed_ms_per_launch": k3_ms / max(k3_launches, 1),
                                  "elapsed_is": ("on a second, low-priority stream NEXT TO the closest-hit launch of the following bounce (librayhip, round 5): it takes "
                                                 "the wave slots that launch leaves idle, so its elapsed time is not its cost -- the traversal block below "
                                                 "prices K2 and K3 together") if overlap_shadow else "alone on the context stream",
                                  "rays_per_sample": c3["rays"] / (n_count * W * H / world)},
                # SURVEY 8d's "traversal": K2 + K3 together.  Bytes: the PMC profile's per-launch figures of both kernels x their launches (or
                # null); time: the closest-hit intervals of the context stream, which end when BOTH the K2 launch and the K3 launch that ran
                # next to it are through (the last K3 of a pass runs alone: its few microseconds are in the shade interval that follows)
                "traversal": traversal_block(traffic, k2_launches, k3_launches, (stages["primary_trace"] + stages["secondary_trace"]) / 1e3 if overlap_shadow else k2_ms,
                                             k3_ms, overlap_shadow, k2_bytes + k3_bytes),
            },
            "stage_us_per_step": {k: v / K for k, v in stages.items() if v},
            "stage_us_per_spp": {k: v / (K * SPP) for k, v in stages.items() if v},
            "scene_build_s": info["build_s"],
        }
        # RendererBase::stats_t is a partition of the pass (exclusive intervals of the context stream, include/rayhip.h): the stage entries may not
        # add up to more than the step (they did in round 5, when the shadow launch beside K2 was booked with its elapsed time)
        out["stage_sum_over_step"] = sum(out["stage_us_per_step"].values()) / 1e3 / out["ms_per_step"]
        if out["stage_sum_over_step"] > 1.02:
            raise SystemExit(f"bench.py: the stage times add up to {out['stage_sum_over_step']:.3f} x the step: rayhip_get_stage_times is not a partition")
        out["render_ms"] = t_render * 1e3 / K   # per frame: clear + the SPP iterations, stream drained
        if dist is None:
            out["readback_ms"] = t_out * 1e3 / K  # per frame: the finished frame to (page-locked) host memory
        if rank_times is not None:
            out["exchange_ms"] = rank_times["exchange_ms_rank0"]
            out["rank_render_ms"] = rank_times["render_ms"]
            out["exchange"] = ("owned tiles through host memory over gloo (ranks share a device: RCCL cannot form the communicator)" if emulated else
                               "rayhip_comm_reduce_framebuffers: owned tiles (1/N of the frame per rank) point-to-point to rank 0 over RCCL"
                               if comm is not None else
                               "torch.distributed gather of the owned tiles over RCCL (rayhip_comm could not be formed, RAY_AMD_ALLOW_FALLBACK=1: see stderr)")
            # top level, for whoever reads a SCALE record: which transport moved the tiles, what RCCL says the communicator is, every rank's time
            out["transport"] = "gloo-host (emulated ranks)" if emulated else ("rayhip_comm/rccl" if comm is not None else "torch.distributed/rccl (fallback)")
            out["ncclCommCount"] = comm_info["nccl_comm_count"] if comm_info else None
            out["render_ms_per_rank"] = rank_times["render_ms"]
        if emulated:
            out["emulated_ranks"] = True
            out["devices"] = n_dev
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
            out["parity"] = parity_check(ctx, wl)
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    else:
        line = None
    if comm is not None:
        comm.close()
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
