"""The reference's own test scene (the material ball of tests/test_shading.cpp, `complex_mat5`: textured metal, rect + disk light, closed room) as
a throughput data point on REAL asset meshes: 1920 x 1080, 64 spp, device against the reference's AVX2 backend on the host cores.  Not a
BASELINE.json configuration -- bench.py has those; this answers "and on the reference's own assets?".  Runs on a GPU box:
    python tools/material_ball_bench.py [entry] [spp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

import oracle_lib as O  # noqa: E402
import ref_material_scene as M  # noqa: E402
import util  # noqa: E402
from ray_amd import api, hip  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "complex_mat5"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    w, h = 1920, 1080
    entry = dict(next(e for e in M.matrix() if e["name"] == name), min_samples=spp, max_samples=spp)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    kind = "AVX2"
    try:
        cpu = O.create_renderer(w, h, kind)
    except RuntimeError:
        kind = "REF"
        cpu = O.create_renderer(w, h, kind)
    s = cpu.CreateScene()
    notes = M.build(s, entry)["notes"]
    t1 = cpu.render_tiled_mt(s, 32, 1, threads)
    n_cpu = int(max(1, min(spp - 1, 12.0 / max(t1, 1e-3))))
    t_cpu = cpu.render_tiled_mt(s, 32, n_cpu, threads, iterations_done=1)
    ctx = hip.Context(0, hip.Library())
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ref = O.create_renderer(w, h, "REF")  # (the blob comes from the scalar backend's scene: the SIMD backends keep a wide tree of their own)
    rs = ref.CreateScene()
    M.build(rs, entry)
    ctx.upload_scene_blob(O.export_scene(rs))
    ctx.reserve_batch(spp)
    ctx.render_batch(1, spp)  # (set-up: buffers of the timed shape)
    ctx.sync()
    times = []
    for _ in range(3):
        ctx.clear()
        t0 = time.perf_counter()
        ctx.render_batch(1, spp)
        ctx.sync()
        times.append(time.perf_counter() - t0)
    dev = w * h * spp / min(times) / 1e6
    ctx.clear()
    ctx.stage_times(reset=True)
    ctx.render_batch(1, spp, flags=hip.FLAG_TIME_STAGES)
    ctx.sync()
    stages = ctx.stage_times(reset=True)
    # visit census of the product's walk on this scene (the counting form of the kernels: two more iterations)
    ctx.trav_counters(reset=True)
    for it in (spp + 1, spp + 2):
        ctx.render(it, flags=hip.FLAG_COUNT_WIDE)
    c2, c3 = ctx.trav_counters(reset=True)
    host = w * h * n_cpu / t_cpu / 1e6
    # parity of this very frame: the first 1 + n_cpu samples of both
    ctx.clear()
    ctx.render_batch(1, 1 + n_cpu)
    ref.render_tiled_mt(rs, 32, 1 + n_cpu, threads)
    m = util.frame_metrics(ctx.readback(hip.BUF_RAW), ref.get_raw_pixels_ref())
    print(f"{name} ({rs.triangle_count()} triangles in the scene's meshes; {'; '.join(notes)}), {w} x {h}:")
    print(f"  device: {spp} spp in {min(times) * 1e3:.1f} ms (best of 3: {', '.join(f'{t * 1e3:.1f}' for t in times)}) = {dev:.1f} Msamples/s")
    print("  stages, ms per frame (stats_t counts microseconds): " + "  ".join(f"{k} {v / 1e3:.1f}" for k, v in stages.items() if v))
    r2, r3 = max(c2["rays"], 1), max(c3["rays"], 1)
    print(f"  census: {c2['rays'] / (2 * w * h):.2f} closest-hit rays per sample, per ray {c2['nodes'] / r2:.2f} top-level nodes, {c2['instances'] / r2:.2f} instances entered, "
          f"{c2['nodes4'] / r2:.2f} wide nodes, {c2['tris'] / r2:.2f} triangle tests; {c3['rays'] / (2 * w * h):.2f} shadow rays per sample, per ray {c3['nodes'] / r3:.2f} top-level nodes, "
          f"{c3['instances'] / r3:.2f} instances, {c3['nodes4'] / r3:.2f} wide nodes, {c3['tris'] / r3:.2f} triangle tests")
    print(f"  reference {kind} backend, {threads} threads: {n_cpu} spp in {t_cpu:.1f} s after a 1-spp warm-up ({t1:.2f} s) = {host:.2f} Msamples/s  -> x{dev / host:.0f}")
    print(f"  parity at {1 + n_cpu} spp against RendererRef: {m['frac_within'] * 100:.4f} % within tolerance, {m['psnr']:.1f} dB, {m['exact'] * 100:.1f} % of the pixels bit-equal")


if __name__ == "__main__":
    main()
