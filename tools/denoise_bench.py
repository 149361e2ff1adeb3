"""Time of RendererBase::DenoiseImage (NLM) on the bench workloads:  python tools/denoise_bench.py [workload] [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: librayhip shares torch's HIP runtime)

import bench
from ray_amd import api, hip


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "bistro"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    wl = bench.WORKLOADS[workload]
    W, H = wl["w"], wl["h"]
    blob, _ = bench.get_scene_blob(workload, wl, 0, 1, lambda: None)
    ctx = hip.Context(0)
    ctx.upload_static(api.pmj_table())
    ctx.resize(W, H)
    ctx.upload_scene_blob(blob)
    ctx.render_batch(1, spp)
    ctx.denoise_nlm(spp)  # warm-up (allocates the intermediates)
    ctx.sync()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.denoise_nlm(spp)
    ctx.sync()
    dt = (time.perf_counter() - t0) / n
    px = W * H
    # per pixel: 49 window positions x 9 patch taps x 4 channels x (3 sub, 2 mul, 2 add, 1 min, 1 div, ...) ~ 30 kflop
    print(f"{workload} {W}x{H}: NLM denoise {dt * 1e3:.2f} ms per frame, {px / dt / 1e6:.0f} Mpixels/s "
          f"(algorithmic traffic 96 B/pixel = {px * 96 / dt / 1e9:.1f} GB/s: compute-bound)")


if __name__ == "__main__":
    main()
