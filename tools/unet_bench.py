#!/usr/bin/env python3
"""The sixteen UNet passes on a 1920 x 1080 frame, N times (for rocprofv3: per-pass kernel times, matrix-core busy cycles).
    python tools/unet_bench.py [n] [f32|f16]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

import oracle_lib as O  # noqa: E402
import util  # noqa: E402
from ray_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = util.make_context(hip.Library(), "cornell_lights", 1920, 1080)
ctx.render_batch(1, 1)
weights, offsets = O.ref_unet_weights()
ctx.unet_init(weights, offsets, 8)
half = len(sys.argv) > 2 and sys.argv[2] == "f16"
ctx.unet_precision(half)
ctx.denoise_unet(-1)
ctx.sync()
t0 = time.perf_counter()
for _ in range(n):
    ctx.denoise_unet(-1)
ctx.sync()
ms = (time.perf_counter() - t0) / n * 1e3
flops = 2 * 125406 * 1920 * 1080
peak = 2500.0 if half else 157.0
print(f"UNet 1080p, {'f16' if half else 'f32'} form: {ms:.2f} ms per frame, {flops / ms / 1e9:.1f} TFLOP/s = {100 * flops / ms / 1e9 / peak:.1f} % of the "
      f"{'f16' if half else 'f32'} matrix peak ({peak:.0f})")
