#!/usr/bin/env python3
"""BASELINE.json configs 2 and 5 at their STATED sample counts against RendererRef, once (the suite stops at 64 spp: the bar of
the stated tolerance is the same from 64 spp on, and the scalar oracle needs minutes for these):

    samples/00_basic Cornell box   1024 x 1024, 256 spp
    samples/03_principled          2048 x 2048, 512 spp

Runs on the GPU box (the oracle library travels with the snapshot); prints the metrics of tests/util.py for every checkpoint.
    python tools/full_spp_parity.py > profiles/r04/full_spp_parity.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: the image pages in)

import util  # noqa: E402
from ray_amd import hip  # noqa: E402
from test_gpu_baseline_configs import Workload  # noqa: E402


def run(name, checkpoints):
    wk = Workload(name)
    n = min(64, wk.ctx.max_batch())
    done = 0
    ok = True
    for spp in checkpoints:
        t0 = time.time()
        while done < spp:
            k = min(n, spp - done)
            wk.ctx.render_batch(done + 1, k)
            done += k
        img = wk.ctx.readback(hip.BUF_RAW)
        t_gpu = time.time() - t0
        m = util.frame_metrics(img, wk.ref_frame(spp))
        good = m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_64SPP and m["alpha_equal"]
        ok = ok and good
        print(f"{name} {wk.w}x{wk.h} {spp:4d} spp: {'PASS' if good else 'FAIL'}  within tolerance {100 * m['frac_within']:.5f} %  PSNR {m['psnr']:.1f} dB  "
              f"max |d| {m['max_abs']:.3g}  bit-equal {100 * m['exact']:.1f} %  (HIP: {t_gpu:.2f} s for this stretch)", flush=True)
    return ok


if __name__ == "__main__":
    print(f"bar: >= {100 * util.MIN_FRACTION} % of the pixels within {util.TOL_REL} * max(1, |ref|), PSNR >= {util.MIN_PSNR_64SPP} dB (>= 64 spp)")
    a = run("cornell", [64, 256])
    b = run("principled", [64, 256, 512])
    raise SystemExit(0 if (a and b) else 1)
