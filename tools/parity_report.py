"""Frame-level parity figures of the HIP path against the committed RendererRef fixtures (tests/golden), as quoted in
DESIGN.md section 7.  Runs on a GPU box:  python tools/parity_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: librayhip shares torch's HIP runtime)

import util  # noqa: E402
from ray_amd import hip  # noqa: E402
from test_gpu_parity import SCENES  # noqa: E402


def main():
    lib = hip.Library()
    for name in SCENES:
        g = util.golden_ref(name)
        ctx = util.make_context(lib, name)
        ctx.render(1)
        m1 = util.frame_metrics(ctx.readback(hip.BUF_RAW), g["raw_spp1"])
        for it in range(2, 9):
            ctx.render(it)
        m8 = util.frame_metrics(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
        mf = util.frame_metrics(ctx.readback(hip.BUF_FINAL), g["final_spp8"])
        print(f"{name:20s} 1 spp: within {m1['frac_within']:.5f} exact {m1['exact']:.4f} max|d| {m1['max_abs']:.2e} PSNR {m1['psnr']:.1f} dB | "
              f"8 spp: within {m8['frac_within']:.5f} exact {m8['exact']:.4f} max|d| {m8['max_abs']:.2e} PSNR {m8['psnr']:.1f} dB | "
              f"final: within {mf['frac_within']:.5f} PSNR {mf['psnr']:.1f} dB")


if __name__ == "__main__":
    main()
