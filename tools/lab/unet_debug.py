import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, util
from ray_amd import hip
lib = hip.Library()
ctx = util.make_context(lib, "cornell_lights", 64, 48)
ctx.render_batch(1, 4)
w, o = O.ref_unet_weights(); ctx.unet_init(w, o, 8)
full0 = ctx.readback(hip.BUF_RAW).copy()
def run(half):
    ctx.unet_precision(half); ctx.denoise_unet(-1)
    return ctx.readback(hip.BUF_RAW).copy(), ctx.readback(hip.BUF_FINAL).copy()
ra, fa = run(False); rb, fb = run(True); rc, fc = run(False)
for n, (x, y) in {"raw f32 vs f16": (ra, rb), "raw f32 vs f32 again": (ra, rc), "final f32 vs f16": (fa, fb), "final f32 vs f32 again": (fa, fc)}.items():
    print(n, util.frame_metrics(y, x))
print("raw alpha", np.abs(ra[..., 3]).max(), np.abs(rb[..., 3]).max(), "raw max", ra[..., :3].max(), rb[..., :3].max())
for (W, H) in ((64, 48), (200, 136)):
    ctx = util.make_context(lib, "cornell_lights", W, H)
    ctx.render_batch(1, 4)
    ctx.unet_init(w, o, 8)
    ctx.unet_precision(True)
    ctx.denoise_unet(-1); r1 = ctx.readback(hip.BUF_RAW).copy()
    for p in range(16):
        ctx.denoise_unet(p)
    r2 = ctx.readback(hip.BUF_RAW).copy()
    ctx.denoise_unet(-1); r3 = ctx.readback(hip.BUF_RAW).copy()
    print(W, H, "f16 all-in-one vs pass by pass:", np.array_equal(r1, r2), float(np.abs(r1 - r2).max()), "again:", np.array_equal(r1, r3), r1[..., :3].max(), r2[..., :3].max())
ctx = util.make_context(lib, "cornell_lights", 64, 48)
ctx.render_batch(1, 4)
ctx.unet_init(w, o, 8)
full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
ref = O.ref_unet_passes(full, base, dn, 15)
for half in (False, True, False):
    ctx.unet_precision(half); ctx.denoise_unet(-1)
    got = ctx.readback(hip.BUF_RAW)
    err = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))
    print("half", half, "vs oracle: max rel", float(err.max()), "max got", float(got[..., :3].max()), "max ref", float(ref[..., :3].max()), "argmax", np.unravel_index(err.argmax(), err.shape))
