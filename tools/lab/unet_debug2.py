import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, util
from ray_amd import hip
lib = hip.Library()
w, o = O.ref_unet_weights()
for (W, H) in ((64, 48), (96, 80), (200, 136)):
    ctx = util.make_context(lib, "cornell_lights", W, H)
    ctx.render_batch(1, 4)
    ctx.unet_init(w, o, 8)
    full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
    ref = O.ref_unet_passes(full, base, dn, 15)
    for mode in ("all", "each", "all"):
        if mode == "all":
            ctx.denoise_unet(-1)
        else:
            for p in range(16):
                ctx.denoise_unet(p)
        got = ctx.readback(hip.BUF_RAW)
        err = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))
        print(W, H, mode, "f32 vs oracle: max rel", float(err.max()), flush=True)
