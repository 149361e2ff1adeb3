"""Quick first-contact probe on the GPU box: timing of the stage schedule on the golden Cornell scenes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util
from ray_amd import hip

lib = hip.Library()
print("devices", lib.device_count())
for name, w, h, spp in [("cornell_basic", 1024, 1024, 16), ("cornell_principled", 1024, 1024, 16)]:
    ctx = util.make_context(lib, name, w, h)
    ctx.render(1); ctx.sync()
    t = time.time()
    for it in range(2, spp + 2):
        ctx.render(it)
    ctx.sync()
    dt = time.time() - t
    print(f"{name} {w}x{h} {spp} spp: {dt*1e3:.1f} ms -> {w*h*spp/dt/1e6:.1f} Msamples/s")
    st = hip.Stats()
    for it in range(spp + 2, spp + 6):
        ctx.render(it, stats=st)
    print("  stats(us, 4 iters):", st.as_dict(), "trav timing", ctx.trav_timing())
    ctx.render(spp + 6, flags=hip.FLAG_COUNT_TRAVERSAL)
    print("  counters:", ctx.trav_counters())
