"""Where do the occasional 15-30 ms at the start of a pass go?  30 passes of the benchmark shape, each after a sync: per pass the
ray-generation stage time (events on the stream) and the host time of the calls around it (RAYHIP_TRACE_LAUNCH)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import bench
from ray_amd import api, hip

os.environ["RAYHIP_TRACE_LAUNCH"] = "1"
wl = bench.WORKLOADS["bistro"]
blob, _ = bench.get_scene_blob("bistro", wl, 0, 1, lambda: None)
ctx = hip.Context(0)
ctx.upload_static(api.pmj_table())
ctx.resize(wl["w"], wl["h"])
ctx.upload_scene_blob(blob)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if os.environ.get("PROBE_SHARD"):
    ctx.set_shard(bench.TILE, int(os.environ["PROBE_SHARD"]), 0)
ctx.reserve_batch(B)
it = 0
use_torch = len(sys.argv) > 3 and sys.argv[3] == "torch"
if use_torch:
    torch.cuda.set_device(0)
    scratch = torch.zeros(1024, device="cuda:0")
slow = 0
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    ctx.sync()
    if use_torch:  # what bench.py does around its timed region
        scratch += 1.0
        torch.cuda.synchronize()
    ctx.stage_times(reset=True)
    if k % 3 == 2:
        ctx.readback(hip.BUF_RAW)  # what the timed region of bench.py does after its pass
    idle = os.environ.get("PROBE_IDLE_MS")  # GPU idle before the pass: "50,200,500" cycles through the list
    if idle:
        ms = [float(x) for x in idle.split(",")]
        time.sleep(ms[k % len(ms)] / 1e3)
    alt = int(os.environ.get("PROBE_ALT", "0"))  # every second pass this many iterations instead of B (bench.py: --warmup 5 before 20)
    n = alt if (alt and k % 2 == 1) else B
    t0 = time.perf_counter()
    ctx.render_batch(it + 1, n, flags=0 if (n != B and os.environ.get("PROBE_ALT_NOFLAG")) else hip.FLAG_TIME_STAGES)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    st = ctx.stage_times(reset=True)
    it += n
    slow += st["primary_ray_gen"] / 1e3 > 2.0 * n / 20.0
    print(f"pass {k:2d} ({n} it): primary trace {st['primary_trace'] / 1e3:6.2f} ms  submit {1e3 * (t1 - t0):7.2f} ms  total {1e3 * (t2 - t0):7.2f} ms  ray gen stage {st['primary_ray_gen'] / 1e3:7.2f} ms", file=sys.stderr, flush=True)
print(f"passes with a slow start: {slow}", file=sys.stderr)
print("last pass, ms per stage:", {k: round(v / 1e3, 2) for k, v in st.items() if v}, file=sys.stderr)
