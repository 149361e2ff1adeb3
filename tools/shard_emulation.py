"""One rank of an N-GPU tile-sharded render, timed on ONE GPU: what each GPU of the node would do (the ranks do not
communicate until the final frame reduce), hence the scaling efficiency the tile sharding can reach.

    python tools/shard_emulation.py [workload] [steps]

Prints, per world size N: iterations per pass, time of rank 0's share, projected whole-job Msamples/s (N x rank rate)
and the efficiency against the N = 1 run of the same process.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: librayhip shares torch's HIP runtime)

import bench
from ray_amd import api, hip, multigpu


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "bistro"
    Ks = [int(x) for x in sys.argv[2:]] or [64, 20]
    wl = bench.WORKLOADS[workload]
    W, H = wl["w"], wl["h"]
    blob, _ = bench.get_scene_blob(workload, wl, 0, 1, lambda: None)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    for K in Ks:
        base = None
        print(f"== {workload} {W}x{H}, one frame of {K} spp (64: a step of `bench.py` since round 4; 20: the frame of round 3's driver line)")
        for world in (1, 2, 4, 8):
            ctx = hip.Context(0)
            ctx.upload_static(api.pmj_table())
            ctx.resize(W, H)
            ctx.upload_scene_blob(blob)
            ctx.set_shard(bench.TILE, world, 0)
            batch = multigpu.batch_size(W * H // world, ctx.max_batch(), K)
            ctx.reserve_batch(batch)
            ctx.render_batch(1, batch)  # set-up pass of the timed shape, like bench.py
            ctx.sync()
            ctx.stage_times(reset=True)
            t0 = time.perf_counter()
            multigpu.render_sharded(ctx, range(batch + 1, batch + 1 + K), 0, world, batch=batch, flags=hip.FLAG_TIME_STAGES)
            # this rank's operand of the frame reduce (the collective itself: 33 MB over xGMI, not emulated here)
            ctx.export_shard_device(hip.BUF_RAW, frame.data_ptr())
            ctx.sync()
            dt = time.perf_counter() - t0
            rate = W * H * K / world / dt / 1e6 * world
            base = base or rate
            print(f"N={world} iterations/pass {batch:4d}  rays/pass {W * H // world * batch / 1e6:6.1f} M  rank time {dt * 1e3:8.1f} ms  "
                  f"projected {rate:7.1f} Msamples/s  efficiency {rate / (base * world):5.3f}", flush=True)
            st = ctx.stage_times(reset=True)
            print("      stage ms: " + "  ".join(f"{k.replace('primary_', 'p.').replace('secondary_', 's.')} {v / 1e3:.2f}" for k, v in st.items() if v), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
