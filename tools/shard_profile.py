"""One rank's share of an N-GPU frame under rocprofv3: which kernels lose against 1/N of the unsharded frame?  (tuning tool)

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o shard -- python tools/shard_profile.py <world> [spp]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import bench
from ray_amd import api, hip, multigpu


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    wl = bench.WORKLOADS["bistro"]
    W, H = wl["w"], wl["h"]
    blob, _ = bench.get_scene_blob("bistro", wl, 0, 1, lambda: None)
    ctx = hip.Context(0)
    ctx.upload_static(api.pmj_table())
    ctx.resize(W, H)
    ctx.upload_scene_blob(blob)
    ctx.set_shard(bench.TILE, world, 0)
    batch = multigpu.batch_size(W * H // world, ctx.max_batch(), spp)
    ctx.reserve_batch(batch)
    for frame in range(3):  # (the first is the set-up pass)
        multigpu.render_sharded(ctx, range(1, 1 + spp), 0, world, batch=batch)
        ctx.sync()
    print("world", world, "spp", spp, "iterations per pass", batch, flush=True)


if __name__ == "__main__":
    main()
