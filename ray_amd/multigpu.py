"""Tile-sharded rendering across the GPUs of one node: one process per GPU, one RCCL reduce per frame.

The reference has no multi-GPU mode; this is new.  Design:
  * the scene and the PMJ table are replicated on every GPU (a Bistro-class scene is < 1 GB of 288 GB HBM)
  * the frame is cut into 64x64 tiles, walked row-major and dealt round-robin to the ranks (rayhip_set_shard);
    each rank renders ALL iterations of ITS tiles -- no per-bounce or per-iteration communication
  * pixels are independent (RNG keyed by x, y, iteration), so every rank's RAW buffer is exact on its tiles; the shards
    are disjoint, so ONE gather of the owned tiles (1/N of the frame per rank, point-to-point to rank 0 over RCCL / xGMI)
    assembles a frame that is bit-identical to a single-GPU render
  * three transports, one packing (rayhip.h): `exchange_frame` below with a rayhip_comm (RCCL called from librayhip behind
    the C ABI -- what bench.py and a C++ host use; aux images and variance estimate included so that DenoiseImage works
    on the root), with torch.distributed on device tensors (rayhip_export_owned / rayhip_import_owned), or -- for the CPU
    tests over gloo and for ranks that share one device -- through host memory
  * rank 0 re-runs the tonemap pass on the combined frame; a rank only ever contributes the pixels it owns, so the
    combined values rank 0 now holds on foreign pixels never re-enter a later exchange: render more iterations, exchange
    again -- still exact
The same code runs on CPU tensors over gloo with the host build of the kernels (tests/test_distributed.py).
"""
from typing import Iterable, Optional

import numpy as np

from . import hip

TILE = 64


def batch_size(owned_pixels: int, limit: int, steps: int = 0, target_rays: int = 256 << 20) -> int:
    """iterations per wavefront pass (rayhip_render_batch).  Measured on one MI355X, Bistro-class 1080p: 228 Msamples/s at 1
    iteration per pass, 274 at 8, 299 at 16, 317 at 32 (then, with later kernels, 382 at 32, 392 at 48, 394 at 60; with the
    final ones 406 at 60, 416 at 120, 422 at 240) -- the fixed cost of a launch (its longest rays) and the thinly
    populated late bounces are shared by all layers.
    `limit` = Context.max_batch() (the stacked frame's coordinates are 16-bit); the target keeps the wavefront state
    (~0.25 KB per ray, plus 48 B per pixel and layer) near 75 GB of the 288 GB; with `steps` given the run is cut into passes of equal size.  A rank of a
    tile-sharded render owns 1/N of the pixels and therefore stacks N times more iterations into a pass: its launches stay
    as full as a single GPU's."""
    b = max(1, min(limit, -(-target_rays // max(owned_pixels, 1))))
    if steps > 0:
        passes = -(-steps // b)
        b = -(-steps // passes)
    return b


def render_sharded(ctx: hip.Context, iterations: Iterable[int], rank: int, world: int, dist=None, frame=None,
                   flags: int = 0, tile: int = TILE, batch: int = 1):
    """Render `iterations` of this rank's tiles, then assemble the frame on rank 0.

    frame: a [H, W, 4] float32 torch tensor used as the reduce buffer -- on the GPU of this rank for the RCCL
    path (rayhip copies into it device-to-device), or a CPU tensor for gloo.  Returns `frame` (valid on rank 0)
    or None when world == 1.
    """
    ctx.set_shard(tile, world, rank)
    its = list(iterations)
    assert its == list(range(its[0], its[0] + len(its))), "iterations must be consecutive"
    done = 0
    while done < len(its):  # bit-identical to one render() per iteration; fewer, fuller launches
        n = min(batch, len(its) - done)
        ctx.render_batch(its[done], n, flags=flags)
        done += n
    if dist is None or (world <= 1 and frame is None):
        return None  # (world == 1 with a frame: the exchange step is still run -- a one-rank reduce -- for testing)
    # the operand of the reduce: this rank's OWNED pixels, zero elsewhere -- whatever the buffers hold on other ranks'
    # pixels (the clear colour; on rank 0 the combined frame of an earlier reduce) must not enter the sum
    if frame.is_cuda:
        ctx.export_shard_device(hip.BUF_RAW, frame.data_ptr())
    else:
        import torch
        own = owned_pixel_mask(ctx.w, ctx.h, rank, world, tile)
        frame.copy_(torch.from_numpy(ctx.readback(hip.BUF_RAW) * own[..., None]))
    dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)
    if frame.is_cuda:
        # the collective is only enqueued (torch stream); librayhip works on its own stream, so wait for it here
        import torch
        torch.cuda.current_stream(frame.device).synchronize()
    if rank == 0 and frame.is_cuda:
        ctx.set_raw_device(frame.data_ptr())
    return frame


def exchange_frame(ctx: hip.Context, rank: int, world: int, comm: Optional[hip.Comm] = None, dist=None,
                   what: int = hip.REDUCE_RADIANCE, via_host: bool = False):
    """The one exchange step of the path, after the ranks rendered their tiles: rank 0 ends up with the whole frame.

    comm: a rayhip_comm of this rank (RCCL behind the C ABI) -- the product path.  Otherwise `dist` (torch.distributed) moves
    the packed tiles: device tensors over its RCCL backend, or (`via_host`) host tensors over gloo when several ranks share
    one device and RCCL therefore cannot form a communicator."""
    if comm is not None:
        comm.reduce_framebuffers(0, ctx.cam, what)
        return
    import torch
    sizes = [ctx.owned_bytes(what, world, r) for r in range(world)]
    cap = max(sizes)  # gather wants equal operands: rank 0 owns the most tiles
    on_host = ctx.L.prefix != "rayhip_"  # the host build of the kernels (tests): its "device memory" is host memory
    dev = torch.device("cpu") if on_host else torch.device("cuda", torch.cuda.current_device())
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if not on_host:
        # the fill is enqueued on torch's stream, the pack below runs on librayhip's own: without this the zeros may land last
        torch.cuda.current_stream(dev).synchronize()
    if rank != 0:
        ctx.export_owned(what, mine.data_ptr(), cap)  # (returns with the pack finished: the gather below may start)
    if via_host or on_host:
        parts = [torch.zeros(cap, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        dist.gather(mine.cpu(), parts, dst=0)
        parts = [p.to(dev) for p in parts] if rank == 0 else None
    else:
        parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        torch.cuda.current_stream(dev).synchronize()  # the collective is only enqueued; librayhip works on its own stream
    if rank == 0:
        for r in range(1, world):
            ctx.import_owned(what, r, parts[r].data_ptr(), sizes[r])
        ctx.finish_import()


def owned_pixel_mask(w: int, h: int, rank: int, world: int, tile: int = TILE) -> np.ndarray:
    """[H, W] bool mask of the pixels rank `rank` owns (same rule as rt::pixel_owned in rt_base.h)."""
    ys, xs = np.mgrid[0:h, 0:w]
    tiles_x = (w + tile - 1) // tile
    return ((ys // tile) * tiles_x + (xs // tile)) % world == rank
