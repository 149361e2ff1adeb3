"""Tile-sharded rendering across the GPUs of one node: one process per GPU, one RCCL reduce per frame.

The reference has no multi-GPU mode; this is new.  Design:
  * the scene and the PMJ table are replicated on every GPU (a Bistro-class scene is < 1 GB of 288 GB HBM)
  * the frame is cut into 64x64 tiles, walked row-major and dealt round-robin to the ranks (rayhip_set_shard);
    each rank renders ALL iterations of ITS tiles -- no per-bounce or per-iteration communication
  * pixels are independent (RNG keyed by x, y, iteration), so every rank's RAW buffer is exact on its tiles and
    zero elsewhere; ONE sum-reduce of the W*H*4 fp32 frame to rank 0 (torch.distributed, backend "nccl" == RCCL
    over xGMI; 33 MB at 1080p) assembles a frame that is bit-identical to a single-GPU render
  * rank 0 re-runs the tonemap pass on the combined frame (rayhip_set_raw_device); a rank only ever contributes the
    pixels it owns (rayhip_export_shard_device), so the combined values rank 0 now holds on foreign pixels never re-enter
    a later reduce: render more iterations, reduce again -- still exact
  * C++ hosts get the same exchange behind the C ABI (rayhip_comm_*: RCCL called from librayhip, aux images and variance
    estimate included so that DenoiseImage works on the root)
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


def owned_pixel_mask(w: int, h: int, rank: int, world: int, tile: int = TILE) -> np.ndarray:
    """[H, W] bool mask of the pixels rank `rank` owns (same rule as rt::pixel_owned in rt_base.h)."""
    ys, xs = np.mgrid[0:h, 0:w]
    tiles_x = (w + tile - 1) // tile
    return ((ys // tile) * tiles_x + (xs // tile)) % world == rank
