"""The frame exchange of the tile-sharded render on real hardware (one GPU box: one RCCL rank, several emulated shards).

  * rayhip_comm_* (RCCL called from librayhip behind the C ABI): a one-rank communicator runs the whole path -- pack the
    owned pixels, ncclReduce, unpack on the root, re-tonemap -- for the radiance image, both aux images and the variance
    estimate, repeatedly (progressive refinement), and must reproduce an unsharded render bit for bit;
  * the pack step with TWO shards on one GPU: the sum of what two contexts export equals the unsharded frame, also after
    the root was handed the combined frame and both rendered more iterations (the stale-combined-frame hazard);
  * the torch.distributed form bench.py uses (process group over RCCL with one rank): render_sharded twice.
"""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (FIRST: torch brings its own HIP runtime, and it must be the one that opens the device -- librayhip, loaded later,
#               shares it; the other way round torch finds "no HIP GPUs" -- which is what this file did when it ran on its own)

import util
from ray_amd import hip, multigpu

pytestmark = pytest.mark.gpu

NAME = "cornell_lights"
W, H = 200, 136  # not a multiple of the 64-pixel shard tile


@pytest.fixture(scope="module")
def lib():
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path"
    return lib


def _reference(lib, n_iter):
    ctx = util.make_context(lib, NAME, W, H)
    ctx.render_batch(1, n_iter)
    return {k: ctx.readback(b) for k, b in (("raw", hip.BUF_RAW), ("final", hip.BUF_FINAL), ("base", hip.BUF_BASE_COLOR),
                                            ("dn", hip.BUF_DEPTH_NORMALS), ("var", hip.BUF_VARIANCE))}


def test_comm_reduce_with_one_rank_is_exact_and_repeatable(lib):
    ctx = util.make_context(lib, NAME, W, H)
    comm = hip.Comm(lib, [0])
    comm.bind(0, ctx)
    done = 0
    for n in (3, 2):  # render, reduce, render more, reduce again
        ctx.render_batch(done + 1, n)
        done += n
        comm.reduce_framebuffers(0, ctx.cam)
        ref = _reference(lib, done)
        for k, b in (("raw", hip.BUF_RAW), ("final", hip.BUF_FINAL), ("base", hip.BUF_BASE_COLOR), ("dn", hip.BUF_DEPTH_NORMALS),
                     ("var", hip.BUF_VARIANCE)):
            assert np.array_equal(ctx.readback(b), ref[k]), (k, done)
    # radiance only
    ctx.render(done + 1)
    comm.reduce_framebuffers(0, ctx.cam, hip.REDUCE_RADIANCE)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), _reference(lib, done + 1)["raw"])
    comm.close()


def test_two_shards_on_one_gpu_sum_to_the_frame_across_refinement_steps(lib):
    import torch
    ctxs = [util.make_context(lib, NAME, W, H) for _ in range(2)]
    for r, c in enumerate(ctxs):
        c.set_shard(64, 2, r)
    parts = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    done = 0
    for n in (2, 3):
        for c in ctxs:
            c.render_batch(done + 1, n)
        done += n
        for which, key in ((hip.BUF_RAW, "raw"), (hip.BUF_BASE_COLOR, "base"), (hip.BUF_DEPTH_NORMALS, "dn"), (hip.BUF_VARIANCE, "var")):
            for c, p in zip(ctxs, parts):
                c.export_shard_device(which, p.data_ptr())
            total = (parts[0] + parts[1])
            assert np.array_equal(total.cpu().numpy(), _reference(lib, done)[key]), (key, done)
            if which == hip.BUF_RAW:
                frame = total.clone()
        # the root now holds the COMBINED frame in its buffers; the next round must not pick it up from there
        ctxs[0].set_raw_device(frame.data_ptr())
        assert np.array_equal(ctxs[0].readback(hip.BUF_RAW), _reference(lib, done)["raw"])
        assert np.array_equal(ctxs[0].readback(hip.BUF_FINAL), _reference(lib, done)["final"])


def test_packed_tiles_of_three_shards_on_one_gpu_assemble_the_frame(lib):
    """the packing of the product's exchange on the device (rayhip_export_owned -> rayhip_import_owned -> rayhip_finish_import):
    three contexts on one GPU stand in for three ranks, the root imports the other two ranks' tiles -- every image of the
    mask, a ragged frame, a second round after more iterations"""
    import torch
    n = 3
    ctxs = [util.make_context(lib, NAME, W, H) for _ in range(n)]
    for r, c in enumerate(ctxs):
        c.set_shard(64, n, r)
    done = 0
    for k in (2, 3):
        for c in ctxs:
            c.render_batch(done + 1, k)
        done += k
        for r in range(1, n):
            nbytes = ctxs[r].owned_bytes(hip.REDUCE_ALL, n, r)
            assert nbytes == ctxs[0].owned_bytes(hip.REDUCE_ALL, n, r) and nbytes % (64 * 64 * 16 * 4) == 0
            buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()  # (the fill runs on torch's stream, the pack on librayhip's own: without this the zeros may land last)
            ctxs[r].export_owned(hip.REDUCE_ALL, buf.data_ptr(), nbytes)
            ctxs[0].import_owned(hip.REDUCE_ALL, r, buf.data_ptr(), nbytes)
        ctxs[0].finish_import()
        ref = _reference(lib, done)
        for key, b in (("raw", hip.BUF_RAW), ("final", hip.BUF_FINAL), ("base", hip.BUF_BASE_COLOR), ("dn", hip.BUF_DEPTH_NORMALS),
                       ("var", hip.BUF_VARIANCE)):
            assert np.array_equal(ctxs[0].readback(b), ref[key]), (key, done)
    with pytest.raises(RuntimeError):  # a buffer that is too small is refused, not overrun
        ctxs[1].export_owned(hip.REDUCE_ALL, buf.data_ptr(), 16)


def test_render_sharded_over_torch_distributed_twice(lib):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = util.make_context(lib, NAME, W, H)
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        multigpu.render_sharded(ctx, range(1, 4), 0, 1, dist=dist, frame=frame, batch=3)
        assert np.array_equal(frame.cpu().numpy(), _reference(lib, 3)["raw"])
        multigpu.render_sharded(ctx, range(4, 6), 0, 1, dist=dist, frame=frame, batch=2)
        assert np.array_equal(frame.cpu().numpy(), _reference(lib, 5)["raw"])
        assert np.array_equal(ctx.readback(hip.BUF_FINAL), _reference(lib, 5)["final"])
    finally:
        if created:
            dist.destroy_process_group()
