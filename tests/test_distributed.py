"""N > 1 path on CPU: two processes over gloo, the host build of the kernels standing in for the GPU.

Checks the decomposition ray_amd/multigpu.py implements (tile ownership + one sum-reduce of the frame): the frame
assembled on rank 0 must be BIT-IDENTICAL to an unsharded render, and every rank must have touched only its tiles.
"""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, out_dir, batch=1, refine=0, clear=None, packed=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_lib as O2
    import util as U
    from ray_amd import hip, multigpu

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = hip.Library(O2.HOSTSIM_LIB, prefix="hostsim_")
    ctx = U.make_context(lib, "cornell_basic", w, h)
    frame = torch.zeros((h, w, 4), dtype=torch.float32)
    if clear is not None:  # whatever a rank's buffers hold on pixels it does not own must not enter the sum
        ctx.clear(clear)
    if packed:
        # the exchange the product uses: every rank's owned tiles, densely packed, gathered on rank 0 (ray_amd/multigpu.py:
        # exchange_frame -- here over gloo, with the host build's twins of rayhip_export_owned / rayhip_import_owned)
        done = 0
        for n in (spp, refine):
            if n:
                multigpu.render_sharded(ctx, range(done + 1, done + 1 + n), rank, world, tile=32, batch=batch)
                done += n
                multigpu.exchange_frame(ctx, rank, world, dist=dist, what=hip.REDUCE_RADIANCE | hip.REDUCE_BASE_COLOR)
                if rank == 0:
                    np.savez(os.path.join(out_dir, f"packed_{done}.npz"), raw=ctx.readback(hip.BUF_RAW), final=ctx.readback(hip.BUF_FINAL),
                             base=ctx.readback(hip.BUF_BASE_COLOR))
        dist.barrier()
        dist.destroy_process_group()
        return
    multigpu.render_sharded(ctx, range(1, spp + 1), rank, world, dist=dist, frame=frame, tile=32, batch=batch)
    part = ctx.readback(hip.BUF_RAW)
    mask = multigpu.owned_pixel_mask(w, h, rank, world, tile=32)
    if clear is None:
        assert not part[~mask].any(), "a rank wrote pixels it does not own"
    assert part[mask][..., 3].any()
    if rank == 0:
        np.save(os.path.join(out_dir, "frame.npy"), frame.numpy())
    if refine:  # progressive refinement: more iterations on the same contexts, one more reduce
        multigpu.render_sharded(ctx, range(spp + 1, spp + 1 + refine), rank, world, dist=dist, frame=frame, tile=32, batch=batch)
        if rank == 0:
            np.save(os.path.join(out_dir, "frame_refined.npy"), frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not O.have_hostsim(), reason="tests/hostsim not built")
def test_two_ranks_gloo_bit_identical(tmp_path):
    import torch.multiprocessing as mp

    w, h, spp, world = 96, 80, 2, 2
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, str(tmp_path)), nprocs=world, join=True)
    from ray_amd import hip

    lib = hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")
    full = util.render_frames(util.make_context(lib, "cornell_basic", w, h), spp)
    got = np.load(os.path.join(str(tmp_path), "frame.npy"))
    assert np.array_equal(got, full)


@pytest.mark.skipif(not O.have_hostsim(), reason="tests/hostsim not built")
def test_three_ranks_batched_gloo_bit_identical(tmp_path):
    """a world size that does not divide the tile count, a frame that is not a whole number of tiles, and every rank
    stacking its iterations into one wavefront pass (what bench.py does): still the single-process frame, bit for bit"""
    import torch.multiprocessing as mp

    w, h, spp, world = 100, 72, 3, 3
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, str(tmp_path), spp), nprocs=world, join=True)
    from ray_amd import hip

    lib = hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")
    full = util.render_frames(util.make_context(lib, "cornell_basic", w, h), spp)
    got = np.load(os.path.join(str(tmp_path), "frame.npy"))
    assert np.array_equal(got, full)


@pytest.mark.skipif(not O.have_hostsim(), reason="tests/hostsim not built")
def test_two_ranks_refine_after_a_reduce_and_nonzero_clear_colour(tmp_path):
    """render, reduce, render more, reduce again -- with the buffers cleared to a non-zero colour first, so that every
    rank holds non-zero values on pixels it does not own: only owned pixels may enter the sums"""
    import torch.multiprocessing as mp

    w, h, spp, refine, world = 96, 80, 2, 2, 2
    clear = (0.25, 0.5, 0.125, 1.0)
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, str(tmp_path), 1, refine, clear), nprocs=world, join=True)
    from ray_amd import hip

    lib = hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")
    ctx = util.make_context(lib, "cornell_basic", w, h)
    ctx.clear(clear)
    full = util.render_frames(ctx, spp)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "frame.npy")), full)
    for it in range(spp + 1, spp + refine + 1):
        ctx.render(it)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "frame_refined.npy")), ctx.readback(hip.BUF_RAW))


@pytest.mark.skipif(not O.have_hostsim(), reason="tests/hostsim not built")
@pytest.mark.parametrize("world,w,h", [(2, 96, 80), (3, 100, 72)])
def test_packed_tile_gather_over_gloo_is_bit_identical(tmp_path, world, w, h):
    """the product's exchange (owned tiles, densely packed, gathered on rank 0 -- rayhip_export_owned / rayhip_import_owned /
    rayhip_finish_import; ragged frames, a world size that does not divide the tile count, a non-zero clear colour, a second
    exchange after more iterations): radiance, tonemapped and base-colour images equal a single-process render bit for bit"""
    import torch.multiprocessing as mp

    spp, refine = 2, 2
    clear = (0.25, 0.5, 0.125, 1.0)
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, str(tmp_path), 1, refine, clear, True), nprocs=world, join=True)
    from ray_amd import hip

    lib = hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")
    ctx = util.make_context(lib, "cornell_basic", w, h)
    ctx.clear(clear)
    for done in (spp, spp + refine):
        for it in range(1 if done == spp else spp + 1, done + 1):
            ctx.render(it)
        got = np.load(os.path.join(str(tmp_path), f"packed_{done}.npz"))
        assert np.array_equal(got["raw"], ctx.readback(hip.BUF_RAW)), done
        assert np.array_equal(got["final"], ctx.readback(hip.BUF_FINAL)), done
        assert np.array_equal(got["base"], ctx.readback(hip.BUF_BASE_COLOR)), done
