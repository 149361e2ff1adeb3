"""One rank of the multi-process RCCL check (started by test_gpu_rccl_processes.py under torch.distributed.run, one process per GPU).

Every rank renders its tiles of a small fixture scene, the product's own exchange step (rayhip_comm_reduce_framebuffers: ncclSend /
ncclRecv behind the C ABI) brings them to rank 0, and rank 0 compares every image with an unsharded render of its own, bit for bit.
torch.distributed (gloo) only hands the RCCL id around and holds the final barrier.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import util  # noqa: E402
from ray_amd import hip  # noqa: E402

NAME, W, H, ITERATIONS = "cornell_lights", 200, 136, 5
IMAGES = (("raw", hip.BUF_RAW), ("final", hip.BUF_FINAL), ("base", hip.BUF_BASE_COLOR), ("dn", hip.BUF_DEPTH_NORMALS),
          ("var", hip.BUF_VARIANCE))


def main() -> int:
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lib = hip.Library()
    devices = lib.device_count()
    # tests/fake_rccl (loaded by librayhip through RAYHIP_RCCL_LIB) moves messages through host memory and does not care which device a
    # rank sits on: the ranks may then share devices -- the way the product's transport code runs with N > 1 on a one-GPU box
    stand_in = "fake_rccl" in os.environ.get("RAYHIP_RCCL_LIB", "")
    assert stand_in or devices >= world, f"{world} ranks need {world} devices, {devices} visible (RCCL refuses two ranks on one device)"
    ctx = util.make_context(lib, NAME, W, H, device=int(os.environ.get("LOCAL_RANK", rank)) % devices)
    ctx.set_shard(64, world, rank)
    ids = [hip.Comm.unique_id(lib) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    comm = hip.Comm.for_rank(lib, ids[0], world, rank, ctx)  # (logs "RCCL communicator up: rank r of N" to stderr)
    bad = []
    for first, count in ((1, 3), (4, ITERATIONS - 3)):  # twice: the root holds a combined frame when the second round starts
        ctx.render_batch(first, count)
        comm.reduce_framebuffers(0, ctx.cam, hip.REDUCE_ALL)
        if rank == 0:
            whole = util.make_context(lib, NAME, W, H, device=0)
            whole.render_batch(1, first + count - 1)
            for key, buf in IMAGES:
                if not np.array_equal(ctx.readback(buf), whole.readback(buf)):
                    bad.append((key, first + count - 1))
    dist.barrier()
    if rank == 0:
        print("RCCL_RANKS_OK" if not bad else f"RCCL_RANKS_DIFFER {bad}", world, flush=True)
    dist.destroy_process_group()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
