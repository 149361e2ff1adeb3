"""One process per GPU over the product's RCCL transport (comm.hip.h: rayhip_comm_create_rank + ncclSend / ncclRecv): N = 2 ... all
visible devices, frames bit-identical to an unsharded render.  A one-GPU box cannot form such a communicator (RCCL refuses two
ranks on one device), so the test skips there -- it is written for the node the multi-GPU bench runs on.
"""
import os
import subprocess
import sys

import pytest

from ray_amd import hip

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _device_count() -> int:
    return hip.Library().device_count()


@pytest.mark.parametrize("ranks", [2, 8])
def test_ranks_in_separate_processes_gather_over_rccl(ranks):
    have = _device_count()
    if have < ranks:
        pytest.skip(f"{ranks} RCCL ranks need {ranks} devices, {have} visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + ranks), os.path.join(HERE, "rccl_ranks_worker.py")]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    sys.stderr.write(done.stderr[-4000:])
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert f"RCCL_RANKS_OK {ranks}" in done.stdout
    # every rank reported what RCCL itself said about the communicator
    for r in range(ranks):
        assert f"RCCL communicator up: rank {r} of {ranks}" in done.stderr
