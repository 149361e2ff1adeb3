"""The C ABI of the product library, without a GPU: every entry point include/rayhip.h declares is exported by
librayhip.so, the ctypes mirror binds exactly that set, and the product refuses to run without a HIP device (there is
no CPU path to fall back on)."""
import ctypes
import os
import re

import pytest

from ray_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_entry_points():
    with open(os.path.join(ROOT, "include", "rayhip.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"RAYHIP_API\s+[\w\s\*]+?\b(rayhip_\w+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(hip.RAYHIP_LIB):
        pytest.skip("librayhip.so not built (run __graft_entry__.build())")
    return hip.Library()


def test_every_declared_symbol_is_exported(lib):
    names = declared_entry_points()
    assert len(names) >= 20
    raw = ctypes.CDLL(hip.RAYHIP_LIB)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_python_mirror_binds_the_declared_set(lib):
    assert sorted("rayhip_" + n for n in hip.ENTRY_POINTS) == declared_entry_points()


def test_no_device_no_render(lib):
    if lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    ctx = ctypes.c_void_p()
    rc = lib.lib.rayhip_ctx_create(0, ctypes.byref(ctx))
    assert rc != 0 and not ctx.value
    assert b"" != lib.lib.rayhip_last_error()
    with pytest.raises(Exception):
        hip.Context(0, lib)


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py is the product path: no device -> a loud exit, nothing on stdout (the contract's one JSON line is only
    ever printed after a measured run)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr
