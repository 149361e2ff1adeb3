"""The C ABI of the product library, without a GPU: every entry point include/rayhip.h declares is exported by
librayhip.so, the ctypes mirror binds exactly that set, and the product refuses to run without a HIP device (there is
no CPU path to fall back on)."""
import ctypes
import os
import re
import sys

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


def test_bench_roofline_traffic_lookup():
    """roofline.traffic comes from the committed rocprofv3 counter runs: exact for the profiled command lines, scaled by rays
    per launch for other pass sizes / rank counts of a profiled workload, absent for a workload that was never profiled"""
    sys.path.insert(0, ROOT)
    import bench
    import json
    with open(bench.traffic_table_path()) as f:
        runs = [e for e in json.load(f)["runs"] if e["workload"] == "bistro"]
    spp0, ipp0 = runs[0].get("spp", runs[0]["steps"]), runs[0]["iterations_per_pass"]  # (a frame of rounds 2-3 was `steps` iterations long)
    exact = bench.measured_traffic("bistro", spp0, ipp0, 1)
    assert exact and exact["exact"] and exact["profiled_avg_launch_ms"] > 0
    assert exact["bytes_per_launch"] == exact["fetch_bytes_per_launch"] + exact["write_bytes_per_launch"]
    # the north star's figure can be recomputed from the file: HBM bytes / launch time / 8 TB/s, a fraction
    frac = exact["bytes_per_launch"] / (exact["profiled_avg_launch_ms"] * 1e-3) / 8e12
    assert 0.05 < frac < 1.0
    nearest = min(runs, key=lambda r: abs(r["iterations_per_pass"] - 120))
    big = bench.measured_traffic("bistro", 480, 120, 1)
    assert big and not big["exact"] and abs(big["scaled_by"] - 120 / nearest["iterations_per_pass"]) < 1e-9
    assert big["scaled_from"]["iterations_per_pass"] == nearest["iterations_per_pass"]
    rank = bench.measured_traffic("bistro", spp0, ipp0, 8)
    assert rank and not rank["exact"] and abs(rank["bytes_per_launch"] * 8 - exact["bytes_per_launch"]) < 1.0
    # a profile belongs to the kernel sources it was taken with: the entry carries their hash, the lookup says whether it still holds
    assert "stale" in exact and exact["profiled_csrc_hash"] == runs[0].get("csrc_hash")
    assert exact["stale"] == (exact["profiled_csrc_hash"] != bench.csrc_hash())
    assert bench.measured_traffic("no_such_workload", 20, 20, 1) is None
    # the third profile (round 3): vector instructions per launch and the lanes they ran with -- the kernel's binding resource
    if "valu_wave_instructions_per_launch" in runs[0]:
        assert exact["valu_wave_instructions_per_launch"] > 1e6 and 1.0 <= exact["valu_active_lanes"] <= 64.0
        rate = exact["valu_wave_instructions_per_launch"] / (exact["profiled_avg_launch_ms"] * 1e-3)
        assert 0.2e12 < rate < 1.2e12  # (between a fifth of the half-rate class's peak and the full-rate class's: tools/valu_bench.hip)
        assert abs(rank["valu_wave_instructions_per_launch"] * 8 - exact["valu_wave_instructions_per_launch"]) < 1.0


def build_c_host(tmp_path):
    """examples/c_abi_render.c compiled as C99 against include/rayhip.h and linked with the product library"""
    import subprocess
    exe = os.path.join(str(tmp_path), "c_abi_render")
    build_dir = os.path.dirname(hip.RAYHIP_LIB)
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200112L", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_render.c"), "-L", build_dir, "-lrayhip", f"-Wl,-rpath,{build_dir}", "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_the_header_is_plain_c_and_a_c_host_links(lib, tmp_path):
    """the boundary is a C ABI: include/rayhip.h must compile as C99 without warnings, a host written in C must link against
    librayhip.so -- and, here, be told that there is no device instead of getting a picture from some CPU path"""
    import subprocess
    if lib.device_count() > 0:
        pytest.skip("a GPU is present: the C host is run by tests/test_gpu_parity.py")
    exe = build_c_host(tmp_path)
    golden = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([exe, os.path.join(golden, "cornell_basic.rayscene"), os.path.join(golden, "pmj02_samples.npy"), "32", "32", "1",
                        os.path.join(str(tmp_path), "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr
    assert not os.path.exists(os.path.join(str(tmp_path), "o.ppm"))


def test_the_documents_state_the_entry_point_count_of_the_header():
    """INTEGRATION.md / DESIGN.md quote the number of `rayhip_*` entry points: it must be the header's (it went stale once)"""
    import re
    n = len(declared_entry_points())
    for doc in ("INTEGRATION.md", "DESIGN.md"):
        with open(os.path.join(ROOT, doc)) as f:
            text = f.read()
        stated = [int(m) for m in re.findall(r"\((\d+) entry points", text)]
        assert stated and all(k == n for k in stated), (doc, stated, n)
