"""RendererHIP over several devices behind the UNCHANGED Ray API (RAY_HIP_DEVICES / settings_t::preferred_device): the scene
is replicated, every RenderScene is dealt out in 64 x 64 tiles, and whatever looks at pixels (get_*_pixels_ref,
DenoiseImage) first gathers the other ranks' tiles on the root (rayhip_comm_create: one process, peer copies).  A one-GPU
box runs it with the SAME device listed several times -- every rank is its own context with its own buffers and stream, only
the copies stay on one device -- and the pictures must be the single-device renderer's, bit for bit: radiance, tonemapped,
both aux images, the NLM-filtered frame (which needs every rank's variance estimate), across Clear, more iterations after a
gather, and a scene mutation that goes through the instance-update path on every rank."""
import os

import numpy as np
import pytest

from ray_amd import api, hip, scenes

pytestmark = pytest.mark.gpu


def _need_host_lib():
    if hip.Library().device_count() <= 0:
        pytest.fail("no HIP device: the product has no CPU path")
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.fail("ray_amd/host/_build/libray_hip.so is missing (RendererHIP)")


def _images(r):
    return {"raw": r.get_raw_pixels_ref().copy(), "final": r.get_pixels_ref().copy(),
            "base": r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor).copy(),
            "dn": r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals).copy()}


def _run(devices, w, h, monkeypatch):
    """the same call sequence on a renderer over `devices`; returns the pictures seen at each step"""
    if devices:
        monkeypatch.setenv("RAY_HIP_DEVICES", devices)
    else:
        monkeypatch.delenv("RAY_HIP_DEVICES", raising=False)
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.cornell_instances_mutable(s)
    seen = []
    region = api.RegionContext((0, 0, w, h))
    for _ in range(3):
        r.RenderScene(s, region)
    seen.append(_images(r))          # gather no. 1
    for _ in range(2):
        r.RenderScene(s, region)     # more iterations on top of what every rank holds
    seen.append(_images(r))          # gather no. 2: the root's stale copies of foreign tiles must be replaced
    r.DenoiseImage(region)           # needs the variance estimate of every tile on the root
    seen.append(_images(r))
    scenes.mutate_instances_scene(s)  # instances / lights / environment only: the update path, on every rank
    r.Clear()
    region = api.RegionContext((0, 0, w, h))
    for _ in range(3):
        r.RenderScene(s, region)
    seen.append(_images(r))
    return seen, r.device_name()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_renderer_hip_over_several_ranks_equals_one_device(devices, monkeypatch):
    _need_host_lib()
    w, h = 200, 136  # not a multiple of the 64-pixel shard tile
    one, name1 = _run("", w, h, monkeypatch)
    many, name_n = _run(devices, w, h, monkeypatch)
    print(name1, "|", name_n)
    assert name_n.endswith(f"x{devices.count(',') + 1}")
    for step, (a, b) in enumerate(zip(one, many)):
        for key in a:
            assert np.array_equal(a[key], b[key]), (step, key)
