"""The drop-in on the device: the reference's own samples/00_basic/main.cpp -- not a line changed -- built against the
reference tree with the INTEGRATION.md patch applied (ray_amd/host/dropin/build_dropin.py, at build time, where the reference
tree exists; the binary ships to the GPU box).  `Ray::CreateRenderer(s, &Ray::g_stdout_log)` with its default arguments must
hand back the HIP renderer, and the TGA the sample writes must be the picture the ctypes path (ray_amd/api.py over the same
RendererHIP) renders for the same scene: 256 x 256, 64 iterations, tonemapped, 8 bits per channel."""
import os
import subprocess

import numpy as np
import pytest

from ray_amd import api, hip, scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "ray_amd", "host", "_build", "dropin", "sample_00_basic")


def _read_tga(path):
    with open(path, "rb") as f:
        data = f.read()
    w, h, bpp = int.from_bytes(data[12:14], "little"), int.from_bytes(data[14:16], "little"), data[16]
    assert data[2] == 2 and bpp == 24
    img = np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=18).reshape(h, w, 3)
    assert data[17] & 0x20  # origin in the upper left corner: rows top-down, as in memory
    return img[:, :, ::-1]  # BGR -> RGB


def test_the_reference_sample_runs_on_the_hip_renderer(tmp_path):
    assert hip.Library().device_count() > 0, "no HIP device: the product has no CPU path"
    if not os.path.exists(EXE):
        pytest.fail("ray_amd/host/_build/dropin/sample_00_basic is missing (built by __graft_entry__.build() where the reference tree exists)")
    r = subprocess.run([EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    print(r.stdout[:1500])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Ray: Creating HIP renderer 256x256" in r.stdout
    assert "Failed to create HIP renderer" not in r.stdout and "Creating Reference renderer" not in r.stdout
    assert "gfx950" in r.stdout
    got = _read_tga(os.path.join(str(tmp_path), "00_basic.tga")).astype(np.int32)

    rr = api.CreateRenderer(api.Settings(256, 256), "HIP")
    s = rr.CreateScene()
    scenes.cornell_basic(s)
    region = api.RegionContext((0, 0, 256, 256))
    for _ in range(64):
        rr.RenderScene(s, region)
    px = rr.get_pixels_ref()[..., :3]
    # WriteTGA of the sample: float_to_byte per channel
    q = np.where(px <= 0.0, 0, np.where(px > 1.0 - 0.5 / 255.0, 255, (255.0 * px + 0.5).astype(np.int32))).astype(np.int32)
    d = np.abs(got - q)
    print("pixels differing:", int((d.max(axis=-1) > 0).sum()), "max", int(d.max()))
    assert d.max() <= 1 and (d.max(axis=-1) > 0).mean() < 0.01
