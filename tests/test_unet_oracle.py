"""The oracle of the UNet tests is itself checked: oracle/ref_shim.cpp restates the pass SCHEDULE of
Cpu::Renderer::DenoiseImage(pass, region) (RendererCPU.h:790-1007) over the reference's own convolution kernels so that tests can
look at every intermediate tensor; here its final image must equal, bit for bit, what the Reference renderer itself produces
through the public API (InitUNetFilter + sixteen DenoiseImage(pass, region) calls), and the synthetic weights must be alive
(tools/gen_ref_blobs.py: deterministic pseudo-random, the trained ones are not in the tree)."""
import numpy as np
import pytest

import oracle_lib as O
from ray_amd import api, scenes

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("w,h", [(72, 40), (64, 48)])
def test_shim_schedule_equals_the_renderer(w, h):
    r, s = O.render_ref(scenes.cornell_lights, w, h, 3)
    full = r.get_raw_pixels_ref().copy()
    base = r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor).copy()
    dn = r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals).copy()
    assert r.InitUNetFilter() == 16
    region = api.RegionContext((0, 0, w, h))
    for p in range(16):
        r.DenoiseImageUNet(p, region)
    out = O.ref_unet_passes(full, base, dn, 15)
    assert np.array_equal(out, r.get_raw_pixels_ref())
    assert not np.array_equal(out[..., :3], full[..., :3])
    for p in (0, 4, 7, 13):
        t = O.ref_unet_passes(full, base, dn, p)
        assert np.isfinite(t).all() and (t > 0).mean() > 0.05 and t.max() < 100.0, p


def test_the_synthetic_weights_have_the_network_shape():
    w, off = O.ref_unet_weights()
    assert off[0] == 0 and (np.diff(off) > 0).all() and off[-1] < w.size
    assert np.isfinite(w).all() and (w != 0).sum() > 900_000  # (the blob over-allocates dec_conv1a: UNetFilter.cpp:355-358, 392)
