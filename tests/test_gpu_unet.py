"""The UNet denoiser on the matrix cores (ray_amd/csrc/unet_kernels.hip behind rayhip_unet_init / rayhip_denoise_unet) against the
reference's own convolution kernels (oracle: Ref::Convolution3x3 / ConvolutionConcat3x3 driven by the schedule of
Cpu::Renderer::DenoiseImage(pass, region), RendererCPU.h:790-1007 -- oracle/ref_shim.cpp: refk_unet_passes, checked bit for bit
against the renderer's own sixteen passes in tests/test_unet_oracle.py).

The trained OIDN weights are not part of the reference tree; tools/gen_ref_blobs.py fills the weight header with deterministic
pseudo-random half-precision values of the right shapes, and BOTH sides compute with them (the product takes them from the
reference's SetupUNetWeights).  What is checked is therefore the arithmetic of all sixteen passes -- 3 x 3 convolution with
bias and ReLU, 2 x 2 max-pooling, nearest-neighbour upsampling and concatenation, the HDR transfer function and its inverse,
zero borders, frames that are not a multiple of 16 -- not the visual quality of a trained network.

Tolerance: the kernels use the f32 matrix instruction (f32 in, f32 accumulate), so the only difference to the reference's
fp32 loops is the ORDER of the ~100..1500 additions per output (and the device's powf / logf / expf in the transfer
functions): every tensor within 2e-5 * max(1, |ref|) of the oracle's, pass by pass on the oracle's own inputs.
"""
import time

import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import api, hip, scenes

pytestmark = pytest.mark.gpu

TOL = 2e-5
# tensor written by pass p (index into unet_filter_tensors_t order; pass 15 writes the image)
OUT_TENSOR = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]


def _ctx(w, h, spp=4):
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path"
    if not O.have_ref():
        pytest.fail("oracle/_ref/libray_ref.so is missing on the GPU box")
    ctx = util.make_context(lib, "cornell_lights", w, h)
    ctx.render_batch(1, spp)
    weights, offsets = O.ref_unet_weights()
    ctx.unet_init(weights, offsets, 8)
    return ctx


@pytest.mark.parametrize("w,h", [(200, 136), (64, 48)])
def test_every_pass_against_the_reference_convolutions(w, h):
    ctx = _ctx(w, h)
    full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
    final_before = ctx.readback(hip.BUF_FINAL)
    worst = 0.0
    for p in range(15):
        ctx.denoise_unet(p)
        got = ctx.unet_read_tensor(OUT_TENSOR[p])
        ref = O.ref_unet_passes(full, base, dn, p)
        assert got.shape == ref.shape, (p, got.shape, ref.shape)
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        worst = max(worst, float(err.max()))
        assert err.max() <= TOL, (p, float(err.max()), np.unravel_index(err.argmax(), err.shape))
        # the one-pixel border stays zero (the reference clears it after every pass)
        assert not got[0].any() and not got[-1].any() and not got[:, 0].any() and not got[:, -1].any(), p
        assert (ref > 0).mean() > 0.05, "a dead tensor would make the comparison meaningless"
    ctx.denoise_unet(15)
    got = ctx.readback(hip.BUF_RAW)
    ref = O.ref_unet_passes(full, base, dn, 15)
    err = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))
    print(f"{w}x{h}: worst relative error over the tensors {worst:.2e}, filtered image {err.max():.2e}")
    assert err.max() <= 5 * TOL  # (the inverse HDR transfer is an exponential: it stretches the last bits)
    assert np.array_equal(got[..., 3], full[..., 3])  # alpha is not the network's business
    assert not np.array_equal(ctx.readback(hip.BUF_FINAL), final_before)  # FINAL = Tonemap(RAW) was redone


def test_all_passes_in_one_call_and_a_second_frame_size():
    ctx = _ctx(200, 136)
    full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
    ctx.denoise_unet(-1)
    ref = O.ref_unet_passes(full, base, dn, 15)
    got = ctx.readback(hip.BUF_RAW)
    assert (np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))).max() <= 5 * TOL
    # a resize re-sizes the tensors (and their zero borders)
    ctx.resize(96, 80)
    ctx.render_batch(1, 2)
    full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
    ctx.denoise_unet(-1)
    ref = O.ref_unet_passes(full, base, dn, 15)
    got = ctx.readback(hip.BUF_RAW)
    assert (np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))).max() <= 5 * TOL


def test_renderer_hip_unet_through_the_ray_api():
    """InitUNetFilter + DenoiseImage(pass, region) x 16 behind the Ray API, against the Reference renderer doing the same on
    ITS frame (the two frames differ in the last bits, and the random network amplifies that: a looser bar)"""
    import os
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.fail("ray_amd/host/_build/libray_hip.so is missing")
    w, h, spp = 96, 64, 4
    ref, rs = O.render_ref(scenes.cornell_lights, w, h, spp)
    assert ref.InitUNetFilter() == 16
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.cornell_lights(s)
    region = api.RegionContext((0, 0, w, h))
    for _ in range(spp):
        r.RenderScene(s, region)
    assert r.InitUNetFilter() == 16
    ref_region = api.RegionContext((0, 0, w, h))
    for p in range(16):
        r.DenoiseImageUNet(p, region)
        ref.DenoiseImageUNet(p, ref_region)
    a, b = r.get_raw_pixels_ref(), ref.get_raw_pixels_ref()
    err = np.abs(a[..., :3] - b[..., :3]) / np.maximum(1.0, np.abs(b[..., :3]))
    print("RendererHIP UNet vs RendererRef UNet:", float(err.max()), float(err.mean()))
    assert err.max() <= 2e-2 and err.mean() <= 1e-4
    m = util.frame_metrics(r.get_pixels_ref(), ref.get_pixels_ref())
    assert m["frac_within"] >= 0.99, m


def test_unet_time_at_1080p():
    """sixteen passes on a 1920 x 1080 frame (informational: printed, with a generous ceiling)"""
    ctx = _ctx(1920, 1080, spp=1)
    ctx.denoise_unet(-1)  # (tensors allocated, kernels loaded)
    ctx.sync()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        ctx.denoise_unet(-1)
    ctx.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    flops = 2 * 125406 * 1920 * 1080  # multiply-adds per pixel of the sixteen convolutions x 2
    print(f"UNet 1080p: {ms:.2f} ms per frame, {flops / ms / 1e9:.1f} TFLOP/s (f32 matrix peak 157)")
    assert ms < 100.0
