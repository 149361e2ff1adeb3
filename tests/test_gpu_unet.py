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


@pytest.mark.parametrize("f32", [True, False])
def test_renderer_hip_unet_through_the_ray_api(f32, monkeypatch):
    """InitUNetFilter + DenoiseImage(pass, region) x 16 behind the Ray API, against the Reference renderer doing the same on
    ITS frame (the two frames differ in the last bits, and the random network amplifies that: a looser bar).  RendererHIP runs the
    f16 form by default, as the reference's GPU backends do on hardware with half-precision matrix arithmetic; RAY_HIP_UNET_F32=1
    keeps the exact form."""
    import os
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.fail("ray_amd/host/_build/libray_hip.so is missing")
    if f32:
        monkeypatch.setenv("RAY_HIP_UNET_F32", "1")
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
    print(f"RendererHIP UNet ({'f32' if f32 else 'f16'} form) vs RendererRef UNet:", float(err.max()), float(err.mean()))
    if f32:
        assert err.max() <= 2e-2 and err.mean() <= 1e-4
    else:
        assert err.max() <= 1e-1 and err.mean() <= 2e-3
    m = util.frame_metrics(r.get_pixels_ref(), ref.get_pixels_ref())
    print("tone-mapped image:", m)
    assert (m["frac_within"] >= 0.99) if f32 else (m["psnr"] >= 45.0), m  # (a half resolves 1e-3 of its value: the 1e-3 band is the f32 form's bar)


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


# ---- the f16 form (rayhip_unet_set_precision(1): f16 tensors and weights, f32 accumulators, v_mfma_f32_16x16x32_f16) ------------------------------
# Bound: a half has 11 significant bits (unit round-off 4.9e-4).  Every pass rounds its inputs (the previous pass's outputs) and its weights to
# halves -- the generated stand-in weights ARE halves, so only the activations lose bits -- and accumulates 144 .. 1440 products in f32; the errors
# of a pass feed the next one, fifteen deep.  Stated per tensor as  max |got - ref| <= F16_TOL * max(1, max |ref|)  (relative to the tensor's
# scale: a half cannot resolve an activation of 1e-3 next to one of 30 any better) and as a mean relative error; the final image additionally as
# PSNR against the f32 form's image.
F16_TOL = 4e-3
F16_MEAN_TOL = 2e-3


@pytest.mark.parametrize("w,h", [(200, 136), (64, 48)])
def test_f16_form_every_pass_within_a_half_precision_bound(w, h):
    ctx = _ctx(w, h)
    full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
    ctx.unet_precision(True)
    lines, bad = [], False
    for p in range(15):
        ctx.denoise_unet(p)
        got = ctx.unet_read_tensor(OUT_TENSOR[p])
        ref = O.ref_unet_passes(full, base, dn, p)
        assert got.shape == ref.shape, (p, got.shape, ref.shape)
        scale = max(1.0, float(np.abs(ref).max()))
        worst, mean = float(np.abs(got - ref).max()) / scale, float(np.abs(got - ref).mean()) / max(float(np.abs(ref).mean()), 1e-12)
        lines.append(f"pass {p:2d}: max |ref| {float(np.abs(ref).max()):9.3f}  max err / scale {worst:.2e}  mean err / mean |ref| {mean:.2e}")
        assert np.isfinite(got).all(), p
        bad = bad or not (worst <= F16_TOL and mean <= F16_MEAN_TOL)
        assert not got[0].any() and not got[-1].any() and not got[:, 0].any() and not got[:, -1].any(), p
    ctx.denoise_unet(15)
    got = ctx.readback(hip.BUF_RAW)
    ref = O.ref_unet_passes(full, base, dn, 15)
    err = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))
    print("\n".join(lines))
    assert not bad, "\n".join(lines)
    print(f"{w}x{h} f16 form: filtered image max rel err {err.max():.2e}, mean {err.mean():.2e}")
    assert err.max() <= 1e-1 and err.mean() <= 2e-3  # (the inverse HDR transfer is an exponential: it stretches what the last tensor lost)
    assert np.array_equal(got[..., 3], full[..., 3])
    # against the exact form on the same frame: PSNR of the tone-mapped image
    final_h = ctx.readback(hip.BUF_FINAL)
    ctx.unet_precision(False)
    ctx.denoise_unet(-1)  # (pass 0 reads `full`, which the last pass left alone: RAW is the output, the running mean the input)
    m = util.frame_metrics(final_h, ctx.readback(hip.BUF_FINAL))
    print(f"{w}x{h}: f16 form against the f32 form, FINAL image: PSNR {m['psnr']:.1f} dB, within tolerance {m['frac_within']:.4f}")
    assert m["psnr"] >= 50.0, m


def test_f16_form_all_passes_in_one_call_rects_and_switching_back():
    ctx = _ctx(200, 136)
    full, base, dn = ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS)
    ref = O.ref_unet_passes(full, base, dn, 15)
    ctx.unet_precision(True)
    ctx.denoise_unet(-1)
    got_h = ctx.readback(hip.BUF_RAW)
    assert (np.abs(got_h[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))).max() <= 1e-1  # (the inverse HDR transfer is an exponential)
    ctx.unet_precision(False)  # the exact form is still exact after the f16 tensors have been in use
    ctx.denoise_unet(-1)
    got = ctx.readback(hip.BUF_RAW)
    assert (np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))).max() <= 5 * TOL
    assert not np.array_equal(got, got_h)


def test_f16_form_stays_finite_on_very_bright_pixels():
    """HDR stress: radiance far beyond the half range goes through the HDR transfer function before it becomes a tensor, and
    the f16 tensors saturate at 65504 instead of overflowing -- the denoised frame of the f16 form (RendererHIP's default, like the reference's
    GPU backends) must be finite everywhere"""
    import torch
    ctx = _ctx(200, 136)
    frame = torch.from_numpy(ctx.readback(hip.BUF_RAW).copy()).cuda()
    frame[..., :3] *= 1.0e4
    frame[40:60, 50:90, :3] = 3.0e6   # a block of pixels three million times white
    frame[100, 120, :3] = 6.5e9
    ctx.set_raw_device(frame.data_ptr())
    torch.cuda.synchronize()
    for half in (True, False):
        ctx.set_raw_device(frame.data_ptr())
        ctx.unet_precision(half)
        ctx.denoise_unet(-1)
        out = ctx.readback(hip.BUF_RAW)
        assert np.isfinite(out).all(), ("f16" if half else "f32", int((~np.isfinite(out)).sum()))
        assert float(out[..., :3].max()) > 1.0e3  # (the bright block is still bright)


def test_f16_unet_time_at_1080p():
    """the f16 form's sixteen passes on a 1920 x 1080 frame (task 6: <= 1.0 ms asked for; the ceiling here is generous,
    the number is printed and profiled under profiles/r05/)"""
    ctx = _ctx(1920, 1080, spp=1)
    out = {}
    for half in (False, True):
        ctx.unet_precision(half)
        ctx.denoise_unet(-1)
        ctx.sync()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            ctx.denoise_unet(-1)
        ctx.sync()
        out[half] = (time.perf_counter() - t0) / n * 1e3
    flops = 2 * 125406 * 1920 * 1080
    print(f"UNet 1080p: f32 form {out[False]:.2f} ms ({flops / out[False] / 1e9:.1f} TFLOP/s), f16 form {out[True]:.2f} ms "
          f"({flops / out[True] / 1e9:.1f} TFLOP/s of the network's own arithmetic)")
    assert out[True] < out[False]
