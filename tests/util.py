"""Shared helpers for the parity tests."""
import os

import numpy as np

from ray_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Stated per-pixel tolerance of the HIP backend against RendererRef (BASELINE.md section 3 / SURVEY.md section 8c),
# calibrated there by rebuilding the reference itself with fma contraction:
#   >= 99.5 % of pixels with max-channel |d| <= 1e-3 * max(1, |ref|)
#   PSNR on linear values clamped to [0,1]: >= 55 dB at 1 spp, >= 70 dB at >= 64 spp
TOL_REL = 1e-3
MIN_FRACTION = 0.995
MIN_PSNR_1SPP = 55.0
MIN_PSNR_8SPP = 60.0
MIN_PSNR_64SPP = 70.0


def pmj() -> np.ndarray:
    return np.load(os.path.join(GOLDEN, "pmj02_samples.npy"))


def golden_scene(name: str) -> bytes:
    with open(os.path.join(GOLDEN, f"{name}.rayscene"), "rb") as f:
        return f.read()


def golden_ref(name: str):
    return np.load(os.path.join(GOLDEN, f"{name}_ref.npz"))


def frame_metrics(img: np.ndarray, ref: np.ndarray) -> dict:
    """img/ref: [H, W, 4] raw linear fp32.  Returns the quantities the tolerance is stated in."""
    d = np.abs(img[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)).max(axis=-1)
    scale = np.maximum(1.0, np.abs(ref[..., :3]).max(axis=-1))
    within = d <= TOL_REL * scale
    a, b = np.clip(img[..., :3], 0.0, 1.0).astype(np.float64), np.clip(ref[..., :3], 0.0, 1.0).astype(np.float64)
    mse = float(((a - b) ** 2).mean())
    psnr = 200.0 if mse == 0.0 else float(-10.0 * np.log10(mse))
    return {"frac_within": float(within.mean()), "max_abs": float(d.max()), "psnr": psnr,
            "exact": float((d == 0).mean()), "alpha_equal": bool(np.array_equal(img[..., 3], ref[..., 3]))}


def make_context(library: hip.Library, name: str, w: int = 64, h: int = 64, device: int = 0) -> hip.Context:
    ctx = hip.Context(device, library)
    ctx.upload_static(pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(golden_scene(name))
    return ctx


def render_frames(ctx: hip.Context, spp: int, flags: int = 0) -> np.ndarray:
    for it in range(1, spp + 1):
        ctx.render(it, flags=flags)
    return ctx.readback(hip.BUF_RAW)


def sort_by_xy(arr: np.ndarray) -> np.ndarray:
    return arr[np.argsort(arr["xy"], kind="stable")]


def assert_hits_identical(got: np.ndarray, ref: np.ndarray):
    """Bit-for-bit equality of hit records in every field that carries information.

    For a ray that ends as a miss (v < 0) the reference leaves `prim_index = tri_indices[<stale index>]` behind
    (the index indirection runs on misses too, CoreRef.cpp:2017-2022) -- a value that depends on the ORDER of the
    triangle array, which librayhip permutes for locality (bvh_layout.h).  Nothing reads it (ShadeSurface tests v first),
    so it is compared only for real hits; t, u, v and obj_index are compared for every ray.
    """
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), f
    assert np.array_equal(got["obj_index"], ref["obj_index"])
    hit = ref["v"] >= 0
    assert np.array_equal(got["prim_index"][hit], ref["prim_index"][hit])
