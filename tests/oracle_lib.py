"""Test-side access to the oracle: the REAL reference built by oracle/Makefile (oracle/_ref/libray_ref.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

from ray_amd import api, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libray_ref.so")
HOSTSIM_LIB = os.path.join(ROOT, "tests", "hostsim", "_build", "libhostsim.so")


def have_ref() -> bool:
    return os.path.exists(REF_LIB)


def have_hostsim() -> bool:
    return os.path.exists(HOSTSIM_LIB)


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        lib = api.load_capi_library(REF_LIB)
        vp = C.c_void_p
        lib.refk_pmj_table.argtypes = [C.POINTER(vp), C.POINTER(C.c_uint32)]
        lib.refk_pmj_table.restype = None
        lib.refk_filter_table.argtypes = [C.c_uint32, C.c_float, vp]
        lib.refk_filter_table.restype = None
        lib.refk_export_scene.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        lib.refk_free.argtypes = [vp]
        lib.refk_free.restype = None
        lib.refk_generate_primary_rays.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int * 4), C.c_int, vp, vp, C.POINTER(C.c_int)]
        lib.refk_intersect_closest.argtypes = [vp, vp, vp, C.c_int, C.c_int]
        lib.refk_intersect_shadow.argtypes = [vp, vp, C.c_int, C.c_int, vp]
        lib.refk_scrambled_rand.argtypes = [vp, vp, vp, C.c_int, vp]
        lib.refk_scrambled_rand.restype = None
        lib.refk_shade.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.POINTER(C.c_int), vp,
                                   C.POINTER(C.c_int)]
        lib.refk_unet_weights.argtypes = [vp, C.c_int, vp]
        lib.refk_unet_passes.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_int * 3)]
        lib.refk_unet_passes.restype = C.c_size_t
        _ref = lib
    return _ref


def ref_unet_weights():
    """(weights, offsets[32]) as the reference's SetupUNetWeights<float>(8) lays them out"""
    lib = ref_lib()
    n = lib.refk_unet_weights(None, 0, None)
    w = np.zeros(n, dtype=np.float32)
    off = np.zeros(32, dtype=np.int32)
    assert lib.refk_unet_weights(w.ctypes.data, n, off.ctypes.data) == n
    return w, off


def ref_unet_passes(full, base, dn, last_pass):
    """the tensor (with border) pass `last_pass` of the reference's UNet writes, given the three input images [h, w, 4]"""
    h, w = full.shape[:2]
    full, base, dn = (np.ascontiguousarray(a, dtype=np.float32) for a in (full, base, dn))
    wr, hr = 16 * ((w + 15) // 16), 16 * ((h + 15) // 16)
    buf = np.zeros((wr + 2) * (hr + 2) * 112, dtype=np.float32)
    dims = (C.c_int * 3)()
    n = ref_lib().refk_unet_passes(w, h, full.ctypes.data, base.ctypes.data, dn.ctypes.data, last_pass, buf.ctypes.data, buf.size, C.byref(dims))
    assert n > 0
    return buf[:n].reshape(dims[0], dims[1], dims[2]).copy()


def create_renderer(w, h, renderer_type="REF", verbose=False, use_tex_compression=False):
    return api.create_renderer_from(ref_lib(), api.Settings(w, h, use_tex_compression=use_tex_compression, verbose=verbose),
                                    renderer_type)


def pmj_table() -> np.ndarray:
    p, n = C.c_void_p(), C.c_uint32()
    ref_lib().refk_pmj_table(C.byref(p), C.byref(n))
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()


def filter_table(pixel_filter=2, width=1.5) -> np.ndarray:
    out = np.zeros(1024, dtype=np.float32)
    ref_lib().refk_filter_table(pixel_filter, width, out.ctypes.data)
    return out


def export_scene(scene) -> bytes:
    p, n = C.c_void_p(), C.c_size_t()
    if ref_lib().refk_export_scene(scene._ptr, C.byref(p), C.byref(n)) != 0:
        raise RuntimeError("refk_export_scene failed")
    try:
        return C.string_at(p, n.value)
    finally:
        ref_lib().refk_free(p)


def ref_generate_primary_rays(scene, w, h, iteration, rect=None):
    rect = rect or (0, 0, w, h)
    n = rect[2] * rect[3]
    rays = np.zeros(n, dtype=hip.RAY_DTYPE)
    hits = np.zeros(n, dtype=hip.HIT_DTYPE)
    cnt = C.c_int()
    r = (C.c_int * 4)(*rect)
    ref_lib().refk_generate_primary_rays(scene._ptr, w, h, C.byref(r), iteration, rays.ctypes.data, hits.ctypes.data, C.byref(cnt))
    return rays[:cnt.value], hits[:cnt.value]


def ref_intersect_closest(scene, rays, hits, iteration):
    rays, hits = np.ascontiguousarray(rays.copy()), np.ascontiguousarray(hits.copy())
    ref_lib().refk_intersect_closest(scene._ptr, rays.ctypes.data, hits.ctypes.data, len(rays), iteration)
    return rays, hits


def ref_intersect_shadow(scene, rays, iteration):
    rays = np.ascontiguousarray(rays)
    out = np.zeros((len(rays), 4), dtype=np.float32)
    ref_lib().refk_intersect_shadow(scene._ptr, rays.ctypes.data, len(rays), iteration, out.ctypes.data)
    return out


def ref_scrambled_rand(dims, seeds, samples):
    dims = np.ascontiguousarray(dims, dtype=np.uint32)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    out = np.zeros((len(dims), 2), dtype=np.float32)
    ref_lib().refk_scrambled_rand(dims.ctypes.data, seeds.ctypes.data, samples.ctypes.data, len(dims), out.ctypes.data)
    return out


def ref_shade(scene, w, h, bounce, iteration, rays, hits, color):
    """returns (color, secondary_rays, shadow_rays)"""
    n = len(rays)
    rays, hits = np.ascontiguousarray(rays), np.ascontiguousarray(hits)
    color = np.ascontiguousarray(color.copy(), dtype=np.float32)
    sec = np.zeros(n + 1, dtype=hip.RAY_DTYPE)
    sh = np.zeros(n + 1, dtype=hip.SHADOW_RAY_DTYPE)
    nsec, nsh = C.c_int(), C.c_int()
    ref_lib().refk_shade(scene._ptr, w, h, bounce, iteration, rays.ctypes.data, hits.ctypes.data, n, color.ctypes.data,
                         sec.ctypes.data, C.byref(nsec), sh.ctypes.data, C.byref(nsh))
    return color, sec[:nsec.value], sh[:nsh.value]


def hostsim_context(w, h, blob: bytes, pmj=None):
    """hostsim = the kernel sources compiled for the host (tests/hostsim); same wrapper class as the GPU"""
    L = hip.Library(HOSTSIM_LIB, prefix="hostsim_")
    ctx = hip.Context(0, L)
    ctx.upload_static(pmj_table() if pmj is None else pmj)
    ctx.resize(w, h)
    ctx.upload_scene_blob(blob)
    return ctx


def render_ref(scene_fn, w, h, spp, renderer_type="REF", use_tex_compression=False, **cam):
    r = create_renderer(w, h, renderer_type, use_tex_compression=use_tex_compression)
    s = r.CreateScene()
    scene_fn(s, **cam)
    region = api.RegionContext((0, 0, w, h))
    for _ in range(spp):
        r.RenderScene(s, region)
    return r, s
