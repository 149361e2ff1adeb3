"""ctypes binding of include/rayhip.h -- the C-ABI of librayhip.so (hand-written HIP kernels for gfx950).

`Context` is a thin object over the rayhip_* entry points; it takes scenes as serialised blobs
(ray_amd/csrc/scene_blob.h) or as rayhip_scene_desc and returns numpy arrays.  There is no CPU path: creating a
Context without a GPU raises.  (`prefix`/`lib_path` exist so that tests can point the same wrapper at
tests/hostsim, the host-compiled copy of the kernel sources used to debug parity without a GPU.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RAYHIP_LIB = os.path.join(_HERE, "csrc", "_build", "librayhip.so")

BUF_FINAL, BUF_RAW, BUF_BASE_COLOR, BUF_DEPTH_NORMALS, BUF_VARIANCE = 0, 1, 2, 3, 4
REDUCE_RADIANCE, REDUCE_BASE_COLOR, REDUCE_DEPTH_NORMALS, REDUCE_VARIANCE, REDUCE_ALL = 1, 2, 4, 8, 15
COMM_ID_BYTES = 128
FLAG_SORT_RAYS = 1 << 0
FLAG_COUNT_TRAVERSAL = 1 << 1
FLAG_TIME_STAGES = 1 << 2
FLAG_COUNT_WIDE = 1 << 3


class PassSettings(C.Structure):
    _fields_ = [
        ("max_diff_depth", C.c_uint8), ("max_spec_depth", C.c_uint8), ("max_refr_depth", C.c_uint8),
        ("max_transp_depth", C.c_uint8), ("max_total_depth", C.c_uint8), ("min_total_depth", C.c_uint8),
        ("min_transp_depth", C.c_uint8), ("flags", C.c_uint8),
        ("clamp_direct", C.c_float), ("clamp_indirect", C.c_float), ("min_samples", C.c_int32),
        ("variance_threshold", C.c_float), ("regularize_alpha", C.c_float),
    ]


class Camera(C.Structure):  # rayhip_camera == Ray::camera_t
    _fields_ = [
        ("type", C.c_uint8), ("filter", C.c_uint8), ("view_transform", C.c_uint8), ("ltype", C.c_uint8),
        ("filter_width", C.c_float),
        ("fov", C.c_float), ("exposure", C.c_float), ("gamma", C.c_float), ("sensor_height", C.c_float),
        ("focus_distance", C.c_float), ("focal_length", C.c_float), ("fstop", C.c_float),
        ("lens_rotation", C.c_float), ("lens_ratio", C.c_float),
        ("lens_blades", C.c_int32),
        ("clip_start", C.c_float), ("clip_end", C.c_float),
        ("origin", C.c_float * 3), ("fwd", C.c_float * 3), ("side", C.c_float * 3), ("up", C.c_float * 3),
        ("shift", C.c_float * 2),
        ("mi_index", C.c_uint32), ("uv_index", C.c_uint32),
        ("pass_settings", PassSettings),
    ]


class Stats(C.Structure):
    _fields_ = [("t", C.c_ulonglong * 11)]
    NAMES = ("primary_ray_gen", "primary_trace", "primary_shade", "primary_shadow", "secondary_sort",
             "secondary_trace", "secondary_shade", "secondary_shadow", "denoise", "cache_update", "cache_resolve")

    def as_dict(self):
        return {n: int(self.t[i]) for i, n in enumerate(self.NAMES)}


class TravCounters(C.Structure):
    _fields_ = [("rays", C.c_ulonglong), ("nodes", C.c_ulonglong), ("tris", C.c_ulonglong), ("instances", C.c_ulonglong),
                ("max_stack", C.c_ulonglong), ("nodes4", C.c_ulonglong)]

    def as_dict(self):
        return {"rays": int(self.rays), "nodes": int(self.nodes), "tris": int(self.tris), "instances": int(self.instances),
                "max_stack": int(self.max_stack), "nodes4": int(self.nodes4)}


RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("pdf", "<f4"), ("c", "<f4", 3), ("ior", "<f4", 4),
                      ("cone_width", "<f4"), ("cone_spread", "<f4"), ("xy", "<u4"), ("depth", "<u4")])
SHADOW_RAY_DTYPE = np.dtype([("o", "<f4", 3), ("depth", "<u4"), ("d", "<f4", 3), ("dist", "<f4"), ("c", "<f4", 3),
                             ("xy", "<u4")])
HIT_DTYPE = np.dtype([("obj_index", "<i4"), ("prim_index", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
assert RAY_DTYPE.itemsize == 72 and SHADOW_RAY_DTYPE.itemsize == 48 and HIT_DTYPE.itemsize == 20

# every symbol include/rayhip.h declares (tests check that the built library exports all of them)
ENTRY_POINTS = (
    "last_error", "abi_version", "device_count", "ctx_create", "ctx_destroy", "ctx_device_name", "upload_static", "resize", "clear",
    "scene_upload", "bake_sky", "bake_sky_blob", "scene_bvh_width", "closest_hit_form", "scene_upload_blob", "scene_update_instances", "scene_update_instances_blob", "set_filter_table", "render", "render_batch", "max_batch", "reserve_batch", "set_tonemap_lut", "denoise_nlm", "readback", "readback_device", "set_raw_device",
    "sync", "set_shard", "get_trav_counters", "get_trav_timing", "get_stage_times", "k_generate_primary_rays", "k_intersect_closest",
    "k_intersect_shadow", "k_scrambled_rand", "k_shade",
    "comm_create", "comm_probe", "comm_info", "comm_unique_id", "comm_create_rank", "comm_bind", "comm_reduce_framebuffers", "comm_destroy",
    "unet_init", "denoise_unet", "unet_set_precision", "unet_read_tensor",
    "export_shard_device", "owned_bytes", "export_owned", "import_owned", "finish_import",
)


def _aligned_copy(buf: bytes, align: int = 64) -> np.ndarray:
    raw = np.empty(len(buf) + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + len(buf)]
    out[:] = np.frombuffer(buf, dtype=np.uint8)
    return out


class Library:
    def __init__(self, lib_path: str = RAYHIP_LIB, prefix: str = "rayhip_"):
        if not os.path.exists(lib_path):
            raise RuntimeError(f"{lib_path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        self.lib = C.CDLL(lib_path)
        self.prefix = prefix
        vp = C.c_void_p
        f = self.fn
        f("last_error").restype = C.c_char_p
        f("ctx_create").argtypes = [C.c_int, C.POINTER(vp)]
        f("ctx_destroy").argtypes = [vp]
        f("ctx_destroy").restype = None
        f("ctx_device_name").argtypes = [vp, C.c_char_p, C.c_int]
        f("upload_static").argtypes = [vp, vp, C.c_uint32]
        f("resize").argtypes = [vp, C.c_int, C.c_int]
        f("clear").argtypes = [vp, C.POINTER(C.c_float * 4)]
        f("scene_upload_blob").argtypes = [vp, vp, C.c_size_t, C.POINTER(Camera)]
        f("set_filter_table").argtypes = [vp, vp, C.c_int]
        if prefix == "rayhip_":
            f("bake_sky_blob").argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
        else:
            f("bake_sky").argtypes = [vp, C.c_int, C.c_int, vp]
            f("env_map_texels").argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_int * 2)]
        f("render").argtypes = [vp, C.POINTER(Camera), C.POINTER(C.c_int * 4), C.c_int, C.c_uint32, C.POINTER(Stats)]
        f("render_batch").argtypes = [vp, C.POINTER(Camera), C.POINTER(C.c_int * 4), C.c_int, C.c_int, C.c_uint32, C.POINTER(Stats)]
        f("max_batch").argtypes = [vp]
        f("scene_bvh_width").argtypes = [vp]
        if prefix == "rayhip_":
            f("closest_hit_form").argtypes = [vp]
        f("reserve_batch").argtypes = [vp, C.c_int]
        f("set_tonemap_lut").argtypes = [vp, C.c_int, vp, C.c_int]
        f("denoise_nlm").argtypes = [vp, C.POINTER(Camera), C.POINTER(C.c_int * 4), C.c_int]
        f("readback").argtypes = [vp, C.c_int, vp, C.c_int]
        f("sync").argtypes = [vp]
        f("set_shard").argtypes = [vp, C.c_int, C.c_int, C.c_int]
        f("get_trav_counters").argtypes = [vp, C.POINTER(TravCounters * 2), C.c_int]
        f("k_generate_primary_rays").argtypes = [vp, C.POINTER(Camera), C.POINTER(C.c_int * 4), C.c_int, vp, vp, C.POINTER(C.c_int)]
        f("k_intersect_closest").argtypes = [vp, C.POINTER(Camera), vp, vp, C.c_int, C.c_int, C.c_uint32, C.POINTER(TravCounters)]
        f("k_intersect_shadow").argtypes = [vp, C.POINTER(Camera), vp, C.c_int, C.c_int, vp, C.POINTER(TravCounters)]
        f("k_scrambled_rand").argtypes = [vp, vp, vp, vp, C.c_int, vp]
        f("k_shade").argtypes = [vp, C.POINTER(Camera), C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.POINTER(C.c_int), vp,
                                 C.POINTER(C.c_int)]
        f("scene_update_instances_blob").argtypes = [vp, vp, C.c_size_t, C.POINTER(Camera)]
        f("owned_bytes").argtypes = [vp, C.c_uint32, C.c_int, C.c_int]
        f("owned_bytes").restype = C.c_size_t
        f("export_owned").argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        f("import_owned").argtypes = [vp, C.c_uint32, C.c_int, vp, C.c_size_t]
        f("finish_import").argtypes = [vp, C.POINTER(Camera)]
        if prefix == "rayhip_":
            f("unet_init").argtypes = [vp, vp, C.c_int, vp, C.c_int]
            f("denoise_unet").argtypes = [vp, C.POINTER(Camera), C.POINTER(C.c_int * 4), C.c_int]
            f("unet_read_tensor").argtypes = [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_int * 3)]
            f("unet_set_precision").argtypes = [vp, C.c_int]
            f("readback_device").argtypes = [vp, C.c_int, vp, C.c_int]
            f("set_raw_device").argtypes = [vp, vp, C.c_int, C.POINTER(Camera)]
            f("export_shard_device").argtypes = [vp, C.c_int, vp]
            f("comm_create").argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
            f("comm_unique_id").argtypes = [vp, C.c_size_t]
            f("comm_probe").argtypes = []
            f("comm_info").argtypes = [vp, C.POINTER(C.c_int * 4)]
            f("comm_create_rank").argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
            f("comm_bind").argtypes = [vp, C.c_int, vp]
            f("comm_reduce_framebuffers").argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(Camera)]
            f("comm_destroy").argtypes = [vp]
            f("comm_destroy").restype = None
            f("get_trav_timing").argtypes = [vp, C.POINTER(C.c_double * 2), C.POINTER(C.c_ulonglong * 2), C.c_int]
            f("get_stage_times").argtypes = [vp, C.POINTER(Stats), C.c_int]

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def device_count(self) -> int:
        return int(self.fn("device_count")())

    def abi_version(self) -> int:
        """RAYHIP_ABI_VERSION of the header the library was built from (include/rayhip.h)"""
        return int(self.fn("abi_version")())

    def check(self, status: int):
        if status != 0:
            raise RuntimeError(f"{self.prefix}*: " + self.fn("last_error")().decode())


class Context:
    """One rayhip context = one GPU (rayhip.h).  Mirrors what RendererHIP holds on the C++ side."""

    def __init__(self, device: int = 0, library: Library = None):
        self.L = library or Library()
        self._ctx = C.c_void_p()
        self.L.check(self.L.fn("ctx_create")(device, C.byref(self._ctx)))
        self.w = self.h = 0
        self.cam = None
        self._blob = None

    def close(self):
        if self._ctx:
            self.L.fn("ctx_destroy")(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self.L.check(self.L.fn("ctx_device_name")(self._ctx, buf, 256))
        return buf.value.decode()

    def upload_static(self, pmj: np.ndarray):
        pmj = np.ascontiguousarray(pmj, dtype=np.uint32)
        self.L.check(self.L.fn("upload_static")(self._ctx, pmj.ctypes.data, pmj.size))

    def resize(self, w: int, h: int):
        self.L.check(self.L.fn("resize")(self._ctx, w, h))
        self.w, self.h = w, h

    def clear(self, rgba=(0.0, 0.0, 0.0, 0.0)):
        v = (C.c_float * 4)(*rgba)
        self.L.check(self.L.fn("clear")(self._ctx, C.byref(v)))

    def upload_scene_blob(self, blob: bytes) -> Camera:
        self._blob = _aligned_copy(blob)
        cam = Camera()
        self.L.check(self.L.fn("scene_upload_blob")(self._ctx, self._blob.ctypes.data, self._blob.size, C.byref(cam)))
        self.cam = cam
        return cam

    def bake_sky(self, w: int, h: int, blob: bytes = None) -> np.ndarray:
        """the sky environment map ([h, w, 4] uint8: shared-exponent RGBE) of a physical-sky scene: on the device from `blob` (default: the blob
        this context uploaded last) -- rayhip_bake_sky_blob; with the host build of the kernels, from the uploaded scene"""
        out = np.zeros((h, w), dtype=np.uint32)
        if self.L.prefix == "rayhip_":
            b = self._blob if blob is None else _aligned_copy(blob)
            self.L.check(self.L.fn("bake_sky_blob")(self._ctx, b.ctypes.data, b.size, w, h, out.ctypes.data))
        else:
            self.L.check(self.L.fn("bake_sky")(self._ctx, w, h, out.ctypes.data))
        return out.view(np.uint8).reshape(h, w, 4)

    def env_map_texels(self) -> np.ndarray:
        """(host build only) the texels of the environment map the uploaded scene came with, [h, w, 4] uint8"""
        buf = np.zeros(4096 * 2048, dtype=np.uint32)
        wh = (C.c_int * 2)()
        self.L.check(self.L.fn("env_map_texels")(self._ctx, buf.ctypes.data, buf.size, C.byref(wh)))
        return buf[:wh[0] * wh[1]].view(np.uint8).reshape(wh[1], wh[0], 4).copy()

    def update_instances(self, blob: bytes) -> int:
        """instances / lights / environment of `blob` over the geometry that is on the device; the top level is rebuilt
        there (rayhip_scene_update_instances).  Returns 0, or 2 if the change needs a full upload_scene()."""
        self._blob2 = _aligned_copy(blob)
        cam = Camera()
        rc = self.L.fn("scene_update_instances_blob")(self._ctx, self._blob2.ctypes.data, self._blob2.size, C.byref(cam))
        if rc == 2:
            return 2
        self.L.check(rc)
        self.cam = cam
        return 0

    def render(self, iteration: int, rect=None, cam: Camera = None, flags: int = 0, stats: Stats = None):
        rect = (0, 0, self.w, self.h) if rect is None else rect
        r = (C.c_int * 4)(*rect)
        cam = cam or self.cam
        self.L.check(self.L.fn("render")(self._ctx, C.byref(cam), C.byref(r), iteration, flags,
                                         C.byref(stats) if stats is not None else None))

    def render_batch(self, first_iteration: int, count: int, rect=None, cam: Camera = None, flags: int = 0, stats: Stats = None):
        """iterations first_iteration .. first_iteration + count - 1 in as few wavefront passes as possible (same pixels as
        `count` render() calls, bit for bit)"""
        rect = (0, 0, self.w, self.h) if rect is None else rect
        r = (C.c_int * 4)(*rect)
        cam = cam or self.cam
        self.L.check(self.L.fn("render_batch")(self._ctx, C.byref(cam), C.byref(r), first_iteration, count, flags,
                                               C.byref(stats) if stats is not None else None))

    def bvh_width(self) -> int:
        """8 / 4: the wide quantised BLAS form the kernels walk; 2: the reference's BVH2"""
        return int(self.L.fn("scene_bvh_width")(self._ctx))

    def closest_hit_form(self) -> int:
        """0: one ray per lane; 1: persistent refill kernel; 2: the pooled kernel (include/rayhip.h)"""
        return int(self.L.fn("closest_hit_form")(self._ctx)) if self.L.prefix == "rayhip_" else 0

    def max_batch(self) -> int:
        """largest number of iterations one wavefront pass of the current frame can carry"""
        return int(self.L.fn("max_batch")(self._ctx))

    def set_tonemap_lut(self, view_transform: int, lut: np.ndarray):
        """dims^3 RGB10_A2 table of a non-Standard view transform (uint32, x fastest)"""
        lut = np.ascontiguousarray(lut, dtype=np.uint32).reshape(-1)
        dims = round(lut.size ** (1.0 / 3.0))
        assert dims ** 3 == lut.size, "the table must be a cube"
        self.L.check(self.L.fn("set_tonemap_lut")(self._ctx, view_transform, lut.ctypes.data, dims))

    def denoise_nlm(self, iteration: int, rect=None, cam: Camera = None):
        """RendererBase::DenoiseImage(region): NLM filter of what `iteration` iterations accumulated -> RAW, FINAL"""
        rect = (0, 0, self.w, self.h) if rect is None else rect
        r = (C.c_int * 4)(*rect)
        cam = cam or self.cam
        self.L.check(self.L.fn("denoise_nlm")(self._ctx, C.byref(cam), C.byref(r), iteration))

    def reserve_batch(self, count: int):
        """allocate the buffers passes of `count` iterations need (under the current shard) ahead of time"""
        self.L.check(self.L.fn("reserve_batch")(self._ctx, count))

    def readback(self, which: int = BUF_RAW, out: np.ndarray = None) -> np.ndarray:
        """`out`: a [h, w, 4] float32 array to fill (a caller that reads frames repeatedly should reuse one: the HIP runtime
        pins the destination pages for the copy, and unmapping pinned pages -- numpy freeing a large array -- makes the kernel
        driver stop and restart the process's GPU queues, 15-30 ms during which running kernels stand still)"""
        if out is None:
            out = np.empty((self.h, self.w, 4), dtype=np.float32)
        assert out.shape == (self.h, self.w, 4) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        self.L.check(self.L.fn("readback")(self._ctx, which, out.ctypes.data, self.w))
        return out

    def readback_device(self, which: int, device_ptr: int, pitch_px: int = None):
        self.L.check(self.L.fn("readback_device")(self._ctx, which, C.c_void_p(device_ptr), pitch_px or self.w))

    def set_raw_device(self, device_ptr: int, pitch_px: int = None, cam: Camera = None):
        cam = cam or self.cam
        self.L.check(self.L.fn("set_raw_device")(self._ctx, C.c_void_p(device_ptr), pitch_px or self.w, C.byref(cam)))

    def export_shard_device(self, which: int, device_ptr: int):
        """this rank's OWNED pixels of image `which` (zero elsewhere) into device memory: the operand of the frame reduce"""
        self.L.check(self.L.fn("export_shard_device")(self._ctx, which, C.c_void_p(device_ptr)))

    # the exchange with the caller's transport (rayhip.h: rayhip_export_owned & co.): the tiles a rank owns, densely packed
    def owned_bytes(self, what: int, nranks: int, rank: int) -> int:
        return int(self.L.fn("owned_bytes")(self._ctx, what, nranks, rank))

    def export_owned(self, what: int, device_ptr: int, capacity_bytes: int):
        self.L.check(self.L.fn("export_owned")(self._ctx, what, C.c_void_p(device_ptr), capacity_bytes))

    def import_owned(self, what: int, from_rank: int, device_ptr: int, nbytes: int):
        self.L.check(self.L.fn("import_owned")(self._ctx, what, from_rank, C.c_void_p(device_ptr), nbytes))

    def finish_import(self, cam: Camera = None):
        self.L.check(self.L.fn("finish_import")(self._ctx, C.byref(cam or self.cam)))

    # UNet denoiser (rayhip.h: rayhip_unet_init / rayhip_denoise_unet)
    def unet_init(self, weights: np.ndarray, offsets: np.ndarray, alignment: int = 8):
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        assert offsets.size == 32
        self.L.check(self.L.fn("unet_init")(self._ctx, weights.ctypes.data, weights.size, offsets.ctypes.data, alignment))

    def unet_precision(self, half: bool):
        """False: the exact f32 form (default); True: f16 tensors / weights with f32 accumulators (what the reference's GPU backends run)"""
        self.L.check(self.L.fn("unet_set_precision")(self._ctx, int(bool(half))))

    def denoise_unet(self, pass_index: int = -1, rect=None, cam: Camera = None):
        r = (C.c_int * 4)(*((0, 0, self.w, self.h) if rect is None else rect))
        self.L.check(self.L.fn("denoise_unet")(self._ctx, C.byref(cam or self.cam), C.byref(r), pass_index))

    def unet_read_tensor(self, which: int) -> np.ndarray:
        wr, hr = 16 * ((self.w + 15) // 16), 16 * ((self.h + 15) // 16)
        buf = np.zeros((wr + 2) * (hr + 2) * 112, dtype=np.float32)
        dims = (C.c_int * 3)()
        self.L.check(self.L.fn("unet_read_tensor")(self._ctx, which, buf.ctypes.data, buf.size, C.byref(dims)))
        n = dims[0] * dims[1] * dims[2]
        return buf[:n].reshape(dims[0], dims[1], dims[2]).copy()

    def set_shard(self, tile: int, shard_count: int, shard_index: int):
        """multi-GPU tile sharding: render only the tiles whose ordinal % shard_count == shard_index"""
        self.L.check(self.L.fn("set_shard")(self._ctx, tile, shard_count, shard_index))

    def sync(self):
        self.L.check(self.L.fn("sync")(self._ctx))

    def trav_counters(self, reset=True):
        out = (TravCounters * 2)()
        self.L.check(self.L.fn("get_trav_counters")(self._ctx, C.byref(out), int(reset)))
        return out[0].as_dict(), out[1].as_dict()

    def trav_timing(self, reset=True):
        ms = (C.c_double * 2)()
        n = (C.c_ulonglong * 2)()
        self.L.check(self.L.fn("get_trav_timing")(self._ctx, C.byref(ms), C.byref(n), int(reset)))
        return (float(ms[0]), int(n[0])), (float(ms[1]), int(n[1]))

    def stage_times(self, reset=True) -> dict:
        st = Stats()
        self.L.check(self.L.fn("get_stage_times")(self._ctx, C.byref(st), int(reset)))
        return st.as_dict()

    # ---- kernel-level hooks ------------------------------------------------------------------------------
    def k_generate_primary_rays(self, iteration: int, rect=None, cam: Camera = None):
        rect = (0, 0, self.w, self.h) if rect is None else rect
        n = rect[2] * rect[3]
        rays = np.zeros(n, dtype=RAY_DTYPE)
        hits = np.zeros(n, dtype=HIT_DTYPE)
        cnt = C.c_int(0)
        r = (C.c_int * 4)(*rect)
        self.L.check(self.L.fn("k_generate_primary_rays")(self._ctx, C.byref(cam or self.cam), C.byref(r), iteration,
                                                          rays.ctypes.data, hits.ctypes.data, C.byref(cnt)))
        return rays[:cnt.value], hits[:cnt.value]

    def k_intersect_closest(self, rays: np.ndarray, hits: np.ndarray, iteration: int, cam: Camera = None,
                            flags: int = FLAG_COUNT_TRAVERSAL):
        """flags=FLAG_COUNT_TRAVERSAL: instrumented walk of the reference BVH2 (+ visit counters); 0: the product kernel"""
        rays = np.ascontiguousarray(rays.copy())
        hits = np.ascontiguousarray(hits.copy())
        tc = TravCounters()
        self.L.check(self.L.fn("k_intersect_closest")(self._ctx, C.byref(cam or self.cam), rays.ctypes.data,
                                                      hits.ctypes.data, len(rays), iteration, flags, C.byref(tc)))
        return rays, hits, tc.as_dict()

    def k_intersect_shadow(self, rays: np.ndarray, iteration: int, cam: Camera = None):
        rays = np.ascontiguousarray(rays)
        out = np.zeros((len(rays), 4), dtype=np.float32)
        tc = TravCounters()
        self.L.check(self.L.fn("k_intersect_shadow")(self._ctx, C.byref(cam or self.cam), rays.ctypes.data, len(rays),
                                                     iteration, out.ctypes.data, C.byref(tc)))
        return out, tc.as_dict()

    def k_shade(self, bounce: int, iteration: int, rays: np.ndarray, hits: np.ndarray, color: np.ndarray, cam: Camera = None):
        """ShadePrimary (bounce 0) / ShadeSecondary on host (ray, hit) pairs; returns (color, secondary rays, shadow rays)
        -- the twin of the oracle's refk_shade.  Emitted rays come back in no particular order."""
        rays, hits = np.ascontiguousarray(rays), np.ascontiguousarray(hits)
        color = np.ascontiguousarray(color.copy(), dtype=np.float32)
        assert color.shape == (self.h, self.w, 4)
        n = len(rays)
        sec = np.zeros(n + 1, dtype=RAY_DTYPE)
        sh = np.zeros(n + 1, dtype=SHADOW_RAY_DTYPE)
        nsec, nsh = C.c_int(), C.c_int()
        self.L.check(self.L.fn("k_shade")(self._ctx, C.byref(cam or self.cam), bounce, iteration, rays.ctypes.data, hits.ctypes.data, n,
                                          color.ctypes.data, sec.ctypes.data, C.byref(nsec), sh.ctypes.data, C.byref(nsh)))
        return color, sec[:nsec.value], sh[:nsh.value]

    def k_scrambled_rand(self, dims, seeds, samples):
        dims = np.ascontiguousarray(dims, dtype=np.uint32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        out = np.zeros((len(dims), 2), dtype=np.float32)
        self.L.check(self.L.fn("k_scrambled_rand")(self._ctx, dims.ctypes.data, seeds.ctypes.data, samples.ctypes.data,
                                                   len(dims), out.ctypes.data))
        return out


class Comm:
    """rayhip_comm: the frame reduce over RCCL behind the C ABI.  `Comm(lib, devices)` drives all GPUs from this process
    (what a C++ host does); `Comm.for_rank(lib, id, nranks, rank, ctx)` is the one-process-per-GPU form."""

    def __init__(self, library: Library, devices=None, _handle=None):
        self.L = library
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
            return
        devs = (C.c_int * len(devices))(*devices)
        self.L.check(self.L.fn("comm_create")(len(devices), devs, C.byref(self._h)))

    @staticmethod
    def probe(library: Library) -> str:
        """'' if this process can load RCCL with every symbol the exchange needs, else the reason -- local, not collective"""
        if library.fn("comm_probe")() == 0:
            return ""
        return (library.fn("last_error")() or b"RCCL unavailable").decode()

    def info(self) -> dict:
        out = (C.c_int * 4)()
        self.L.check(self.L.fn("comm_info")(self._h, C.byref(out)))
        return {"nranks": out[0], "rank": out[1], "nccl_comm_count": out[2], "in_process": bool(out[3])}

    @staticmethod
    def unique_id(library: Library) -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        library.check(library.fn("comm_unique_id")(buf, COMM_ID_BYTES))
        return buf.raw

    @classmethod
    def for_rank(cls, library: Library, unique_id: bytes, nranks: int, rank: int, ctx: "Context"):
        h = C.c_void_p()
        library.check(library.fn("comm_create_rank")(C.c_char_p(unique_id), nranks, rank, ctx._ctx, C.byref(h)))
        return cls(library, _handle=h)

    def bind(self, rank: int, ctx: "Context"):
        self.L.check(self.L.fn("comm_bind")(self._h, rank, ctx._ctx))

    def reduce_framebuffers(self, root: int, cam: Camera, what: int = REDUCE_ALL):
        self.L.check(self.L.fn("comm_reduce_framebuffers")(self._h, root, what, C.byref(cam)))

    def close(self):
        if self._h:
            self.L.fn("comm_destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
