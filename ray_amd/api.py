"""Host-side mirror of sergcpp/Ray's public interface (Ray.h / RendererBase.h / SceneBase.h), over ray_capi.

Names, argument meaning and error behaviour follow the reference so that code written against
`Ray::CreateRenderer -> RendererBase::CreateScene -> SceneBase::Add* -> Finalize -> RenderScene` reads the same
here:

    r = ray_amd.CreateRenderer(Settings(w, h), renderer_type="HIP")
    s = r.CreateScene()
    mat = s.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(.5, .5, .5)))
    ...
    s.Finalize()
    region = RegionContext((0, 0, w, h))
    for _ in range(spp): r.RenderScene(s, region)
    img = r.get_pixels_ref()

`CreateRenderer(..., renderer_type="HIP")` is the product: the reference's host-side scene code with the
RendererHIP/SceneHIP backend (ray_amd/host) driving librayhip.so.  It raises RuntimeError when no gfx950 device
is present -- exactly like the reference's GPU factories throw (Ray.cpp:58-63) -- and never falls back to a CPU
path.  The CPU renderer types exist only in the oracle library (tests load it through `tests/oracle_lib.py`).
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from enum import IntEnum
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_HOST_LIB = os.path.join(_HERE, "host", "_build", "libray_hip.so")


class eShadingNode(IntEnum):  # SceneBase.h:46
    Diffuse = 0
    Glossy = 1
    Refractive = 2
    Emissive = 3
    Mix = 4
    Transparent = 5
    Principled = 6


class eTextureFormat(IntEnum):  # SceneBase.h:151
    Undefined = 0
    RGBA8888 = 1
    RGB888 = 2
    RG88 = 3
    R8 = 4
    BC1 = 5  # block-compressed inputs: `data` holds the 4x4 blocks of `mips_count` levels, width / height are given
    BC3 = 6
    BC4 = 7
    BC5 = 8


class ePixelFilter(IntEnum):  # Types.h:60
    Box = 0
    Gaussian = 1
    BlackmanHarris = 2


class eAUXBuffer(IntEnum):  # Types.h:47
    SHL1 = 0
    BaseColor = 1
    DepthNormals = 2


InvalidHandle = _capi.INVALID_HANDLE
PhysicalSkyTexture = _capi.PHYSICAL_SKY_TEXTURE  # as env_map / back_map: the analytic sky, lit by the scene's directional lights


@dataclass
class Settings:  # Ray::settings_t, RendererBase.h:52-63
    w: int = 0
    h: int = 0
    use_tex_compression: bool = False  # (the reference's default is True; the HIP backend decodes such textures at export)
    verbose: bool = False


class RegionContext:
    """Ray::RegionContext (RendererBase.h:78-92): a rectangle + the number of iterations rendered on it."""

    def __init__(self, rect: Tuple[int, int, int, int], _lib=None):
        self._rect = tuple(int(v) for v in rect)
        self._lib = _lib
        self._ptr = None

    def _bind(self, lib):
        if self._ptr is None:
            self._lib = lib
            self._ptr = lib.ray_region_create(*self._rect)
        return self._ptr

    def rect(self):
        return self._rect

    @property
    def iteration(self) -> int:
        return 0 if self._ptr is None else int(self._lib.ray_region_iteration(self._ptr))

    @iteration.setter
    def iteration(self, v: int):
        if self._ptr is not None:
            self._lib.ray_region_set_iteration(self._ptr, int(v))

    def Clear(self):
        self.iteration = 0

    def __del__(self):
        if self._ptr is not None and self._lib is not None:
            self._lib.ray_region_destroy(self._ptr)
            self._ptr = None


def _set(struct, **kw):
    for k, v in kw.items():
        if v is None:
            continue
        cur = getattr(struct, k)
        if isinstance(cur, C.Array):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(struct, k, v)


class SceneBase:
    """Mirror of Ray::SceneBase (SceneBase.h:371-516)."""

    def __init__(self, lib, ptr):
        self._lib = lib
        self._ptr = ptr
        self._keep = []  # host arrays that must outlive AddMesh/AddTexture calls (the C++ side copies them)

    def __del__(self):
        if getattr(self, "_ptr", None):
            self._lib.ray_scene_destroy(self._ptr)
            self._ptr = None

    # -- environment ---------------------------------------------------------------------------------------
    def SetEnvironment(self, env_col=(0.0, 0.0, 0.0), back_col=(0.0, 0.0, 0.0), env_map=InvalidHandle,
                       back_map=InvalidHandle, env_map_rotation=0.0, back_map_rotation=0.0, importance_sample=True, **atmosphere):
        """atmosphere: envmap_resolution, clouds_density, cirrus_clouds_amount, stars_brightness, moon_radius, clouds_offset_x / _z
        (environment_desc_t::envmap_resolution and atmosphere_params_t, SceneBase.h:314-353); the rest stays at its default"""
        d = _capi.EnvDesc()
        self._lib.ray_default_env(C.byref(d))
        _set(d, env_col=env_col, back_col=back_col, env_map=env_map, back_map=back_map,
             env_map_rotation=env_map_rotation, back_map_rotation=back_map_rotation,
             importance_sample=int(importance_sample), **atmosphere)
        self._lib.ray_scene_set_environment(self._ptr, C.byref(d))

    # -- textures / materials ----------------------------------------------------------------------------------
    def AddTexture(self, data: np.ndarray, fmt=eTextureFormat.RGBA8888, is_srgb=True, is_normalmap=False,
                   generate_mipmaps=False, reconstruct_z=False, force_no_compression=True, size=None, mips_count=1,
                   is_YCoCg=False, convention_dx=False) -> int:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        h, w = (size[1], size[0]) if size is not None else (data.shape[0], data.shape[1])  # size = (w, h): block-compressed data
        d = _capi.TexDesc()
        d.mips_count, d.is_YCoCg = int(mips_count), int(is_YCoCg)
        d.convention = 1 if convention_dx else 0  # eTextureConvention::DX: y of a normal map inverted, block textures stored top-down
        d.format = int(fmt)
        d.data = data.ctypes.data_as(C.POINTER(C.c_uint8))
        d.data_size = data.size
        d.w, d.h = w, h
        d.is_srgb, d.is_normalmap = int(is_srgb), int(is_normalmap)
        d.force_no_compression = int(force_no_compression)  # False: settings_t::use_tex_compression decides
        d.generate_mipmaps, d.reconstruct_z = int(generate_mipmaps), int(reconstruct_z)
        return int(self._lib.ray_scene_add_texture(self._ptr, C.byref(d)))

    def AddMaterial(self, desc) -> int:
        """desc: ShadingNode or PrincipledMat (the two AddMaterial overloads, SceneBase.h:399-405)."""
        if isinstance(desc, ShadingNode):
            d = _capi.ShadingNodeDesc()
            self._lib.ray_default_shading_node(C.byref(d))
            _set(d, **{k: (int(v) if isinstance(v, (bool, IntEnum)) else v) for k, v in desc.__dict__.items()})
            return int(self._lib.ray_scene_add_material_node(self._ptr, C.byref(d)))
        if isinstance(desc, PrincipledMat):
            d = _capi.PrincipledMatDesc()
            self._lib.ray_default_principled(C.byref(d))
            _set(d, **{k: (int(v) if isinstance(v, bool) else v) for k, v in desc.__dict__.items()})
            return int(self._lib.ray_scene_add_material_principled(self._ptr, C.byref(d)))
        raise TypeError("AddMaterial takes a ShadingNode or a PrincipledMat")

    # -- geometry ------------------------------------------------------------------------------------------
    def AddMesh(self, attrs: np.ndarray, indices: np.ndarray, groups: Sequence[Tuple], stride=8, pos_offset=0,
                nrm_offset=3, uv_offset=6, bnm_offset=-1, allow_spatial_splits=False, use_fast_bvh_build=False) -> int:
        """attrs: interleaved float32 vertex attributes (default layout = position3, normal3, uv2 as in
        samples/00_basic); groups: (front_mat, back_mat|None, vtx_start, vtx_count) like mat_group_desc_t."""
        attrs = np.ascontiguousarray(attrs, dtype=np.float32).ravel()
        indices = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        g = (_capi.MatGroupDesc * len(groups))()
        for i, (front, back, start, count) in enumerate(groups):
            g[i].front_mat = front
            g[i].back_mat = front if back is None else back
            g[i].vtx_start, g[i].vtx_count = start, count
        d = _capi.MeshDesc()
        d.attrs = attrs.ctypes.data_as(C.POINTER(C.c_float))
        d.attrs_count = attrs.size
        d.stride, d.pos_offset, d.nrm_offset, d.uv_offset, d.bnm_offset = stride, pos_offset, nrm_offset, uv_offset, bnm_offset
        d.indices = indices.ctypes.data_as(C.POINTER(C.c_uint32))
        d.indices_count = indices.size
        d.base_vertex = 0
        d.groups = g
        d.groups_count = len(groups)
        d.allow_spatial_splits, d.use_fast_bvh_build = int(allow_spatial_splits), int(use_fast_bvh_build)
        return int(self._lib.ray_scene_add_mesh(self._ptr, C.byref(d)))

    def AddMeshInstance(self, mesh: int, xform=None, camera=True, diffuse=True, specular=True, refraction=True, shadow=True) -> int:
        """mesh_instance_desc_t (SceneBase.h:135-143); xform: 4x4, column-major as the reference takes it"""
        m = (C.c_float * 16)(*(np.eye(4, dtype=np.float32).ravel() if xform is None else np.asarray(xform, np.float32).ravel()))
        vis = (1 if camera else 0) | (2 if diffuse else 0) | (4 if specular else 0) | (8 if refraction else 0) | (16 if shadow else 0)
        if vis == 31:
            return int(self._lib.ray_scene_add_mesh_instance(self._ptr, mesh, C.byref(m)))
        return int(self._lib.ray_scene_add_mesh_instance_vis(self._ptr, mesh, C.byref(m), vis))

    def SetMeshInstanceTransform(self, mi: int, xform):
        m = (C.c_float * 16)(*np.asarray(xform, np.float32).ravel())
        self._lib.ray_scene_set_mesh_instance_transform(self._ptr, mi, C.byref(m))

    def RemoveMesh(self, mesh: int):
        self._lib.ray_scene_remove_mesh(self._ptr, mesh)

    def RemoveMeshInstance(self, mi: int):
        self._lib.ray_scene_remove_mesh_instance(self._ptr, mi)

    def RemoveLight(self, light: int):
        self._lib.ray_scene_remove_light(self._ptr, light)

    def AddLight(self, kind: str, **kw) -> int:
        """kind: 'directional' | 'sphere' | 'spot' | 'rect' | 'disk' | 'line' (the six AddLight overloads)."""
        kinds = {"directional": 0, "sphere": 1, "spot": 2, "rect": 3, "disk": 4, "line": 5}
        d = _capi.LightDesc()
        self._lib.ray_default_light(C.byref(d), kinds[kind])
        if "xform" in kw and kw["xform"] is not None:
            kw["xform"] = np.asarray(kw["xform"], np.float32).ravel()
        _set(d, **{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
        return int(self._lib.ray_scene_add_light(self._ptr, C.byref(d)))

    # -- camera ---------------------------------------------------------------------------------------------
    def AddCamera(self, **kw) -> int:
        d = _capi.CameraDesc()
        self._lib.ray_default_camera(C.byref(d))
        _set(d, **{k: (int(v) if isinstance(v, (bool, IntEnum)) else v) for k, v in kw.items()})
        return int(self._lib.ray_scene_add_camera(self._ptr, C.byref(d)))

    def set_current_cam(self, cam: int):
        self._lib.ray_scene_set_current_cam(self._ptr, cam)

    def Finalize(self):
        self._lib.ray_scene_finalize(self._ptr)

    def triangle_count(self) -> int:
        return int(self._lib.ray_scene_triangle_count(self._ptr))

    def node_count(self) -> int:
        return int(self._lib.ray_scene_node_count(self._ptr))

    def sky_bake_info(self) -> str:
        """SceneHIP only: where the last Finalize baked the sky environment map -- "device", "host" or "none" (Ray::Hip::SkyBakedOn)"""
        f = self._lib.ray_hip_sky_baked_on
        f.restype, f.argtypes = C.c_char_p, [C.c_void_p]
        return f(self._ptr).decode()


@dataclass
class ShadingNode:  # Ray::shading_node_desc_t (only the fields set here override the C++ defaults)
    type: eShadingNode = eShadingNode.Diffuse
    base_color: Optional[Tuple[float, float, float]] = None
    base_texture: Optional[int] = None
    normal_map: Optional[int] = None
    normal_map_intensity: Optional[float] = None
    mix_materials: Optional[Tuple[int, int]] = None
    roughness: Optional[float] = None
    roughness_texture: Optional[int] = None
    anisotropic: Optional[float] = None
    anisotropic_rotation: Optional[float] = None
    sheen: Optional[float] = None
    specular: Optional[float] = None
    strength: Optional[float] = None
    fresnel: Optional[float] = None
    ior: Optional[float] = None
    tint: Optional[float] = None
    metallic_texture: Optional[int] = None
    importance_sample: Optional[bool] = None
    mix_add: Optional[bool] = None


@dataclass
class PrincipledMat:  # Ray::principled_mat_desc_t
    base_color: Optional[Tuple[float, float, float]] = None
    base_texture: Optional[int] = None
    metallic: Optional[float] = None
    metallic_texture: Optional[int] = None
    specular: Optional[float] = None
    specular_texture: Optional[int] = None
    specular_tint: Optional[float] = None
    roughness: Optional[float] = None
    roughness_texture: Optional[int] = None
    anisotropic: Optional[float] = None
    anisotropic_rotation: Optional[float] = None
    sheen: Optional[float] = None
    sheen_tint: Optional[float] = None
    clearcoat: Optional[float] = None
    clearcoat_roughness: Optional[float] = None
    ior: Optional[float] = None
    transmission: Optional[float] = None
    transmission_roughness: Optional[float] = None
    emission_color: Optional[Tuple[float, float, float]] = None
    emission_texture: Optional[int] = None
    emission_strength: Optional[float] = None
    alpha: Optional[float] = None
    alpha_texture: Optional[int] = None
    normal_map: Optional[int] = None
    normal_map_intensity: Optional[float] = None
    importance_sample: Optional[bool] = None


class RendererBase:
    """Mirror of Ray::RendererBase (RendererBase.h:133-253)."""

    def __init__(self, lib, ptr):
        self._lib = lib
        self._ptr = ptr

    def __del__(self):
        if getattr(self, "_ptr", None):
            self._lib.ray_renderer_destroy(self._ptr)
            self._ptr = None

    def type(self) -> str:
        buf = C.create_string_buffer(64)
        self._lib.ray_renderer_type_name(self._ptr, buf, 64)
        return buf.value.decode()

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._lib.ray_renderer_device_name(self._ptr, buf, 256)
        return buf.value.decode()

    def size(self) -> Tuple[int, int]:
        wh = (C.c_int * 2)()
        self._lib.ray_renderer_size(self._ptr, C.byref(wh))
        return int(wh[0]), int(wh[1])

    def Resize(self, w: int, h: int):
        self._lib.ray_renderer_resize(self._ptr, w, h)

    def Clear(self, c=(0.0, 0.0, 0.0, 0.0)):
        v = (C.c_float * 4)(*c)
        self._lib.ray_renderer_clear(self._ptr, C.byref(v))

    def CreateScene(self) -> SceneBase:
        return SceneBase(self._lib, self._lib.ray_renderer_create_scene(self._ptr))

    def RenderScene(self, scene: SceneBase, region: RegionContext):
        self._lib.ray_renderer_render(self._ptr, scene._ptr, region._bind(self._lib))

    def DenoiseImage(self, region: RegionContext):
        """RendererBase::DenoiseImage(const RegionContext &): the NLM denoiser"""
        self._lib.ray_renderer_denoise(self._ptr, region._bind(self._lib))

    def InitUNetFilter(self) -> int:
        """RendererBase::InitUNetFilter(alias_memory=false): returns the number of passes (16)"""
        return int(self._lib.ray_renderer_init_unet(self._ptr))

    def DenoiseImageUNet(self, pass_index: int, region: RegionContext):
        """RendererBase::DenoiseImage(int pass, const RegionContext &): one pass of the UNet denoiser"""
        self._lib.ray_renderer_denoise_unet(self._ptr, pass_index, region._bind(self._lib))

    def _pixels(self, which: int) -> np.ndarray:
        w, h = self.size()
        out = np.empty((h, w, 4), dtype=np.float32)
        if self._lib.ray_renderer_get_pixels(self._ptr, which, out.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError(self._lib.ray_last_error().decode())
        return out

    def get_pixels_ref(self) -> np.ndarray:
        return self._pixels(0)

    def get_raw_pixels_ref(self) -> np.ndarray:
        return self._pixels(1)

    def get_aux_pixels_ref(self, buf: eAUXBuffer) -> np.ndarray:
        return self._pixels({eAUXBuffer.BaseColor: 2, eAUXBuffer.DepthNormals: 3}[buf])

    def GetStats(self) -> dict:
        st = _capi.Stats()
        self._lib.ray_renderer_get_stats(self._ptr, C.byref(st))
        return st.as_dict()

    def ResetStats(self):
        self._lib.ray_renderer_reset_stats(self._ptr)

    def render_tiled_mt(self, scene: SceneBase, tile: int, spp: int, threads: int, iterations_done: int = 0) -> float:
        """README.md:336-356 multithreading pattern: `threads` workers pull tile x tile regions from a queue and run `spp`
        iterations on each, continuing after `iterations_done`; returns wall seconds (CPU backends only)."""
        return float(self._lib.ray_renderer_render_tiled_from(self._ptr, scene._ptr, tile, iterations_done, spp, threads))


_LIBS = {}


def load_capi_library(path: str):
    lib = _LIBS.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _capi.declare(lib)
        _LIBS[path] = lib
    return lib


def create_renderer_from(lib, s: Settings, renderer_type: str) -> RendererBase:
    ptr = lib.ray_renderer_create(renderer_type.encode(), s.w, s.h, int(s.use_tex_compression), int(s.verbose))
    if not ptr:
        raise RuntimeError(lib.ray_last_error().decode())
    return RendererBase(lib, ptr)


def _hip_lib():
    lib = load_capi_library(HIP_HOST_LIB)
    if not hasattr(lib, "_hip_extras"):
        vp = C.c_void_p
        lib.ray_hip_create_scene.restype = vp
        lib.ray_hip_create_scene.argtypes = [C.c_int]
        lib.ray_hip_create_scene_ex.restype = vp
        lib.ray_hip_create_scene_ex.argtypes = [C.c_int, C.c_int]
        lib.ray_hip_export_scene.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
        lib.ray_hip_free.argtypes = [vp]
        lib.ray_hip_free.restype = None
        lib.ray_hip_pmj_table.argtypes = [C.POINTER(vp), C.POINTER(C.c_uint32)]
        lib.ray_hip_pmj_table.restype = None
        lib._hip_extras = True
    return lib


def CreateSceneHIP(verbose: bool = False, use_tex_compression: bool = False) -> SceneBase:
    """A SceneHIP without a renderer: scene construction (BVH, light tree) is host work and needs no GPU.
    use_tex_compression: settings_t::use_tex_compression for the textures added without force_no_compression."""
    lib = _hip_lib()
    return SceneBase(lib, lib.ray_hip_create_scene_ex(int(verbose), int(use_tex_compression)))


def export_scene_blob(scene: SceneBase) -> bytes:
    """Flat arrays + current camera + filter table of a finalized SceneHIP (ray_amd/csrc/scene_blob.h)."""
    lib = _hip_lib()
    p, n = C.c_void_p(), C.c_uint64()
    if lib.ray_hip_export_scene(scene._ptr, C.byref(p), C.byref(n)) != 0:
        raise RuntimeError(lib.ray_last_error().decode())
    try:
        return C.string_at(p, n.value)
    finally:
        lib.ray_hip_free(p)


def pmj_table() -> np.ndarray:
    """The PMJ02 sample table the backend renders with (uploaded by RendererHIP at start-up)."""
    lib = _hip_lib()
    p, n = C.c_void_p(), C.c_uint32()
    lib.ray_hip_pmj_table(C.byref(p), C.byref(n))
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()


def CreateRenderer(s: Settings, renderer_type: str = "HIP") -> RendererBase:
    """Ray::CreateRenderer (Ray.h:25-28) for the HIP backend.  No fallback chain: a missing GPU is an error."""
    if renderer_type != "HIP":
        raise ValueError("ray_amd only ships the HIP backend; the reference's CPU backends live in the test oracle")
    return create_renderer_from(load_capi_library(HIP_HOST_LIB), s, "HIP")
