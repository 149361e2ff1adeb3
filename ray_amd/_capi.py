"""ctypes declarations for ray_amd/host/ray_capi.h (the C view of sergcpp/Ray's public API).

The same declarations serve two libraries built from ray_capi.cpp:
  * ray_amd/host/_build/libray_hip.so  -- reference host-side scene code + RendererHIP/SceneHIP (the product)
  * oracle/_ref/libray_ref.so          -- the reference's own CPU backends (test oracle; tests only)
"""
import ctypes as C

ray_handle = C.c_uint64
INVALID_HANDLE = 0xFFFFFFFF
PHYSICAL_SKY_TEXTURE = 0xFFFFFFFE  # Ray::PhysicalSkyTexture (SceneBase.h:35)


class ShadingNodeDesc(C.Structure):  # ray_shading_node_desc
    _fields_ = [
        ("type", C.c_uint32),
        ("base_color", C.c_float * 3),
        ("base_texture", ray_handle),
        ("normal_map", ray_handle),
        ("normal_map_intensity", C.c_float),
        ("mix_materials", ray_handle * 2),
        ("roughness", C.c_float),
        ("roughness_texture", ray_handle),
        ("anisotropic", C.c_float),
        ("anisotropic_rotation", C.c_float),
        ("sheen", C.c_float),
        ("specular", C.c_float),
        ("strength", C.c_float),
        ("fresnel", C.c_float),
        ("ior", C.c_float),
        ("tint", C.c_float),
        ("metallic_texture", ray_handle),
        ("importance_sample", C.c_int32),
        ("mix_add", C.c_int32),
    ]


class PrincipledMatDesc(C.Structure):  # ray_principled_mat_desc
    _fields_ = [
        ("base_color", C.c_float * 3),
        ("base_texture", ray_handle),
        ("metallic", C.c_float),
        ("metallic_texture", ray_handle),
        ("specular", C.c_float),
        ("specular_texture", ray_handle),
        ("specular_tint", C.c_float),
        ("roughness", C.c_float),
        ("roughness_texture", ray_handle),
        ("anisotropic", C.c_float),
        ("anisotropic_rotation", C.c_float),
        ("sheen", C.c_float),
        ("sheen_tint", C.c_float),
        ("clearcoat", C.c_float),
        ("clearcoat_roughness", C.c_float),
        ("ior", C.c_float),
        ("transmission", C.c_float),
        ("transmission_roughness", C.c_float),
        ("emission_color", C.c_float * 3),
        ("emission_texture", ray_handle),
        ("emission_strength", C.c_float),
        ("alpha", C.c_float),
        ("alpha_texture", ray_handle),
        ("normal_map", ray_handle),
        ("normal_map_intensity", C.c_float),
        ("importance_sample", C.c_int32),
    ]


class MatGroupDesc(C.Structure):  # ray_mat_group_desc
    _fields_ = [("front_mat", ray_handle), ("back_mat", ray_handle), ("vtx_start", C.c_uint64), ("vtx_count", C.c_uint64)]


class MeshDesc(C.Structure):  # ray_mesh_desc
    _fields_ = [
        ("attrs", C.POINTER(C.c_float)),
        ("attrs_count", C.c_uint64),
        ("stride", C.c_int32),
        ("pos_offset", C.c_int32),
        ("nrm_offset", C.c_int32),
        ("uv_offset", C.c_int32),
        ("bnm_offset", C.c_int32),
        ("indices", C.POINTER(C.c_uint32)),
        ("indices_count", C.c_uint64),
        ("base_vertex", C.c_int32),
        ("groups", C.POINTER(MatGroupDesc)),
        ("groups_count", C.c_uint32),
        ("allow_spatial_splits", C.c_int32),
        ("use_fast_bvh_build", C.c_int32),
    ]


class TexDesc(C.Structure):  # ray_tex_desc
    _fields_ = [
        ("format", C.c_uint32),
        ("data", C.POINTER(C.c_uint8)),
        ("data_size", C.c_uint64),
        ("w", C.c_int32),
        ("h", C.c_int32),
        ("is_srgb", C.c_int32),
        ("is_normalmap", C.c_int32),
        ("is_YCoCg", C.c_int32),
        ("force_no_compression", C.c_int32),
        ("generate_mipmaps", C.c_int32),
        ("reconstruct_z", C.c_int32),
        ("mips_count", C.c_int32),
        ("convention", C.c_int32),
    ]


class LightDesc(C.Structure):  # ray_light_desc
    _fields_ = [
        ("kind", C.c_uint32),
        ("color", C.c_float * 3),
        ("direction", C.c_float * 3),
        ("angle", C.c_float),
        ("position", C.c_float * 3),
        ("radius", C.c_float),
        ("spot_size", C.c_float),
        ("spot_blend", C.c_float),
        ("width", C.c_float),
        ("height", C.c_float),
        ("doublesided", C.c_int32),
        ("sky_portal", C.c_int32),
        ("multiple_importance", C.c_int32),
        ("cast_shadow", C.c_int32),
        ("diffuse_visibility", C.c_int32),
        ("specular_visibility", C.c_int32),
        ("refraction_visibility", C.c_int32),
        ("xform", C.c_float * 16),
    ]


class CameraDesc(C.Structure):  # ray_camera_desc
    _fields_ = [
        ("type", C.c_uint32),
        ("filter", C.c_uint32),
        ("view_transform", C.c_uint32),
        ("ltype", C.c_uint32),
        ("filter_width", C.c_float),
        ("origin", C.c_float * 3),
        ("fwd", C.c_float * 3),
        ("up", C.c_float * 3),
        ("shift", C.c_float * 2),
        ("exposure", C.c_float),
        ("fov", C.c_float),
        ("gamma", C.c_float),
        ("sensor_height", C.c_float),
        ("focus_distance", C.c_float),
        ("focal_length", C.c_float),
        ("fstop", C.c_float),
        ("lens_rotation", C.c_float),
        ("lens_ratio", C.c_float),
        ("lens_blades", C.c_int32),
        ("clip_start", C.c_float),
        ("clip_end", C.c_float),
        ("mi_index", C.c_uint32),
        ("uv_index", C.c_uint32),
        ("lighting_only", C.c_int32),
        ("skip_direct_lighting", C.c_int32),
        ("skip_indirect_lighting", C.c_int32),
        ("no_background", C.c_int32),
        ("output_sh", C.c_int32),
        ("max_diff_depth", C.c_int32),
        ("max_spec_depth", C.c_int32),
        ("max_refr_depth", C.c_int32),
        ("max_transp_depth", C.c_int32),
        ("max_total_depth", C.c_int32),
        ("min_total_depth", C.c_int32),
        ("min_transp_depth", C.c_int32),
        ("clamp_direct", C.c_float),
        ("clamp_indirect", C.c_float),
        ("min_samples", C.c_int32),
        ("variance_threshold", C.c_float),
        ("regularize_alpha", C.c_float),
    ]


class EnvDesc(C.Structure):  # ray_env_desc
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("env_col", C.c_float * 3),
        ("env_map", ray_handle),
        ("back_col", C.c_float * 3),
        ("back_map", ray_handle),
        ("env_map_rotation", C.c_float),
        ("back_map_rotation", C.c_float),
        ("importance_sample", C.c_int32),
        ("envmap_resolution", C.c_int32),
        ("clouds_density", C.c_float),
        ("cirrus_clouds_amount", C.c_float),
        ("stars_brightness", C.c_float),
        ("moon_radius", C.c_float),
        ("clouds_offset_x", C.c_float),
        ("clouds_offset_z", C.c_float),
    ]


class Stats(C.Structure):  # ray_stats == RendererBase::stats_t
    _fields_ = [("t", C.c_ulonglong * 11)]

    NAMES = (
        "primary_ray_gen", "primary_trace", "primary_shade", "primary_shadow", "secondary_sort", "secondary_trace",
        "secondary_shade", "secondary_shadow", "denoise", "cache_update", "cache_resolve",
    )

    def as_dict(self):
        return {n: int(self.t[i]) for i, n in enumerate(self.NAMES)}


def declare(lib):
    """Attach argtypes/restypes for every ray_* function to a loaded library."""
    vp = C.c_void_p
    sig = {
        "ray_default_shading_node": (None, [C.POINTER(ShadingNodeDesc)]),
        "ray_default_principled": (None, [C.POINTER(PrincipledMatDesc)]),
        "ray_default_light": (None, [C.POINTER(LightDesc), C.c_uint32]),
        "ray_default_camera": (None, [C.POINTER(CameraDesc)]),
        "ray_default_env": (None, [C.POINTER(EnvDesc)]),
        "ray_last_error": (C.c_char_p, []),
        "ray_renderer_create": (vp, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ray_renderer_destroy": (None, [vp]),
        "ray_renderer_type_name": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "ray_renderer_device_name": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "ray_renderer_size": (None, [vp, C.POINTER(C.c_int * 2)]),
        "ray_renderer_resize": (None, [vp, C.c_int, C.c_int]),
        "ray_renderer_clear": (None, [vp, C.POINTER(C.c_float * 4)]),
        "ray_renderer_create_scene": (vp, [vp]),
        "ray_renderer_render": (None, [vp, vp, vp]),
        "ray_renderer_denoise": (None, [vp, vp]),
        "ray_renderer_init_unet": (C.c_int, [vp]),
        "ray_renderer_denoise_unet": (None, [vp, C.c_int, vp]),
        "ray_renderer_get_pixels": (C.c_int, [vp, C.c_int, vp]),
        "ray_renderer_get_stats": (None, [vp, C.POINTER(Stats)]),
        "ray_renderer_reset_stats": (None, [vp]),
        "ray_renderer_render_tiled_mt": (C.c_double, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "ray_renderer_render_tiled_from": (C.c_double, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ray_region_create": (vp, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "ray_region_destroy": (None, [vp]),
        "ray_region_iteration": (C.c_int, [vp]),
        "ray_region_set_iteration": (None, [vp, C.c_int]),
        "ray_scene_destroy": (None, [vp]),
        "ray_scene_set_environment": (None, [vp, C.POINTER(EnvDesc)]),
        "ray_scene_add_texture": (ray_handle, [vp, C.POINTER(TexDesc)]),
        "ray_scene_add_material_node": (ray_handle, [vp, C.POINTER(ShadingNodeDesc)]),
        "ray_scene_add_material_principled": (ray_handle, [vp, C.POINTER(PrincipledMatDesc)]),
        "ray_scene_add_mesh": (ray_handle, [vp, C.POINTER(MeshDesc)]),
        "ray_scene_add_mesh_instance": (ray_handle, [vp, ray_handle, C.POINTER(C.c_float * 16)]),
        "ray_scene_add_mesh_instance_vis": (ray_handle, [vp, ray_handle, C.POINTER(C.c_float * 16), C.c_uint]),
        "ray_scene_set_mesh_instance_transform": (None, [vp, ray_handle, C.POINTER(C.c_float * 16)]),
        "ray_scene_remove_mesh_instance": (None, [vp, ray_handle]),
        "ray_scene_remove_mesh": (None, [vp, ray_handle]),
        "ray_scene_remove_light": (None, [vp, ray_handle]),
        "ray_scene_add_light": (ray_handle, [vp, C.POINTER(LightDesc)]),
        "ray_scene_add_camera": (ray_handle, [vp, C.POINTER(CameraDesc)]),
        "ray_scene_set_current_cam": (None, [vp, ray_handle]),
        "ray_scene_finalize": (None, [vp]),
        "ray_scene_triangle_count": (C.c_uint32, [vp]),
        "ray_scene_node_count": (C.c_uint32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sorted(sig)
