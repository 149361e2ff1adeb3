"""Scene builders for the BASELINE.json configurations, written against the SceneBase mirror (ray_amd.api).

* cornell_basic      -- the scene of reference samples/00_basic (Cornell box, 32 triangles, 2 emissive tris)
* cornell_principled -- reference samples/03_principled (same box; floor = Principled + 128^2 checker RGBA8)
* atrium             -- synthetic "Sponza/Bistro-class" scene: no real asset exists offline, so
                        a deterministic procedural atrium is generated: displaced floor, colonnade of fluted
                        columns, arches, draped cloth sheets, clutter of noisy spheres/tori, emissive ceiling
                        strips.  `detail` scales the tessellation; ~0.25 M tris ("sponza") to ~3 M ("bistro").

Every builder takes an object with the SceneBase API, so the same code feeds the HIP backend and the test
oracle.  Geometry here is data (the classic Cornell measurements / procedural math), not reference code.
"""
import math
import os
from typing import Tuple

import numpy as np

from . import api
from .api import InvalidHandle, PrincipledMat, ShadingNode, eShadingNode, eTextureFormat

# ---- Cornell box ----------------------------------------------------------------------------------------------
# quad = (4 corners, normal, 4 uvs, index pattern).  Measurements are the classic Cornell data in metres with x
# negated, as used by the reference samples (samples/00_basic/main.cpp:61-148).
_Z = (0.0, 0.0)
_P_A = (0, 2, 1, 0, 3, 2)
_P_B = (0, 1, 2, 0, 2, 3)
_P_C = (0, 1, 2, 1, 3, 2)
_P_D = (0, 1, 2, 2, 1, 3)

_CORNELL_QUADS = [
    # floor
    (((0.0, 0.0, -0.5592), (0.0, 0.0, 0.0), (-0.5528, 0.0, 0.0), (-0.5496, 0.0, -0.5592)), (0.0, 1.0, 0.0),
     ((1.0, 1.0), (1.0, 0.0), (0.0, 0.0), (0.0, 1.0)), _P_A),
    # back wall
    (((0.0, 0.0, -0.5592), (-0.5496, 0.0, -0.5592), (-0.556, 0.5488, -0.5592), (0.0, 0.5488, -0.5592)), (0.0, 0.0, 1.0),
     (_Z, _Z, _Z, _Z), _P_A),
    # ceiling
    (((-0.556, 0.5488, -0.5592), (0.0, 0.5488, -0.5592), (0.0, 0.5488, 0.0), (-0.556, 0.5488, 0.0)), (0.0, -1.0, 0.0),
     (_Z, _Z, _Z, _Z), _P_B),
    # left wall
    (((-0.5528, 0.0, 0.0), (-0.5496, 0.0, -0.5592), (-0.556, 0.5488, 0.0), (-0.556, 0.5488, -0.5592)), (1.0, 0.0, 0.0),
     (_Z, _Z, _Z, _Z), _P_C),
    # right wall
    (((0.0, 0.0, -0.5592), (0.0, 0.0, 0.0), (0.0, 0.5488, -0.5592), (0.0, 0.5488, 0.0)), (-1.0, 0.0, 0.0),
     (_Z, _Z, _Z, _Z), _P_D),
    # light
    (((-0.213, 0.5478, -0.227), (-0.343, 0.5478, -0.227), (-0.343, 0.5478, -0.332), (-0.213, 0.5478, -0.332)),
     (0.0, -1.0, 0.0), (_Z, _Z, _Z, _Z), _P_B),
]

# blocks: base corners (x, z) a, b, c, d and height; faces listed as in the sample
_SHORT = dict(a=(-0.240464, -0.271646), b=(-0.082354, -0.224464), c=(-0.129536, -0.066354), d=(-0.287646, -0.113536),
              h=0.165, n1=(0.285951942, 0.0, -0.958243966), n2=(-0.958243966, 0.0, -0.285951942),
              n3=(0.958243966, 0.0, 0.285951942), n4=(-0.285951942, 0.0, 0.958243966))
_TALL = dict(a=(-0.471239, -0.405353), b=(-0.313647, -0.454239), c=(-0.264761, -0.296647), d=(-0.422353, -0.247761),
             h=0.33, n1=(-0.296278358, 0.0, -0.955101609), n2=(0.955101609, 0.0, -0.296278358),
             n3=(-0.955101609, 0.0, 0.296278358), n4=(0.296278358, 0.0, 0.955101609))


def _block_quads(kind: str):
    def v(p, y):
        return (p[0], y, p[1])

    if kind == "short":
        s = _SHORT
        a, b, c, d, h = s["a"], s["b"], s["c"], s["d"], s["h"]
        return [
            ((v(a, 0.0), v(a, h), v(b, h), v(b, 0.0)), s["n1"], (_Z,) * 4, _P_B),
            ((v(a, 0.0), v(a, h), v(d, h), v(d, 0.0)), s["n2"], (_Z,) * 4, _P_A),
            ((v(b, 0.0), v(b, h), v(c, h), v(c, 0.0)), s["n3"], (_Z,) * 4, _P_B),
            ((v(d, 0.0), v(d, h), v(c, h), v(c, 0.0)), s["n4"], (_Z,) * 4, _P_A),
            ((v(a, h), v(b, h), v(c, h), v(d, h)), (0.0, 1.0, 0.0), (_Z,) * 4, _P_A),
        ]
    s = _TALL
    a, b, c, d, h = s["a"], s["b"], s["c"], s["d"], s["h"]
    return [
        ((v(a, 0.0), v(a, h), v(b, h), v(b, 0.0)), s["n1"], (_Z,) * 4, _P_B),
        ((v(c, 0.0), v(c, h), v(b, h), v(b, 0.0)), s["n2"], (_Z,) * 4, _P_A),
        ((v(a, 0.0), v(a, h), v(d, h), v(d, 0.0)), s["n3"], (_Z,) * 4, _P_A),
        ((v(d, 0.0), v(d, h), v(c, h), v(c, 0.0)), s["n4"], (_Z,) * 4, _P_A),
        ((v(a, h), v(b, h), v(c, h), v(d, h)), (0.0, 1.0, 0.0), (_Z,) * 4, _P_A),
    ]


def cornell_mesh_arrays(quads=None) -> Tuple[np.ndarray, np.ndarray]:
    """(attrs [64, 8] = pos3 nrm3 uv2, indices [96]) of the Cornell box mesh (or of the given subset of its quads)."""
    quads = (_CORNELL_QUADS + _block_quads("short") + _block_quads("tall")) if quads is None else quads
    attrs, idx = [], []
    for qi, (corners, n, uvs, pat) in enumerate(quads):
        for c, uv in zip(corners, uvs):
            attrs.append((*c, *n, *uv))
        idx.extend(4 * qi + k for k in pat)
    return np.asarray(attrs, dtype=np.float32), np.asarray(idx, dtype=np.uint32)


def checkerboard(res: int = 128, square: int = 16) -> np.ndarray:
    """samples/03_principled GenerateCheckerboard: RGBA8, 10/250 grey squares."""
    i, j = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    dark = ((j // square + i // square) % 2) == 0
    img = np.empty((res, res, 4), dtype=np.uint8)
    img[..., :3] = np.where(dark, 10, 250)[..., None]
    img[..., 3] = 255
    return img


def _cornell_camera(scene, **cam_overrides):
    kw = dict(type=0, origin=(-0.278, 0.273, 0.8), fwd=(0.0, 0.0, -1.0), fov=39.1463)
    kw.update(cam_overrides)
    cam = scene.AddCamera(**kw)
    scene.set_current_cam(cam)
    return cam


def cornell_basic(scene, **cam_overrides):
    """reference samples/00_basic/main.cpp:30-186"""
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    mat1 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    mat2 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    mat3 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    mat4 = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays()
    # groups exactly as the sample passes them ({mat, vtx_start, vtx_count}); the light has no back material
    groups = [(mat1, None, 0, 18), (mat2, None, 19, 6), (mat3, None, 25, 6), (mat4, 0xFFFFFFFF, 31, 6), (mat1, None, 37, 60)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_fresnel_mix(scene, **cam_overrides):
    """the Cornell box with its blocks in a Fresnel-weighted mix (a Mix node WITH an ior: its weight takes the medium outside the
    surface from the ray's ior stack, ShadeRef.cpp mix loop) of a diffuse and a glossy material -- and nothing that refracts: the scene
    whose passes leave the rays' ior plane alone although one of its nodes asks for the stack"""
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    lamp = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    gloss = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.9, 0.9, 0.9), roughness=0.1))
    coated = scene.AddMaterial(ShadingNode(type=eShadingNode.Mix, strength=1.0, ior=1.5, mix_materials=(grey, gloss)))
    attrs, idx = cornell_mesh_arrays()
    groups = [(grey, None, 0, 18), (red, None, 19, 6), (green, None, 25, 6), (lamp, 0xFFFFFFFF, 31, 6), (coated, None, 37, 60)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_filmic(scene, **cam_overrides):
    """cornell_basic through a look-up-table view transform (eViewTransform.Filmic_HighContrast = 8, TonemapRef.cpp:15-26)
    and a display gamma: exercises TonemapFilmic + the pow() branch of Tonemap()"""
    kw = dict(view_transform=8, gamma=2.2, exposure=-1.0)
    kw.update(cam_overrides)
    cornell_basic(scene, **kw)


def cornell_principled(scene, **cam_overrides):
    """reference samples/03_principled/main.cpp:30-205"""
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    tex = scene.AddTexture(checkerboard(128, 16))
    mat0 = scene.AddMaterial(PrincipledMat(base_texture=tex, roughness=0.25, roughness_texture=tex))
    mat1 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    mat2 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    mat3 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    mat4 = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays()
    groups = [(mat0, None, 0, 6), (mat1, None, 6, 12), (mat2, None, 19, 6), (mat3, None, 25, 6),
              (mat4, 0xFFFFFFFF, 31, 6), (mat1, None, 37, 60)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def bump_normal_map(res: int = 64) -> np.ndarray:
    """tangent-space normal map of a sine bump field, RGBA8 (x, y, z in 0..255; z reconstructed from x, y)"""
    i, j = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    dx = 0.35 * np.cos(2 * math.pi * j / 16.0)
    dy = 0.35 * np.cos(2 * math.pi * i / 16.0)
    n = np.stack([-dx, -dy, np.ones_like(dx)], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    img = np.empty((res, res, 4), dtype=np.uint8)
    img[..., :3] = np.clip(np.rint((n * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return img


def _translate(x, y, z, rot_x_deg=0.0, rot_z_deg=0.0):
    """column-major 4x4 (the layout Ray's AddLight(xform) takes): R_z * R_x then translation"""
    ax, az = math.radians(rot_x_deg), math.radians(rot_z_deg)
    rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    rz = np.array([[math.cos(az), -math.sin(az), 0], [math.sin(az), math.cos(az), 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = rz @ rx
    m[:3, 3] = (x, y, z)
    return m.T.astype(np.float32).ravel()  # column-major


def cornell_lights(scene, light_flags=None, **cam_overrides):
    """Cornell box lit by every analytic light type (all visible to secondary rays -> IntersectAreaLights, MIS) with a
    zoo of shading nodes: Principled floor with textures, Oren-Nayar walls, normal-mapped glossy back wall, refractive
    short block, Mix(diffuse, transparent) tall block.  Not a reference sample; built to cover SURVEY 8 a11-a14."""
    scene.SetEnvironment(env_col=(0.02, 0.03, 0.05))
    tex = scene.AddTexture(checkerboard(128, 16))
    nmap = scene.AddTexture(bump_normal_map(64), is_srgb=False, is_normalmap=True)
    floor_m = scene.AddMaterial(PrincipledMat(base_texture=tex, roughness=0.4, specular=0.5, clearcoat=0.5,
                                              clearcoat_roughness=0.1, sheen=0.3))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5), roughness=0.6))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.05, 0.05), roughness=0.3))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.05, 0.5, 0.05)))
    back = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.7, 0.7, 0.75), roughness=0.3,
                                         normal_map=nmap, normal_map_intensity=0.8))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=15.0, importance_sample=True))
    glass = scene.AddMaterial(ShadingNode(type=eShadingNode.Refractive, base_color=(0.9, 0.95, 1.0), roughness=0.05, ior=1.45))
    blue = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.1, 0.15, 0.6)))
    clear = scene.AddMaterial(ShadingNode(type=eShadingNode.Transparent, base_color=(0.9, 0.9, 0.9)))
    veil = scene.AddMaterial(ShadingNode(type=eShadingNode.Mix, mix_materials=(blue, clear), strength=0.5))
    attrs, idx = cornell_mesh_arrays()
    # floor | back wall | ceiling | left | right | light quad | short block (5 quads) | tall block (5 quads)
    groups = [(floor_m, None, 0, 6), (back, None, 6, 6), (grey, None, 12, 6), (red, None, 19, 6), (green, None, 25, 6),
              (emit, 0xFFFFFFFF, 31, 6), (glass, glass, 37, 30), (veil, veil, 67, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)

    # `light_flags`: per-light overrides of the light_desc flags (tests: cast_shadow, *_visibility, multiple_importance)
    lf = light_flags or {}
    scene.AddLight("sphere", color=(6.0, 5.0, 4.0), position=(-0.12, 0.42, -0.12), radius=0.025, **lf.get("sphere", {}))
    scene.AddLight("spot", color=(20.0, 20.0, 26.0), position=(-0.47, 0.50, -0.10), direction=(0.45, -0.85, -0.3),
                   radius=0.015, spot_size=55.0, spot_blend=0.2, **lf.get("spot", {}))
    scene.AddLight("rect", color=(8.0, 8.0, 7.0), width=0.16, height=0.10, xform=_translate(-0.30, 0.52, -0.44), **lf.get("rect", {}))
    scene.AddLight("disk", color=(2.0, 7.0, 2.5), width=0.09, height=0.12, doublesided=True,
                   xform=_translate(-0.545, 0.30, -0.30, rot_z_deg=-90.0), **lf.get("disk", {}))
    scene.AddLight("line", color=(7.0, 2.0, 2.0), radius=0.006, height=0.25, xform=_translate(-0.02, 0.35, -0.30, rot_x_deg=90.0),
                   **lf.get("line", {}))
    scene.AddLight("directional", color=(1.2, 1.1, 1.0), direction=(0.25, -0.45, -1.0), angle=4.0, **lf.get("directional", {}))
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def rgbe_sky(w: int = 128, h: int = 64) -> np.ndarray:
    """small latlong HDR sky in RGBE8 (what Ray's env maps are: RGBA8888 bytes = mantissas + shared exponent): blue-ish
    gradient, warm horizon band and a sun 3 orders of magnitude brighter -- something worth importance sampling"""
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    theta, phi = v * math.pi, u * 2 * math.pi
    up = np.cos(theta)
    sky = np.stack([0.25 + 0.2 * (1 - up), 0.35 + 0.25 * (1 - up), 0.7 + 0.1 * up], axis=-1) * np.clip(up * 0.5 + 0.6, 0.05, None)[..., None]
    horizon = np.exp(-((theta - math.pi / 2) / 0.12) ** 2)[..., None] * np.array([0.9, 0.55, 0.3])
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], axis=-1)
    sun_dir = _unit((0.35, 0.75, 0.55))
    sun = (np.clip((d * np.asarray(sun_dir)).sum(-1), 0, 1) ** 600)[..., None] * np.array([900.0, 820.0, 640.0])
    rgb = (sky + horizon + sun).astype(np.float64)
    m = rgb.max(axis=-1)
    e = np.where(m > 1e-32, np.ceil(np.log2(np.maximum(m, 1e-32))), 0.0)
    scale = np.where(m > 1e-32, 256.0 / np.exp2(e), 0.0)
    img = np.zeros((h, w, 4), dtype=np.uint8)
    img[..., :3] = np.clip(np.floor(rgb * scale[..., None]), 0, 255).astype(np.uint8)
    img[..., 3] = np.where(m > 1e-32, np.clip(e + 128, 0, 255), 0).astype(np.uint8)
    return img


def cornell_env(scene, **cam_overrides):
    """Cornell box without its ceiling, lit only by an importance-sampled HDR environment (env-map quadtree, SURVEY 8
    a11): Scene::PrepareEnvMapQTree builds the tree on the host, SampleLightSource / Evaluate_EnvColor use it."""
    sky = scene.AddTexture(rgbe_sky(), is_srgb=False)
    scene.SetEnvironment(env_col=(1.0, 1.0, 1.0), back_col=(1.0, 1.0, 1.0), env_map=sky, back_map=sky,
                         env_map_rotation=0.3, back_map_rotation=0.3, importance_sample=True)
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.05, 0.05)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.05, 0.5, 0.05)))
    shiny = scene.AddMaterial(PrincipledMat(base_color=(0.8, 0.8, 0.85), metallic=1.0, roughness=0.15))
    # floor, back wall, left wall, right wall (no ceiling, no light quad), then the two blocks
    q = _CORNELL_QUADS
    attrs, idx = cornell_mesh_arrays([q[0], q[1], q[3], q[4]] + _block_quads("short") + _block_quads("tall"))
    groups = [(grey, None, 0, 12), (red, None, 12, 6), (green, None, 18, 6), (grey, None, 24, 30), (shiny, None, 54, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_sky(scene, night: bool = False, envmap_resolution: int = 256, **cam_overrides):
    """Cornell box without ceiling and back wall under the PHYSICAL SKY: the environment is
    Ray::PhysicalSkyTexture, lit by a directional light (the sun).  Camera rays and the mirror-like block's reflections that leave
    the box are narrower than the baked map resolves and are evaluated analytically (ShadeSkyPrimary / ShadeSkySecondary ->
    IntegrateScattering: air, the cloud layer with its shadow marches, cirrus, the sun's disk; at `night` stars and the moon);
    diffuse bounces read the map the host baked from the same integrator, importance-sampled through its quadtree."""
    scene.SetEnvironment(env_col=(1.0, 1.0, 1.0), back_col=(1.0, 1.0, 1.0), env_map=api.PhysicalSkyTexture, back_map=api.PhysicalSkyTexture,
                         importance_sample=True, envmap_resolution=envmap_resolution, clouds_density=0.6, cirrus_clouds_amount=0.6)
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.05, 0.05)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.05, 0.5, 0.05)))
    mirror = scene.AddMaterial(PrincipledMat(base_color=(0.9, 0.9, 0.9), metallic=1.0, roughness=0.0))
    q = _CORNELL_QUADS
    attrs, idx = cornell_mesh_arrays([q[0], q[3], q[4]] + _block_quads("short") + _block_quads("tall"))  # floor, left, right; two blocks
    groups = [(grey, None, 0, 6), (red, None, 6, 6), (green, None, 12, 6), (grey, None, 18, 30), (mirror, None, 48, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    if night:  # the sun below the horizon: moonlight, stars, the moon's disk
        scene.AddLight("directional", color=(12.0, 11.0, 10.0), direction=(0.2, 0.3, -1.0), angle=0.5)
    else:
        scene.AddLight("directional", color=(12.0, 11.0, 10.0), direction=(0.3, -0.5, -1.0), angle=0.6)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_principled_zoo(scene, **cam_overrides):
    """every quad group of the Cornell mesh gets its own corner of principled_mat_desc_t / shading_node_desc_t: metals
    with anisotropy, rough and clear transmission, sheen + tints, clearcoat, emission inside a Principled material,
    alpha (-> Mix with Transparent), Glossy / Refractive / Diffuse nodes with textures, a Mix with mix_add"""
    scene.SetEnvironment(env_col=(0.05, 0.06, 0.08))
    tex = scene.AddTexture(checkerboard(64, 8), generate_mipmaps=True)
    alpha = scene.AddTexture(checkerboard(32, 4), is_srgb=False)
    nmap = scene.AddTexture(bump_normal_map(64), is_srgb=False, is_normalmap=True)
    P, N = PrincipledMat, ShadingNode
    floor = scene.AddMaterial(P(base_texture=tex, metallic=1.0, roughness=0.3, anisotropic=0.8, anisotropic_rotation=0.25))
    back = scene.AddMaterial(P(base_color=(0.7, 0.6, 0.9), roughness=0.5, sheen=1.0, sheen_tint=0.6, specular=0.8, specular_tint=0.7))
    ceil = scene.AddMaterial(P(base_color=(0.8, 0.8, 0.8), roughness=0.7, emission_color=(1.0, 0.8, 0.6), emission_strength=0.5))
    left = scene.AddMaterial(P(base_color=(0.6, 0.1, 0.1), clearcoat=1.0, clearcoat_roughness=0.05, roughness=0.6, normal_map=nmap,
                               normal_map_intensity=0.5))
    right = scene.AddMaterial(N(type=eShadingNode.Glossy, base_texture=tex, roughness=0.15, anisotropic=0.5))
    emit = scene.AddMaterial(N(type=eShadingNode.Emissive, strength=60.0, importance_sample=True))
    short = scene.AddMaterial(P(base_color=(0.9, 1.0, 0.9), transmission=1.0, roughness=0.05, ior=1.5, transmission_roughness=0.2))
    tall_a = scene.AddMaterial(P(base_color=(0.2, 0.4, 0.9), roughness=0.4, alpha=0.6, alpha_texture=alpha))
    d1 = scene.AddMaterial(N(type=eShadingNode.Diffuse, base_color=(0.8, 0.3, 0.1), roughness=0.9))
    d2 = scene.AddMaterial(N(type=eShadingNode.Refractive, base_color=(1.0, 1.0, 1.0), roughness=0.3, ior=1.33))
    tall_b = scene.AddMaterial(N(type=eShadingNode.Mix, mix_materials=(d1, d2), strength=0.35, mix_add=True))
    attrs, idx = cornell_mesh_arrays()
    # floor | ceiling | back | left | right | light | short block: 5 quads | tall block: 5 quads
    groups = [(floor, None, 0, 6), (ceil, None, 6, 6), (back, None, 12, 6), (left, None, 19, 6), (right, None, 25, 6),
              (emit, 0xFFFFFFFF, 31, 6), (short, short, 37, 30), (tall_a, tall_a, 67, 12), (tall_b, tall_b, 79, 18)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    scene.AddLight("sphere", color=(3.0, 3.0, 3.5), position=(-0.40, 0.45, -0.15), radius=0.02)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def mutate_instances_scene(scene):
    """a cornell_instances scene after it was rendered: drop an instance and a light, move another instance, add a light,
    change the environment, Finalize again.  The scene's arrays are sparse pools -- freed slots keep stale contents --
    and the TLAS / light tree are rebuilt: what a backend must re-upload (RendererHIP: version counter)."""
    h = scene._test_handles
    scene.RemoveMeshInstance(h["hidden"])
    scene.SetMeshInstanceTransform(h["scaled"], _xform(translate=(-0.33, 0.05, 0.05), rot_y_deg=-15.0, scale=(0.5, 1.5, 0.7)))
    scene.RemoveLight(h["light_a"])
    scene.AddLight("sphere", color=(2.0, 6.0, 3.0), position=(-0.45, 0.35, -0.35), radius=0.03)
    scene.SetEnvironment(env_col=(0.10, 0.04, 0.02))
    scene.Finalize()


def cornell_instances_mutable(scene, **cam_overrides):
    """cornell_instances plus two analytic lights, with the handles the mutation above needs kept on the scene object"""
    scene.SetEnvironment(env_col=(0.02, 0.03, 0.05))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays(_CORNELL_QUADS)
    room = scene.AddMesh(attrs, idx, [(grey, None, 0, 18), (red, None, 18, 6), (green, None, 24, 6), (emit, 0xFFFFFFFF, 30, 6)])
    attrs, idx = cornell_mesh_arrays(_block_quads("short"))
    block = scene.AddMesh(attrs, idx, [(grey, None, 0, 30)])
    scene.AddMeshInstance(room)
    scene.AddMeshInstance(block)
    scaled = scene.AddMeshInstance(block, _xform(translate=(-0.30, 0.0, 0.10), rot_y_deg=35.0, scale=(0.6, 1.8, 0.6)))
    hidden = scene.AddMeshInstance(block, _xform(translate=(0.12, 0.30, -0.10), rot_y_deg=-20.0, rot_z_deg=25.0, scale=(0.5, 0.5, 0.5)))
    scene.AddMeshInstance(block, _xform(translate=(-0.02, 0.0, 0.22), rot_y_deg=10.0, scale=(0.7, 1.2, 0.3)))
    light_a = scene.AddLight("sphere", color=(6.0, 5.0, 4.0), position=(-0.12, 0.42, -0.12), radius=0.025)
    scene.AddLight("rect", color=(8.0, 8.0, 7.0), width=0.16, height=0.10, xform=_translate(-0.30, 0.52, -0.44))
    scene._test_handles = dict(scaled=scaled, hidden=hidden, light_a=light_a)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_delta_lights(scene, **cam_overrides):
    """corner cases of the light and texture code: delta lights (sphere / spot of radius 0, directional of angle 0 -- never hit
    by rays, pdf-less sampling), an emitter that is NOT importance-sampled, a textured emitter, a two-sided group with
    different front / back materials, non-power-of-two textures with mip chains, a 1x1 texture"""
    scene.SetEnvironment(env_col=(0.01, 0.01, 0.015))
    i, j = np.meshgrid(np.arange(20), np.arange(48), indexing="ij")
    npot = np.empty((20, 48, 4), dtype=np.uint8)
    npot[..., 0] = (40 + 4 * j) % 256
    npot[..., 1] = (30 + 11 * i) % 256
    npot[..., 2] = np.where((i // 3 + j // 5) % 2 == 0, 230, 40)
    npot[..., 3] = 255
    t_npot = scene.AddTexture(npot, generate_mipmaps=True)
    t_one = scene.AddTexture(np.array([[[200, 120, 60, 255]]], dtype=np.uint8))
    floor = scene.AddMaterial(PrincipledMat(base_texture=t_npot, roughness=0.5, specular=0.4))
    back = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_texture=t_one, roughness=0.2))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.1, 0.6, 0.1), roughness=0.35))
    emit_plain = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=12.0, importance_sample=False))
    emit_tex = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, base_texture=t_npot, strength=6.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays()
    # floor | ceiling | back | left | right | light | short block | tall block (front: textured emitter, back: grey)
    groups = [(floor, None, 0, 6), (grey, None, 6, 6), (back, None, 12, 6), (red, green, 19, 6), (green, red, 25, 6),
              (emit_plain, 0xFFFFFFFF, 31, 6), (grey, None, 37, 30), (emit_tex, grey, 67, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    scene.AddLight("sphere", color=(0.8, 0.7, 0.6), position=(-0.12, 0.42, -0.12), radius=0.0)
    scene.AddLight("spot", color=(2.0, 2.0, 2.6), position=(-0.47, 0.50, -0.10), direction=(0.45, -0.85, -0.3), radius=0.0,
                   spot_size=40.0, spot_blend=0.5)
    scene.AddLight("directional", color=(0.6, 0.55, 0.5), direction=(0.25, -0.45, -1.0), angle=0.0)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def empty_scene(scene, **cam_overrides):
    """no geometry at all (no TLAS): every ray leaves into the background"""
    scene.SetEnvironment(env_col=(0.3, 0.4, 0.5), back_col=(0.1, 0.2, 0.3))
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def lights_only_scene(scene, **cam_overrides):
    """analytic lights and a background, but no geometry (primary rays never see analytic lights: IntersectAreaLights runs
    on secondary rays only)"""
    scene.SetEnvironment(env_col=(0.02, 0.02, 0.03), back_col=(0.05, 0.05, 0.08))
    scene.AddLight("sphere", color=(6.0, 5.0, 4.0), position=(-0.28, 0.27, -0.3), radius=0.08)
    scene.AddLight("directional", color=(1.0, 1.0, 1.0), direction=(0.0, -1.0, -0.2), angle=2.0)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_portals(scene, **cam_overrides):
    """cornell_env's open box under the RGBE sky, with the opening covered by sky portals -- a rectangular and a disk
    light with sky_portal = true (SceneBase.h rect/disk_light_desc_t): SampleLightSource takes their colour from the
    environment map, IntersectAreaLights / Evaluate_LightColor treat them as windows onto it."""
    sky = scene.AddTexture(rgbe_sky(), is_srgb=False)
    scene.SetEnvironment(env_col=(1.0, 1.0, 1.0), back_col=(1.0, 1.0, 1.0), env_map=sky, back_map=sky,
                         env_map_rotation=0.3, back_map_rotation=0.3, importance_sample=False)
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.05, 0.05)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.05, 0.5, 0.05)))
    shiny = scene.AddMaterial(PrincipledMat(base_color=(0.8, 0.8, 0.85), metallic=1.0, roughness=0.15))
    q = _CORNELL_QUADS
    attrs, idx = cornell_mesh_arrays([q[0], q[1], q[3], q[4]] + _block_quads("short") + _block_quads("tall"))
    groups = [(grey, None, 0, 12), (red, None, 12, 6), (green, None, 18, 6), (grey, None, 24, 30), (shiny, None, 54, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    # facing down into the box (rect/disk lights emit along -y of their frame after the rotation about x)
    scene.AddLight("rect", color=(1.0, 1.0, 1.0), width=0.30, height=0.50, sky_portal=True,
                   xform=_translate(-0.41, 0.5488, -0.28))
    scene.AddLight("disk", color=(1.0, 1.0, 1.0), width=0.24, height=0.24, sky_portal=True,
                   xform=_translate(-0.13, 0.5488, -0.28))
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def cornell_textures(scene, **cam_overrides):
    """Cornell box whose floor / back wall / blocks carry an RGB888 base colour map, an R8 roughness map and a normal map,
    all with mip chains and WITHOUT force_no_compression: under settings_t::use_tex_compression they land in the BC3
    (YCoCg), BC4 and BC5 storages (SceneCPU.cpp:60-200), otherwise in RGB / R / RG."""
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    res = 64
    i, j = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    rgb = np.empty((res, res, 3), dtype=np.uint8)  # smooth gradients + a grid: lossy under BC3
    rgb[..., 0] = (40 + 180 * (0.5 + 0.5 * np.sin(i * 0.31) * np.cos(j * 0.17))).astype(np.uint8)
    rgb[..., 1] = (30 + 200 * (j / (res - 1.0))).astype(np.uint8)
    rgb[..., 2] = np.where(((i // 8) + (j // 8)) % 2 == 0, 220, 60).astype(np.uint8)
    rough = (20 + 200 * (0.5 + 0.5 * np.sin(i * 0.23 + j * 0.41))).astype(np.uint8)[..., None]
    kw = dict(generate_mipmaps=True, force_no_compression=False)
    t_rgb = scene.AddTexture(rgb, fmt=eTextureFormat.RGB888, is_srgb=True, **kw)
    t_rough = scene.AddTexture(rough, fmt=eTextureFormat.R8, is_srgb=False, **kw)
    t_nrm = scene.AddTexture(bump_normal_map(res), is_srgb=False, is_normalmap=True, **kw)
    tex = scene.AddMaterial(PrincipledMat(base_texture=t_rgb, roughness=1.0, roughness_texture=t_rough, normal_map=t_nrm,
                                          normal_map_intensity=1.0, specular=0.5))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays()
    groups = [(tex, None, 0, 6), (grey, None, 6, 6), (tex, None, 12, 6), (red, None, 19, 6), (green, None, 25, 6),
              (emit, 0xFFFFFFFF, 31, 6), (tex, None, 37, 60)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def random_blocks(fmt: eTextureFormat, w: int, h: int, mips: int, seed: int) -> np.ndarray:
    """`mips` levels of 4x4 blocks with random bytes: every bit pattern is a valid BC1 / BC3 / BC4 / BC5 block, so this walks
    all selector values, both end-point orders and the tile padding of sizes that are not multiples of four"""
    rs = np.random.RandomState(4200 + seed)
    block = 16 if fmt in (eTextureFormat.BC3, eTextureFormat.BC5) else 8
    out = []
    for _ in range(mips):
        out.append(rs.randint(0, 256, size=((w + 3) // 4) * ((h + 3) // 4) * block, dtype=np.uint8))
        w, h = max(w // 2, 1), max(h // 2, 1)
    return np.concatenate(out)


def cornell_block_textures(scene, **cam_overrides):
    """Cornell box textured from pre-compressed inputs (eTextureFormat::BC1 / BC3 / BC4 / BC5 -> the four block storages,
    SceneCPU.cpp:160-178): base colour maps in BC1 and BC3, a roughness map in BC4, a two-channel normal map in BC5
    (z reconstructed), with mip levels and odd sizes"""
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    F = eTextureFormat
    t_bc1 = scene.AddTexture(random_blocks(F.BC1, 24, 20, 3, 1), fmt=F.BC1, size=(24, 20), mips_count=3, is_srgb=True)
    t_bc3 = scene.AddTexture(random_blocks(F.BC3, 18, 30, 2, 2), fmt=F.BC3, size=(18, 30), mips_count=2, is_srgb=True)
    t_bc4 = scene.AddTexture(random_blocks(F.BC4, 32, 32, 4, 3), fmt=F.BC4, size=(32, 32), mips_count=4, is_srgb=False)
    t_bc5 = scene.AddTexture(random_blocks(F.BC5, 16, 16, 1, 4), fmt=F.BC5, size=(16, 16), mips_count=1, is_srgb=False, is_normalmap=True)
    m1 = scene.AddMaterial(PrincipledMat(base_texture=t_bc1, roughness=1.0, roughness_texture=t_bc4, specular=0.5))
    m3 = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_texture=t_bc3, normal_map=t_bc5, normal_map_intensity=0.6))
    m4 = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_texture=t_bc1, roughness=0.8, roughness_texture=t_bc4))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays()
    groups = [(m1, None, 0, 6), (m3, None, 6, 6), (m3, None, 12, 6), (red, None, 19, 6), (m4, None, 25, 6),
              (emit, 0xFFFFFFFF, 31, 6), (m1, None, 37, 30), (m4, None, 67, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def _xform(translate=(0.0, 0.0, 0.0), rot_y_deg=0.0, rot_z_deg=0.0, scale=(1.0, 1.0, 1.0)) -> np.ndarray:
    """4x4 T * Ry * Rz * S in the layout the reference takes (column vectors, translation in elements 12..14)"""
    cy, sy = np.cos(np.radians(rot_y_deg)), np.sin(np.radians(rot_y_deg))
    cz, sz = np.cos(np.radians(rot_z_deg)), np.sin(np.radians(rot_z_deg))
    ry = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1]], dtype=np.float64)
    rz = np.array([[cz, -sz, 0, 0], [sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    sc = np.diag([scale[0], scale[1], scale[2], 1.0])
    m = ry @ rz @ sc
    m[:3, 3] = translate
    return m.T.astype(np.float32).copy()  # row-major memory of the transpose == column-major memory of m


def cornell_instances(scene, **cam_overrides):
    """Two-level hierarchy: the room is one mesh instance, the short block is ONE mesh instanced five times (shared BLAS)
    with translations, rotations and a non-uniform scale, three of them with restricted visibility (hidden from camera
    rays / from shadow rays / from diffuse bounces), one made of a transparent-mix material, and the tall block is a
    second instanced mesh.  Covers Traverse_TLAS_* over several leaves, TransformPoint/Direction/Normal, the ray-type
    visibility masks (mesh_instance_t::ray_visibility) and transparency rounds across instances."""
    scene.SetEnvironment(env_col=(0.02, 0.03, 0.05))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    shiny = scene.AddMaterial(PrincipledMat(base_color=(0.8, 0.7, 0.3), metallic=1.0, roughness=0.2))
    glass = scene.AddMaterial(ShadingNode(type=eShadingNode.Transparent, base_color=(0.7, 0.9, 0.7)))
    blue = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.1, 0.2, 0.7)))
    veil = scene.AddMaterial(ShadingNode(type=eShadingNode.Mix, strength=0.5, mix_materials=(blue, glass)))
    q = _CORNELL_QUADS
    attrs, idx = cornell_mesh_arrays(q)  # floor, ceiling, back, left, right, light
    room = scene.AddMesh(attrs, idx, [(grey, None, 0, 18), (red, None, 18, 6), (green, None, 24, 6), (emit, 0xFFFFFFFF, 30, 6)])
    attrs, idx = cornell_mesh_arrays(_block_quads("short"))
    block = scene.AddMesh(attrs, idx, [(grey, None, 0, 30)])
    block_shiny = scene.AddMesh(attrs, idx, [(shiny, None, 0, 30)])
    block_veil = scene.AddMesh(attrs, idx, [(veil, None, 0, 30)])
    attrs, idx = cornell_mesh_arrays(_block_quads("tall"))
    tall = scene.AddMesh(attrs, idx, [(grey, None, 0, 30)])
    scene.AddMeshInstance(room)
    scene.AddMeshInstance(block)                                                          # where the sample has it
    scene.AddMeshInstance(block, _xform(translate=(-0.30, 0.0, 0.10), rot_y_deg=35.0, scale=(0.6, 1.8, 0.6)))
    scene.AddMeshInstance(block, _xform(translate=(0.12, 0.30, -0.10), rot_y_deg=-20.0, rot_z_deg=25.0, scale=(0.5, 0.5, 0.5)),
                          camera=False)                                                   # seen only through its shadow / bounces
    scene.AddMeshInstance(block_shiny, _xform(translate=(-0.12, 0.165, -0.02), rot_y_deg=60.0, scale=(0.5, 0.6, 0.5)), shadow=False)
    scene.AddMeshInstance(block_veil, _xform(translate=(-0.02, 0.0, 0.22), rot_y_deg=10.0, scale=(0.7, 1.2, 0.3)), diffuse=False)
    scene.AddMeshInstance(tall, _xform(translate=(0.02, 0.0, 0.03), rot_y_deg=-8.0))
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


# ---- procedural atrium ("Sponza / Bistro class") ---------------------------------------------------------------
def _grid(nu: int, nv: int, fn, flip=False):
    """Tessellated parametric patch.  fn(u, v) -> (P[...,3], N[...,3]); returns attrs [n,8], tri indices."""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    P, N = fn(u, v)
    uv = np.stack([u * 4.0, v * 4.0], axis=-1)
    attrs = np.concatenate([P, N, uv], axis=-1).reshape(-1, 8).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b, c, d = a + (nv + 1), a + (nv + 1) + 1, a + 1
    tris = np.stack([a, b, c, a, c, d], axis=-1) if not flip else np.stack([a, c, b, a, d, c], axis=-1)
    return attrs, tris.reshape(-1).astype(np.uint32)


def _normalize(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-12)


def _noise(p, seed):
    """cheap deterministic value noise from sines (no RNG state)"""
    s = np.sin(p[..., 0] * 12.9898 + p[..., 1] * 78.233 + p[..., 2] * 37.719 + seed * 1.618) * 43758.5453
    return s - np.floor(s)


def _finite_normals(fn, eps=1e-3):
    def wrapped(u, v):
        P = fn(u, v)
        Pu = fn(np.clip(u + eps, 0, 1), v) - fn(np.clip(u - eps, 0, 1), v)
        Pv = fn(u, np.clip(v + eps, 0, 1)) - fn(u, np.clip(v - eps, 0, 1))
        N = _normalize(np.cross(Pu, Pv))
        N = np.where(np.isfinite(N), N, np.array([0.0, 1.0, 0.0]))
        return P, N
    return wrapped


class _MeshBuilder:
    def __init__(self):
        self.attrs, self.idx, self.groups, self.nv, self.ni = [], [], [], 0, 0

    def add(self, attrs, idx, mat, back=None):
        self.attrs.append(attrs)
        self.idx.append(idx + self.nv)
        self.groups.append((mat, back, self.ni, len(idx)))
        self.nv += len(attrs)
        self.ni += len(idx)

    def finish(self):
        return np.concatenate(self.attrs), np.concatenate(self.idx), self.groups


def _procedural_texture(res: int, seed: int, kind: str) -> np.ndarray:
    """tileable RGBA8 pattern from sines and value noise: "albedo" (mottled, mid-grey mean so that a base colour still tints
    it), "rough" (one channel repeated) or "normal" (tangent-space bumps)"""
    y, x = np.meshgrid(np.arange(res) / res, np.arange(res) / res, indexing="ij")
    two_pi = 2.0 * math.pi
    f = np.zeros((res, res))
    for k, (fx, fy) in enumerate(((3, 5), (7, 2), (13, 11), (29, 17), (61, 47))):
        f += np.sin(two_pi * (fx * x + fy * y) + seed * (k + 1.3)) * np.cos(two_pi * (fy * x - fx * y) + seed * 0.7 * k) / (k + 1.5)
    f = (f - f.min()) / (f.max() - f.min())
    grain = _noise(np.stack([x * res, y * res, x * 0 + seed], -1), seed)
    if kind == "albedo":
        v = 0.55 + 0.35 * (f - 0.5) + 0.12 * (grain - 0.5)
        rgb = np.stack([v * (1.0 + 0.08 * np.sin(seed)), v, v * (1.0 - 0.08 * np.cos(seed))], -1)
    elif kind == "rough":
        v = 0.35 + 0.5 * f + 0.1 * (grain - 0.5)
        rgb = np.stack([v, v, v], -1)
    else:
        gy, gx = np.gradient(f + 0.05 * grain)
        nrm = _normalize(np.stack([-gx * res * 0.02, -gy * res * 0.02, np.ones_like(f)], -1))
        rgb = nrm * 0.5 + 0.5
    out = np.empty((res, res, 4), dtype=np.uint8)
    out[..., :3] = np.clip(rgb * 255.0 + 0.5, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def atrium(scene, detail: float = 1.0, cam_overrides=None, textured: bool = False, tex_res: int = 1024, compress: bool = False,
           one_material_type: bool = False, spatial_splits: bool = False, fast_bvh_build: bool = False):
    """Synthetic atrium, 30 x 12 x 18 (SURVEY.md section 8d input 3/4).  Triangle count ~ 250k * detail.
    textured=True: every large surface gets its own mip-mapped base-colour map, the stone also a normal map, the floor a
    roughness map (11 maps of tex_res^2: the material -> texture gathers a textured asset set causes in the shade stage).
    spatial_splits / fast_bvh_build: mesh_desc_t::allow_spatial_splits / use_fast_bvh_build (SceneBase.h:130-131) of the one mesh."""
    d = max(detail, 0.02)
    s = math.sqrt(d)
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    tex = {}
    if textured:
        for k, (name, kind, srgb, nm) in enumerate((("stone", "albedo", True, False), ("stone_n", "normal", False, True),
                                                     ("floor", "albedo", True, False), ("floor_r", "rough", False, False),
                                                     ("cloth_r", "albedo", True, False), ("cloth_g", "albedo", True, False),
                                                     ("cloth_b", "albedo", True, False), ("metal", "albedo", True, False),
                                                     ("metal_r", "rough", False, False), ("glossy", "albedo", True, False),
                                                     ("stone_r", "rough", False, False))):
            # compress: leave the choice to settings_t::use_tex_compression (BC3 / BC4 / BC5 storages when the scene has it on)
            # (RGB888 / R8 inputs: with compression on they land in the BC3 (YCoCg) / BC5 / BC4 storages, SceneCPU.cpp:90-160;
            # an RGBA8888 colour map is never compressed)
            img = _procedural_texture(tex_res, 3 + k, kind)
            img, fmt = (img[..., :1], eTextureFormat.R8) if kind == "rough" else (img[..., :3], eTextureFormat.RGB888)
            tex[name] = scene.AddTexture(img, fmt=fmt, is_srgb=srgb, is_normalmap=nm, generate_mipmaps=True, force_no_compression=not compress)
    t = lambda name: tex.get(name)  # noqa: E731
    stone = scene.AddMaterial(PrincipledMat(base_color=(0.62, 0.58, 0.50), roughness=0.7, specular=0.3, base_texture=t("stone"),
                                            normal_map=t("stone_n"), roughness_texture=t("stone_r")))
    floor_m = scene.AddMaterial(PrincipledMat(base_color=(0.35, 0.33, 0.32), roughness=0.35, specular=0.5, base_texture=t("floor"),
                                              roughness_texture=t("floor_r")))
    cloth_r = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.55, 0.08, 0.07), roughness=0.5, base_texture=t("cloth_r")))
    cloth_g = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.10, 0.40, 0.12), roughness=0.5, base_texture=t("cloth_g")))
    cloth_b = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.10, 0.15, 0.50), roughness=0.5, base_texture=t("cloth_b")))
    metal = scene.AddMaterial(PrincipledMat(base_color=(0.90, 0.75, 0.40), metallic=1.0, roughness=0.25, base_texture=t("metal"),
                                            roughness_texture=t("metal_r")))
    glossy = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.8, 0.8, 0.85), roughness=0.15, base_texture=t("glossy")))
    if one_material_type:  # tuning experiment (how much does material-type divergence inside a wavefront cost?): Principled everywhere
        cloth_r = scene.AddMaterial(PrincipledMat(base_color=(0.55, 0.08, 0.07), roughness=0.5, specular=0.0))
        cloth_g = scene.AddMaterial(PrincipledMat(base_color=(0.10, 0.40, 0.12), roughness=0.5, specular=0.0))
        cloth_b = scene.AddMaterial(PrincipledMat(base_color=(0.10, 0.15, 0.50), roughness=0.5, specular=0.0))
        glossy = scene.AddMaterial(PrincipledMat(base_color=(0.8, 0.8, 0.85), metallic=1.0, roughness=0.15))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=18.0, base_color=(1.0, 0.95, 0.85),
                                         importance_sample=True))
    X, Y, Z = 30.0, 12.0, 18.0
    mb = _MeshBuilder()

    def n(k):
        return max(2, int(round(k * s)))

    # displaced floor + walls + ceiling (big patches: top of the tree)
    mb.add(*_grid(n(220), n(140), _finite_normals(lambda u, v: np.stack(
        [u * X - X / 2, 0.04 * np.sin(u * 70) * np.sin(v * 45) + 0.02 * _noise(np.stack([u, v, u * 0], -1), 1), v * Z - Z / 2], -1))),
        floor_m)
    mb.add(*_grid(n(120), n(60), _finite_normals(lambda u, v: np.stack(
        [u * X - X / 2, v * Y, -Z / 2 + 0.08 * np.sin(u * 60) * np.cos(v * 30)], -1))), stone)
    mb.add(*_grid(n(120), n(60), _finite_normals(lambda u, v: np.stack(
        [X / 2 - u * X, v * Y, Z / 2 - 0.08 * np.sin(u * 60) * np.cos(v * 30)], -1))), stone)
    mb.add(*_grid(n(80), n(60), _finite_normals(lambda u, v: np.stack(
        [-X / 2 + 0.08 * np.sin(u * 40) * np.cos(v * 30), v * Y, Z / 2 - u * Z], -1))), stone)
    mb.add(*_grid(n(80), n(60), _finite_normals(lambda u, v: np.stack(
        [X / 2 - 0.08 * np.sin(u * 40) * np.cos(v * 30), v * Y, u * Z - Z / 2], -1))), stone)
    mb.add(*_grid(n(100), n(60), _finite_normals(lambda u, v: np.stack(
        [X / 2 - u * X, Y + 0.3 * np.sin(u * math.pi) * np.sin(v * math.pi), v * Z - Z / 2], -1))), stone)

    # colonnade: two rows of fluted columns + arches between them
    ncol = 10
    for row, zc in enumerate((-Z / 4, Z / 4)):
        for k in range(ncol):
            xc = -X / 2 + (k + 0.5) * X / ncol

            def column(u, v, xc=xc, zc=zc):
                ang = u * 2 * math.pi
                r = 0.45 * (1.0 + 0.06 * np.cos(ang * 20)) * (1.0 - 0.12 * v) + 0.25 * np.exp(-((v - 0.02) * 30) ** 2) \
                    + 0.25 * np.exp(-((v - 0.98) * 30) ** 2)
                return np.stack([xc + r * np.cos(ang), v * 8.0, zc + r * np.sin(ang)], -1)
            mb.add(*_grid(n(96), n(64), _finite_normals(column)), stone)
            if k + 1 < ncol:
                def arch(u, v, xc=xc, zc=zc):
                    ang = u * math.pi
                    half = X / ncol / 2
                    cx = xc + half - half * np.cos(ang)
                    cy = 8.0 + 1.6 * np.sin(ang)
                    return np.stack([cx, cy + 0.5 * v + 0.03 * np.sin(u * 90), zc + (v - 0.5) * 0.9], -1)
                mb.add(*_grid(n(64), n(10), _finite_normals(arch)), stone, back=stone)

    # draped cloth sheets hanging between the rows
    for k, mat in enumerate((cloth_r, cloth_g, cloth_b, cloth_r, cloth_b)):
        x0 = -X / 2 + 3.0 + k * 5.5

        def cloth(u, v, x0=x0, k=k):
            sag = 1.4 * np.sin(u * math.pi) + 0.25 * np.sin(v * 14 + k) * np.sin(u * 9)
            return np.stack([x0 + 0.35 * np.sin(v * 11 + u * 5 + k), 9.5 - sag - 0.1 * v, -Z / 4 + u * Z / 2], -1)
        mb.add(*_grid(n(150), n(110), _finite_normals(cloth)), mat, back=mat)

    # clutter: noisy spheres and tori on the floor
    for k in range(14):
        cx = -X / 2 + 2.0 + (k * 7.3) % (X - 4.0)
        cz = -Z / 2 + 2.0 + (k * 4.1) % (Z - 4.0)
        rad = 0.5 + 0.35 * ((k * 37) % 7) / 7.0
        if k % 2 == 0:
            def blob(u, v, cx=cx, cz=cz, rad=rad, k=k):
                th, ph = v * math.pi, u * 2 * math.pi
                dirv = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1)
                r = rad * (1.0 + 0.12 * np.sin(6 * ph + k) * np.sin(5 * th))
                return np.array([cx, rad, cz]) + dirv * r[..., None]
            mb.add(*_grid(n(90), n(60), _finite_normals(blob)), metal if k % 4 == 0 else glossy)
        else:
            def torus(u, v, cx=cx, cz=cz, rad=rad):
                a, b = u * 2 * math.pi, v * 2 * math.pi
                R, r = rad, 0.35 * rad
                return np.stack([cx + (R + r * np.cos(b)) * np.cos(a), r + r * np.sin(b) + 0.02,
                                 cz + (R + r * np.cos(b)) * np.sin(a)], -1)
            mb.add(*_grid(n(100), n(48), _finite_normals(torus)), metal if k % 3 == 0 else stone)

    # emissive ceiling strips (TRI lights through MAT_FLAG_IMP_SAMPLE)
    for k in range(4):
        x0 = -X / 2 + 4.0 + k * 7.0

        def strip(u, v, x0=x0):
            return np.stack([x0 + u * 2.5, 0 * u + Y - 0.45, -Z / 2 + 2.0 + v * (Z - 4.0)], -1)
        a, i = _grid(2, 6, lambda u, v, f=strip: (f(u, v), np.broadcast_to(np.array([0.0, -1.0, 0.0]), (*u.shape, 3))), flip=True)
        mb.add(a, i, emit, back=0xFFFFFFFF)

    attrs, idx, groups = mb.finish()
    mesh = scene.AddMesh(attrs, idx, groups, allow_spatial_splits=spatial_splits, use_fast_bvh_build=fast_bvh_build)
    scene.AddMeshInstance(mesh)
    kw = dict(type=0, origin=(-X / 2 + 2.5, 2.2, 0.6), fwd=_unit((1.0, 0.12, -0.08)), fov=60.0)
    kw.update(cam_overrides or {})
    cam = scene.AddCamera(**kw)
    scene.set_current_cam(cam)
    scene.Finalize()
    return int(len(idx) // 3)


# ---- the Bistro-class street on the reference's own asset mesh (SURVEY.md section 8d input 4) --------
ASSET_MESHES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "assets", "_ref", "meshes.npz")


def have_asset_meshes() -> bool:
    return os.path.exists(ASSET_MESHES)


def load_asset_mesh(stem: str):
    """the reference's tests/test_data/meshes/mat_test/<stem>.bin as tests/golden/stage_ref_assets.py staged it (a data file: interleaved
    position3 / normal3 / uv2, indices; loader layout tests/utils.cpp:72-114): attrs [n, 8] f32, indices u32"""
    if not have_asset_meshes():
        raise RuntimeError(f"{ASSET_MESHES} is not staged (tests/golden/stage_ref_assets.py, run by __graft_entry__.build() where the reference tree exists)")
    m = np.load(ASSET_MESHES)
    return m[stem + ".attrs"].reshape(-1, 8).astype(np.float32), m[stem + ".indices"].astype(np.uint32)


def _rot_y(deg):
    a = math.radians(deg)
    return np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])


def _rot_x(deg):
    a = math.radians(deg)
    return np.array([[1.0, 0.0, 0.0], [0.0, math.cos(a), -math.sin(a)], [0.0, math.sin(a), math.cos(a)]])


def street_assets(scene, copies: int = 38, instanced: bool = False, spatial_splits: bool = False, fast_bvh_build: bool = False,
                  cam_overrides=None):
    """`copies` jittered / rotated / scaled copies of mat_test/model.bin (77 762 triangles of a real asset mesh each: a material ball with
    long thin triangles, creases and nested shells -- not a displaced grid) standing along a street between two facades, on a ground
    plane, under ~200 emissive triangles (lamps).  copies = 38 is the ~3 M-triangle scene BASELINE.md 4.3 (4) specifies.
    instanced = False: the copies are BAKED into one mesh (one bottom-level tree over 3 M triangles: the working set the survey asks for);
    instanced = True:  the copies are mesh INSTANCES (four meshes -- one per material -- of 77 762 triangles, a top-level tree over
                       `copies` + 3 instances: the two-level walk, transforms and all, is on the clock).
    Deterministic (no RNG state): the jitter comes from the copy's index."""
    attrs0, idx0 = load_asset_mesh("model")
    lo, hi = attrs0[:, :3].min(0), attrs0[:, :3].max(0)
    centre_xz = np.array([(lo[0] + hi[0]) / 2, lo[1], (lo[2] + hi[2]) / 2])
    unit = 1.0 / float((hi - lo).max())  # the ball is ~0.11 units across: bring it to 1, then scale per copy
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    mats = [scene.AddMaterial(PrincipledMat(base_color=(0.90, 0.75, 0.40), metallic=1.0, roughness=0.25)),
            scene.AddMaterial(PrincipledMat(base_color=(0.55, 0.12, 0.10), roughness=0.45, specular=0.5)),
            scene.AddMaterial(PrincipledMat(base_color=(0.15, 0.30, 0.55), roughness=0.6, specular=0.3, clearcoat=0.5)),
            scene.AddMaterial(PrincipledMat(base_color=(0.70, 0.70, 0.72), roughness=0.8, specular=0.2))]
    stone = scene.AddMaterial(PrincipledMat(base_color=(0.55, 0.52, 0.47), roughness=0.75, specular=0.3))
    asphalt = scene.AddMaterial(PrincipledMat(base_color=(0.20, 0.20, 0.21), roughness=0.5, specular=0.4))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=25.0, base_color=(1.0, 0.93, 0.80), importance_sample=True))
    L, W, H = 64.0, 14.0, 16.0  # street along x; facades at z = +-W/2

    def placement(k):
        """scale, rotation matrix, translation of copy k: two staggered rows along the street, sizes 1.6 .. 3.4 m, any heading, a slight lean"""
        row, col = k % 2, k // 2
        n_col = (copies + 1) // 2
        x = -L / 2 + 3.0 + (col + 0.5 * row) * (L - 6.0) / max(n_col, 1)
        z = (-1.0 if row == 0 else 1.0) * (2.2 + 1.3 * _noise(np.array([k * 0.37, 0.1, 0.7]), 3))
        scale = 1.6 + 1.8 * float(_noise(np.array([k * 0.91, 0.5, 0.2]), 5))
        R = _rot_y(360.0 * float(_noise(np.array([k * 0.53, 0.9, 0.4]), 7))) @ _rot_x(10.0 * (float(_noise(np.array([k * 0.77, 0.3, 0.6]), 11)) - 0.5))
        return scale * unit, R, np.array([x, 0.0, float(z)])

    env = _MeshBuilder()
    env.add(*_grid(200, 48, _finite_normals(lambda u, v: np.stack(
        [u * L - L / 2, 0.03 * np.sin(u * 160) * np.sin(v * 31) + 0.015 * _noise(np.stack([u, v, u * 0], -1), 1), v * W - W / 2], -1))), asphalt)
    for side in (-1.0, 1.0):
        def facade(u, v, side=side):
            # window bays and cornices: a relief of boxes smoothed by sines (long, thin triangles where the relief is steep)
            bay = 0.35 * (np.sin(u * 2 * math.pi * 16) > 0.55) * (np.sin(v * 2 * math.pi * 5) > 0.2)
            relief = 0.12 * np.sin(u * 130) * np.cos(v * 47) + bay
            x = u * L - L / 2 if side < 0 else L / 2 - u * L
            return np.stack([x, v * H, side * (W / 2 + relief)], -1)
        env.add(*_grid(260, 64, _finite_normals(facade)), stone)
    for end in (-1.0, 1.0):  # the street's ends are closed by plain walls
        def wall(u, v, end=end):
            z = u * W - W / 2 if end > 0 else W / 2 - u * W
            return np.stack([end * L / 2 + 0 * u, v * H, z], -1)
        env.add(*_grid(24, 24, _finite_normals(wall)), stone)
    lamps = _MeshBuilder()
    n_lamps = 25
    for k in range(n_lamps):  # 25 lamps x (2 x 2 quads x 2 triangles) = 200 emissive triangles, hanging over the middle of the street
        x0 = -L / 2 + 2.0 + k * (L - 4.0) / (n_lamps - 1)
        z0 = 1.5 * math.sin(k * 1.7)

        def lamp(u, v, x0=x0, z0=z0):
            return np.stack([x0 + (u - 0.5) * 0.9, 0 * u + 7.5 + 0.4 * math.sin(k * 0.9), z0 + (v - 0.5) * 0.5], -1)
        a, i = _grid(2, 2, lambda u, v, f=lamp: (f(u, v), np.broadcast_to(np.array([0.0, -1.0, 0.0]), (*u.shape, 3))), flip=True)
        lamps.add(a, i, emit, back=0xFFFFFFFF)

    n_tris = 0
    if instanced:
        P0 = (attrs0[:, :3] - centre_xz).astype(np.float32)
        base = np.concatenate([P0, attrs0[:, 3:]], axis=-1)
        meshes = [scene.AddMesh(base, idx0, [(m, None, 0, len(idx0))], allow_spatial_splits=spatial_splits, use_fast_bvh_build=fast_bvh_build)
                  for m in mats]
        for k in range(copies):
            sc_k, R, t = placement(k)
            M = np.eye(4)
            M[:3, :3] = R * sc_k
            M[:3, 3] = t
            scene.AddMeshInstance(meshes[k % len(mats)], M.T.astype(np.float32))  # (column-major, as the reference takes it)
        n_tris += len(mats) * len(idx0) // 3
        a, i, g = env.finish()
        scene.AddMeshInstance(scene.AddMesh(a, i, g, allow_spatial_splits=spatial_splits, use_fast_bvh_build=fast_bvh_build))
        n_tris += len(i) // 3
        a, i, g = lamps.finish()
        scene.AddMeshInstance(scene.AddMesh(a, i, g))
        n_tris += len(i) // 3
    else:
        mb = _MeshBuilder()
        for k in range(copies):
            sc_k, R, t = placement(k)
            P = (attrs0[:, :3] - centre_xz) @ (R * sc_k).T + t
            N = attrs0[:, 3:6] @ R.T  # (uniform scale: normals rotate)
            mb.add(np.concatenate([P, N, attrs0[:, 6:8]], axis=-1).astype(np.float32), idx0, mats[k % len(mats)])
        for src in (env, lamps):
            for a, i, (front, back, _, _) in zip(src.attrs, src.idx, src.groups):
                mb.add(a, i - _group_base(src, a), front, back=back)
        a, i, g = mb.finish()
        scene.AddMeshInstance(scene.AddMesh(a, i, g, allow_spatial_splits=spatial_splits, use_fast_bvh_build=fast_bvh_build))
        n_tris += len(i) // 3
    kw = dict(type=0, origin=(-L / 2 + 3.0, 1.7, 0.4), fwd=_unit((1.0, 0.06, -0.03)), fov=60.0)
    kw.update(cam_overrides or {})
    cam = scene.AddCamera(**kw)
    scene.set_current_cam(cam)
    scene.Finalize()
    return int(n_tris)


def _group_base(builder, attrs):
    """first vertex a part of `builder` was given (its indices were rebased by that much in _MeshBuilder.add)"""
    nv = 0
    for a in builder.attrs:
        if a is attrs:
            return nv
        nv += len(a)
    raise KeyError


def cornell_needles(scene, spatial_splits: bool = False, fast_bvh_build: bool = False, seed: int = 5, **cam_overrides):
    """The Cornell room (big wall triangles) with ~300 long thin slivers strung diagonally through it and a carpet of small triangles on the
    floor, all ONE mesh: the case mesh_desc_t::allow_spatial_splits exists for (BVHSplit.cpp:323-470 -- a sliver's box covers half the room,
    so the builder duplicates its reference into several leaves), which is exactly what the leaf refinement, the layout permutation and
    the 4-wide collapse on the upload path must survive.  use_fast_bvh_build selects the reference's binned builder."""
    rs = np.random.RandomState(9100 + seed)
    scene.SetEnvironment(env_col=(0.0, 0.0, 0.0))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    red = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.0, 0.0)))
    green = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.0, 0.5, 0.0)))
    lamp = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=100.0, importance_sample=True))
    gloss = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.8, 0.7, 0.3), roughness=0.2))
    attrs, idx = cornell_mesh_arrays(_CORNELL_QUADS)
    attrs, idx = [attrs], [idx]
    n_room = int(attrs[0].shape[0])
    groups = [(grey, None, 0, 18), (red, None, 19, 6), (green, None, 25, 6), (lamp, 0xFFFFFFFF, 31, 6)]
    base = n_room
    n_idx = 37  # (the room's index list: 6 quads x 6 + the sample's off-by-one start of every group, kept as the reference sample has it)
    n_idx = int(idx[0].size)

    def add_tris(tri_pts, mat):
        nonlocal base, n_idx
        tri_pts = np.asarray(tri_pts, dtype=np.float32).reshape(-1, 3, 3)
        nrm = np.cross(tri_pts[:, 1] - tri_pts[:, 0], tri_pts[:, 2] - tri_pts[:, 0])
        nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-12)
        a = np.zeros((tri_pts.shape[0] * 3, 8), dtype=np.float32)
        a[:, 0:3] = tri_pts.reshape(-1, 3)
        a[:, 3:6] = np.repeat(nrm, 3, axis=0)
        attrs.append(a)
        idx.append(np.arange(base, base + a.shape[0], dtype=np.uint32))
        groups.append((mat, mat, n_idx, a.shape[0]))
        base += a.shape[0]
        n_idx += a.shape[0]

    # slivers: 0.5 long, 0.004 wide -- half of them axis-aligned planks over the carpet (their boxes overlap hundreds of small triangles' boxes:
    # what the builder's spatial split is for), half random diagonals through the room
    lo, hi = np.array([-0.54, 0.02, -0.54]), np.array([-0.02, 0.52, -0.02])
    p0 = lo + rs.uniform(size=(300, 3)) * (hi - lo)
    p1 = lo + rs.uniform(size=(300, 3)) * (hi - lo)
    planks = np.arange(300) < 150
    along_x = planks & (np.arange(300) % 2 == 0)
    along_z = planks & ~along_x
    p0[planks, 1] = p1[planks, 1] = 0.012 + 0.0004 * np.arange(150)
    p0[along_x, 0], p1[along_x, 0], p1[along_x, 2] = -0.535, -0.025, p0[along_x, 2]
    p0[along_z, 2], p1[along_z, 2], p1[along_z, 0] = -0.535, -0.025, p0[along_z, 0]
    side = np.cross(p1 - p0, np.where(planks[:, None], np.array([0.0, 1.0, 0.0]), rs.normal(size=(300, 3))))
    side *= 0.002 / np.maximum(np.linalg.norm(side, axis=-1, keepdims=True), 1e-9)
    add_tris(np.stack([p0 - side, p0 + side, p1], axis=1), gloss)
    # carpet: a 40 x 40 grid of small triangles just above the floor
    g = np.linspace(-0.53, -0.03, 41)
    x0, z0 = np.meshgrid(g[:-1], g[:-1], indexing="ij")
    d = g[1] - g[0]
    y = 0.004 + 0.003 * np.sin(x0 * 60.0) * np.cos(z0 * 45.0)
    a = np.stack([x0, y, z0], -1).reshape(-1, 3)
    b = a + np.array([d, 0.0, 0.0])
    c = a + np.array([0.0, 0.0, d])
    add_tris(np.stack([a, c, b], axis=1), grey)
    mesh = scene.AddMesh(np.concatenate(attrs), np.concatenate(idx), groups, allow_spatial_splits=spatial_splits, use_fast_bvh_build=fast_bvh_build)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return tuple(float(x) for x in v / np.linalg.norm(v))


def random_cornell(scene, seed: int = 0, **cam_overrides):
    """Cornell box with pseudo-random materials, lights and camera drawn from `seed` (tests: a fuzzer over the corners of
    shading_node_desc_t / principled_mat_desc_t / the light descriptors / camera_desc_t)"""
    rs = np.random.RandomState(seed)
    u = lambda lo=0.0, hi=1.0: float(rs.uniform(lo, hi))  # noqa: E731
    col = lambda lo=0.05, hi=0.95: (u(lo, hi), u(lo, hi), u(lo, hi))  # noqa: E731
    scene.SetEnvironment(env_col=col(0.0, 0.08), back_col=col(0.0, 0.05))
    tex = scene.AddTexture(checkerboard(64, int(rs.choice([4, 8, 16]))), generate_mipmaps=bool(rs.randint(2)))
    nmap = scene.AddTexture(bump_normal_map(32), is_srgb=False, is_normalmap=True)

    def leaf_material():
        kind = rs.randint(4)
        if kind == 0:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=col(), roughness=u(),
                                                 base_texture=tex if rs.randint(3) == 0 else None))
        if kind == 1:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=col(), roughness=u(0.02, 0.9), anisotropic=u(0.0, 0.9),
                                                 anisotropic_rotation=u(), normal_map=nmap if rs.randint(2) else None,
                                                 normal_map_intensity=u(0.2, 1.0)))
        if kind == 2:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Refractive, base_color=col(0.6, 1.0), roughness=u(0.0, 0.5), ior=u(1.05, 2.0)))
        return scene.AddMaterial(PrincipledMat(
            base_color=col(), base_texture=tex if rs.randint(3) == 0 else None, metallic=float(rs.choice([0.0, u(), 1.0])), specular=u(),
            specular_tint=u(), roughness=u(0.02, 1.0), anisotropic=u(0.0, 0.9), anisotropic_rotation=u(), sheen=u(0.0, 1.0) * rs.randint(2),
            sheen_tint=u(), clearcoat=u() * rs.randint(2), clearcoat_roughness=u(0.0, 0.5), ior=u(1.1, 1.9),
            transmission=float(rs.choice([0.0, 0.0, u(), 1.0])), transmission_roughness=u(0.0, 0.6),
            emission_color=col(), emission_strength=u(0.0, 0.8) * rs.randint(2), alpha=float(rs.choice([1.0, 1.0, u(0.3, 1.0)])),
            normal_map=nmap if rs.randint(3) == 0 else None, normal_map_intensity=u(0.2, 1.0)))

    def material():
        if rs.randint(5) == 0:
            a, b = leaf_material(), leaf_material()
            if rs.randint(2):
                b = scene.AddMaterial(ShadingNode(type=eShadingNode.Transparent, base_color=col(0.5, 1.0)))
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Mix, mix_materials=(a, b), strength=u(0.1, 0.9), mix_add=bool(rs.randint(2)),
                                                 base_texture=tex if rs.randint(3) == 0 else None))
        return leaf_material()

    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=u(20.0, 100.0), base_color=col(0.6, 1.0),
                                         importance_sample=bool(rs.randint(4))))
    attrs, idx = cornell_mesh_arrays()
    mats = [material() for _ in range(7)]
    groups = [(mats[0], None, 0, 6), (mats[1], None, 6, 6), (mats[2], None, 12, 6), (mats[3], None, 19, 6), (mats[4], None, 25, 6),
              (emit, 0xFFFFFFFF, 31, 6), (mats[5], mats[5], 37, 30), (mats[6], mats[6], 67, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    flags = lambda: dict(multiple_importance=bool(rs.randint(2)), cast_shadow=bool(rs.randint(4)),  # noqa: E731
                         diffuse_visibility=bool(rs.randint(4)), specular_visibility=bool(rs.randint(4)),
                         refraction_visibility=bool(rs.randint(4)))
    for _ in range(rs.randint(1, 5)):
        kind = rs.choice(["sphere", "spot", "rect", "disk", "line", "directional"])
        pos = (u(-0.5, -0.05), u(0.15, 0.5), u(-0.5, -0.05))
        if kind == "sphere":
            scene.AddLight("sphere", color=col(0.5, 8.0), position=pos, radius=float(rs.choice([0.0, u(0.005, 0.05)])), **flags())
        elif kind == "spot":
            scene.AddLight("spot", color=col(2.0, 20.0), position=pos, direction=(u(-0.5, 0.5), -1.0, u(-0.5, 0.5)),
                           radius=float(rs.choice([0.0, u(0.005, 0.03)])), spot_size=u(20.0, 90.0), spot_blend=u(0.0, 1.0), **flags())
        elif kind == "rect":
            scene.AddLight("rect", color=col(1.0, 10.0), width=u(0.05, 0.2), height=u(0.05, 0.2), doublesided=bool(rs.randint(2)),
                           xform=_translate(*pos, rot_x_deg=u(-40.0, 40.0), rot_z_deg=u(-40.0, 40.0)), **flags())
        elif kind == "disk":
            scene.AddLight("disk", color=col(1.0, 10.0), width=u(0.05, 0.2), height=u(0.05, 0.2), doublesided=bool(rs.randint(2)),
                           xform=_translate(*pos, rot_x_deg=u(-40.0, 40.0), rot_z_deg=u(-40.0, 40.0)), **flags())
        elif kind == "line":
            scene.AddLight("line", color=col(1.0, 10.0), radius=u(0.002, 0.01), height=u(0.1, 0.3),
                           xform=_translate(*pos, rot_x_deg=u(0.0, 180.0), rot_z_deg=u(0.0, 180.0)), **flags())
        else:
            scene.AddLight("directional", color=col(0.2, 2.0), direction=(u(-0.5, 0.5), -1.0, u(-1.0, 0.0)),
                           angle=float(rs.choice([0.0, u(0.5, 8.0)])), **flags())
    kw = dict(fov=u(30.0, 60.0), filter=int(rs.randint(3)), filter_width=u(1.0, 2.5), exposure=u(-1.0, 1.0), gamma=float(rs.choice([1.0, 2.2])),
              max_diff_depth=int(rs.randint(1, 5)), max_spec_depth=int(rs.randint(1, 7)), max_refr_depth=int(rs.randint(1, 7)),
              max_transp_depth=int(rs.randint(1, 7)), max_total_depth=int(rs.randint(2, 8)), min_total_depth=int(rs.randint(1, 4)),
              min_transp_depth=int(rs.randint(1, 4)), clamp_direct=float(rs.choice([0.0, u(1.0, 10.0)])),
              clamp_indirect=float(rs.choice([0.0, u(1.0, 10.0)])), regularize_alpha=float(rs.choice([0.0, u(0.01, 0.1)])))
    if rs.randint(3) == 0:
        kw.update(fstop=u(1.0, 8.0), focus_distance=u(0.4, 1.0), focal_length=u(0.02, 0.08), lens_blades=int(rs.choice([0, 5, 7])),
                  lens_rotation=u(0.0, 1.0), lens_ratio=u(0.7, 1.5))
    kw.update(cam_overrides)
    _cornell_camera(scene, **kw)
    scene.Finalize()


def random_instances(scene, seed: int = 0, **cam_overrides):
    """fuzzer over the two-level hierarchy and the environment: an open or closed room, 2-9 instances of two block meshes
    with random rotations / non-uniform scales / translations / ray-type visibility, random leaf materials, an HDR
    environment (importance-sampled or not, rotated) or a constant one, optional sky portals"""
    rs = np.random.RandomState(10000 + seed)
    u = lambda lo=0.0, hi=1.0: float(rs.uniform(lo, hi))  # noqa: E731
    col = lambda lo=0.05, hi=0.95: (u(lo, hi), u(lo, hi), u(lo, hi))  # noqa: E731
    open_top = bool(rs.randint(2))
    if rs.randint(3):
        sky = scene.AddTexture(rgbe_sky(), is_srgb=False)
        rot = u(0.0, 6.0)
        scene.SetEnvironment(env_col=col(0.3, 1.5), back_col=col(0.3, 1.5), env_map=sky, back_map=sky if rs.randint(2) else InvalidHandle,
                             env_map_rotation=rot, back_map_rotation=rot, importance_sample=bool(rs.randint(2)))
    else:
        scene.SetEnvironment(env_col=col(0.0, 0.6), back_col=col(0.0, 0.3))
    tex = scene.AddTexture(checkerboard(32, 4), generate_mipmaps=True)

    def material():
        kind = rs.randint(5)
        if kind == 0:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=col(), roughness=u(), base_texture=tex if rs.randint(3) == 0 else None))
        if kind == 1:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=col(), roughness=u(0.02, 0.8)))
        if kind == 2:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Refractive, base_color=col(0.7, 1.0), roughness=u(0.0, 0.3), ior=u(1.1, 1.8)))
        if kind == 3:
            a = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=col()))
            b = scene.AddMaterial(ShadingNode(type=eShadingNode.Transparent, base_color=col(0.5, 1.0)))
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Mix, mix_materials=(a, b), strength=u(0.2, 0.8)))
        return scene.AddMaterial(PrincipledMat(base_color=col(), metallic=float(rs.randint(2)), roughness=u(0.05, 0.9), specular=u(),
                                               transmission=float(rs.choice([0.0, 0.0, 1.0])), ior=u(1.1, 1.8), clearcoat=u() * rs.randint(2)))

    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=u(20.0, 80.0), importance_sample=True))
    q = _CORNELL_QUADS
    if open_top:
        attrs, idx = cornell_mesh_arrays([q[0], q[2], q[3], q[4]])  # floor, back, left, right
        room = scene.AddMesh(attrs, idx, [(material(), None, 0, 6), (grey, None, 6, 6), (material(), None, 12, 12)])
    else:
        attrs, idx = cornell_mesh_arrays(q)
        room = scene.AddMesh(attrs, idx, [(material(), None, 0, 6), (grey, None, 6, 12), (material(), None, 18, 12), (emit, 0xFFFFFFFF, 30, 6)])
    scene.AddMeshInstance(room)
    meshes = []
    for kind in ("short", "tall"):
        attrs, idx = cornell_mesh_arrays(_block_quads(kind))
        for _ in range(2):
            m = material()
            meshes.append(scene.AddMesh(attrs, idx, [(m, m, 0, 30)]))
    for _ in range(rs.randint(2, 10)):
        xf = _xform(translate=(u(-0.25, 0.2), u(0.0, 0.3), u(-0.15, 0.25)), rot_y_deg=u(0.0, 360.0), rot_z_deg=u(-30.0, 30.0) * rs.randint(2),
                    scale=(u(0.3, 1.0), u(0.3, 1.4), u(0.3, 1.0)))
        vis = dict(camera=bool(rs.randint(5)), diffuse=bool(rs.randint(5)), specular=bool(rs.randint(5)), refraction=bool(rs.randint(5)),
                   shadow=bool(rs.randint(5)))
        scene.AddMeshInstance(int(rs.choice(meshes)), xf, **vis)
    if open_top and rs.randint(2):
        scene.AddLight("rect", color=(1.0, 1.0, 1.0), width=0.5, height=0.5, sky_portal=True, xform=_translate(-0.28, 0.5488, -0.28))
    if rs.randint(2):
        scene.AddLight("sphere", color=col(1.0, 6.0), position=(u(-0.5, -0.05), u(0.3, 0.5), u(-0.5, -0.05)), radius=u(0.0, 0.04))
    kw = dict(fov=u(35.0, 55.0), max_total_depth=int(rs.randint(2, 8)), max_transp_depth=int(rs.randint(1, 8)))
    kw.update(cam_overrides)
    _cornell_camera(scene, **kw)
    scene.Finalize()


def random_textures(scene, seed: int = 0, **cam_overrides):
    """fuzzer over the texture code: base-colour / roughness / metallic / specular / emission / alpha / normal maps of random
    sizes (incl. non-power-of-two and 1 pixel wide), channel formats, sRGB flags and mip chains, none of them with
    force_no_compression -- so under settings_t::use_tex_compression they take the block-compressed storages"""
    rs = np.random.RandomState(20000 + seed)
    u = lambda lo=0.0, hi=1.0: float(rs.uniform(lo, hi))  # noqa: E731
    scene.SetEnvironment(env_col=(0.03, 0.03, 0.04))

    def image(channels):
        h_, w_ = int(rs.choice([1, 3, 8, 20, 32, 48, 64])), int(rs.choice([1, 5, 8, 24, 32, 64]))
        i, j = np.meshgrid(np.arange(h_), np.arange(w_), indexing="ij")
        planes = [(rs.randint(20, 120) + rs.randint(40, 130) * (0.5 + 0.5 * np.sin(i * u(0.1, 0.9) + j * u(0.1, 0.9) + u(0, 6)))) for _ in range(channels)]
        img = np.stack(planes, axis=-1)
        if rs.randint(2):
            img += 40.0 * (((i // max(1, h_ // 4)) + (j // max(1, w_ // 4))) % 2)[..., None]
        img = np.clip(img, 0, 255).astype(np.uint8)
        if channels == 4:
            img[..., 3] = 255
        return img

    def texture(kind):
        mips = bool(rs.randint(2))
        if kind == "rgb":
            if rs.randint(2):
                return scene.AddTexture(image(3), fmt=eTextureFormat.RGB888, is_srgb=bool(rs.randint(2)), generate_mipmaps=mips, force_no_compression=False)
            return scene.AddTexture(image(4), fmt=eTextureFormat.RGBA8888, is_srgb=bool(rs.randint(2)), generate_mipmaps=mips, force_no_compression=False)
        if kind == "r":
            return scene.AddTexture(image(1), fmt=eTextureFormat.R8, is_srgb=bool(rs.randint(2)), generate_mipmaps=mips, force_no_compression=False)
        res = int(rs.choice([16, 32, 64]))
        return scene.AddTexture(bump_normal_map(res), is_srgb=False, is_normalmap=True, generate_mipmaps=mips, force_no_compression=False)

    def material():
        if rs.randint(4) == 0:
            return scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_texture=texture("rgb"), roughness=u(),
                                                 normal_map=texture("n") if rs.randint(2) else None, normal_map_intensity=u(0.3, 1.0)))
        return scene.AddMaterial(PrincipledMat(
            base_texture=texture("rgb") if rs.randint(4) else None, base_color=(u(0.2, 0.9), u(0.2, 0.9), u(0.2, 0.9)), roughness=u(0.1, 1.0),
            roughness_texture=texture("r") if rs.randint(2) else None, metallic=u(), metallic_texture=texture("r") if rs.randint(3) == 0 else None,
            specular=u(), specular_texture=texture("r") if rs.randint(3) == 0 else None,
            emission_color=(1.0, 0.9, 0.8), emission_strength=u(0.0, 0.6) * rs.randint(2), emission_texture=texture("rgb") if rs.randint(3) == 0 else None,
            alpha=float(rs.choice([1.0, 1.0, u(0.4, 1.0)])), alpha_texture=texture("r") if rs.randint(4) == 0 else None,
            normal_map=texture("n") if rs.randint(3) == 0 else None, normal_map_intensity=u(0.3, 1.0)))

    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=60.0, base_texture=texture("rgb") if rs.randint(2) else None,
                                         importance_sample=True))
    attrs, idx = cornell_mesh_arrays()
    mats = [material() for _ in range(7)]
    groups = [(mats[0], None, 0, 6), (mats[1], None, 6, 6), (mats[2], None, 12, 6), (mats[3], None, 19, 6), (mats[4], None, 25, 6),
              (emit, 0xFFFFFFFF, 31, 6), (mats[5], mats[5], 37, 30), (mats[6], mats[6], 67, 30)]
    mesh = scene.AddMesh(attrs, idx, groups)
    scene.AddMeshInstance(mesh)
    _cornell_camera(scene, **cam_overrides)
    scene.Finalize()


def atrium_small(scene, **cam_overrides):
    """the bench atrium at 1/20 of the Sponza-class detail (tests)"""
    atrium(scene, 0.05, cam_overrides or None)


SCENES = {
    "cornell_basic": cornell_basic,
    "cornell_principled": cornell_principled,
    "cornell_lights": cornell_lights,
    "cornell_env": cornell_env,
    "cornell_filmic": cornell_filmic,
    "cornell_instances": cornell_instances,
}


# scenes whose golden fixtures are frames only (tests/golden/make_fixtures.py)
FRAME_SCENES = {
    "cornell_sky": cornell_sky,
}


def instance_field(scene, count: int = 1000, seed: int = 0, moved: float = 0.0, **cam_overrides):
    """`count` instances of two block meshes (one of them emissive: every instance brings its triangle lights) scattered over a
    floor: the dynamic-scene case of SURVEY.md section 8f (N1).  The instance handles are kept on the scene object;
    move_instance_field() gives every one of them a new transform."""
    rs = np.random.RandomState(77000 + seed)
    scene.SetEnvironment(env_col=(0.25, 0.3, 0.4), back_col=(0.25, 0.3, 0.4))
    grey = scene.AddMaterial(ShadingNode(type=eShadingNode.Diffuse, base_color=(0.55, 0.55, 0.5)))
    blue = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.2, 0.3, 0.7), roughness=0.3))
    emit = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, base_color=(1.0, 0.7, 0.4), strength=6.0, importance_sample=True))
    attrs, idx = cornell_mesh_arrays([_CORNELL_QUADS[0]])
    floor = scene.AddMesh(attrs, idx, [(grey, None, 0, 6)])
    scene.AddMeshInstance(floor, _xform(translate=(0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0)))
    attrs, idx = cornell_mesh_arrays(_block_quads("short"))
    block = scene.AddMesh(attrs, idx, [(blue, None, 0, 30)])
    lamp = scene.AddMesh(attrs, idx, [(emit, None, 0, 30)])
    scene._field = []
    for k in range(count):
        mesh = lamp if k % 64 == 0 else block
        scene._field.append((scene.AddMeshInstance(mesh, _field_xform(rs, 0.0)), k))
    scene.AddLight("sphere", color=(5.0, 5.0, 5.0), position=(-0.28, 0.5, -0.28), radius=0.03)
    kw = dict(origin=(-0.278, 0.55, 0.25), fwd=(0.0, -0.9, -1.0), fov=55.0, max_total_depth=4)
    kw.update(cam_overrides)
    _cornell_camera(scene, **kw)
    scene.Finalize()


def _field_xform(rs, phase: float) -> np.ndarray:
    s = float(rs.uniform(0.02, 0.06))
    return _xform(translate=(float(rs.uniform(-0.55, 0.0)) - 0.1 * s, 0.02 * phase, float(rs.uniform(-0.55, 0.0))),
                  rot_y_deg=float(rs.uniform(0.0, 360.0)) + 40.0 * phase, scale=(s, s * float(rs.uniform(0.5, 3.0)), s))


def move_instance_field(scene, seed: int = 1):
    """a new transform for every instance of instance_field(), then Finalize (the light tree follows the lamps)"""
    rs = np.random.RandomState(78000 + seed)
    for handle, _ in scene._field:
        scene.SetMeshInstanceTransform(handle, _field_xform(rs, 1.0))
    scene.Finalize()
