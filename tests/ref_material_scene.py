"""The reference's material test scene (tests/test_scene.cpp:229-1008, `setup_test_scene`) through the Python mirror of the Ray API.

Test infrastructure.  `build(scene, entry)` sets up what `run_material_test` sets up for one entry of the reference's test matrix
(tests/golden/material_matrix.json, extracted from tests/test_shading.cpp by tests/golden/make_material_matrix.py): the camera of the scene
variant, the fixed grey / emissive / glossy / glass materials, the reference's OWN test meshes (mat_test/*.bin: the 77 762-triangle material ball,
its core, base, text, the glass ball, the light meshes ...), the lights of the variant, the environment, and the material under test with its
textures.  It works on any SceneBase -- the oracle's (RendererRef) and the HIP backend's -- so one call per side gives the two frames to compare.

What cannot be the reference's here, and what stands in for it (every substitution is listed in the dict `build` returns):
  * `env.bin` / `env_floor.bin` (the room around the ball) are absent from the checkout: `_room()` / `_floor_stage()` build a
    room / a floor with blocks of the same material groups;
  * 10 of the 23 texture files are absent: a procedural map of the same role (albedo / roughness / metallic / normal) takes their place;
  * `studio_small_03_2k.hdr` is absent: ray_amd.scenes.rgbe_sky.
The meshes and the textures that ARE in the checkout are staged by tests/golden/stage_ref_assets.py into tests/assets/_ref/ (git-ignored, it
travels to the GPU box like oracle/_ref): /root/reference does not exist there."""
import json
import os

import numpy as np

from ray_amd import api, scenes
from ray_amd.api import PrincipledMat, ShadingNode, eShadingNode, eTextureFormat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, "tests", "assets", "_ref")
MATRIX = os.path.join(ROOT, "tests", "golden", "material_matrix.json")

VIEW_TRANSFORM = {"Standard": 0, "AgX": 1, "Filmic_HighContrast": 8}  # Types.h:70-82


def have_assets() -> bool:
    return os.path.exists(os.path.join(STAGED, "meshes.npz"))


def matrix():
    with open(MATRIX) as f:
        return json.load(f)["tests"]


# ---- staged files (tests/golden/stage_ref_assets.py parsed them) ----------------------------------------------------------------------------
_MESHES = None


def load_bin(name):
    """what tests/utils.cpp:72-114 (LoadBIN) returns for mat_test/<name>: interleaved attributes (8 floats per vertex), indices, group ranges"""
    global _MESHES
    if _MESHES is None:
        _MESHES = np.load(os.path.join(STAGED, "meshes.npz"))
    stem = name[:-4]
    return _MESHES[stem + ".attrs"].reshape(-1, 8), _MESHES[stem + ".indices"], [int(g) for g in _MESHES[stem + ".groups"]]


def load_image(name):
    """a staged .tga in the row order LoadTGA(flip_y = true) hands to AddTexture: [h, w, 3] u8, or None when the checkout does not have the file"""
    p = os.path.join(STAGED, "textures", name + ".npz")
    return np.load(p)["rgb"] if os.path.exists(p) else None


def load_dds(name):
    """what tests/utils.cpp:161-201 (LoadDDS) returns: the blocks behind the header with all their mip levels, w, h, mips, channels; or None"""
    p = os.path.join(STAGED, "textures", name + ".npz")
    if not os.path.exists(p):
        return None
    z = np.load(p)
    w, h, mips, channels = (int(v) for v in z["shape"])
    return z["blocks"], w, h, mips, channels


def stand_in(name, role, res=512):
    """a deterministic map for a texture file the checkout does not have"""
    seed = sum(name.encode()) % 97
    kind = {"base": "albedo", "normal": "normal"}.get(role, "rough")
    return scenes._procedural_texture(res, seed, kind)[..., :3]


# ---- textures of the material under test (tests/test_scene.cpp:36-227, load_needed_textures) ----------------------------------------------
ROLES = {  # field -> (role, uncompressed format, block format, channels of the block format, srgb, normal map, mips generated)
    "base_texture": ("base", eTextureFormat.RGB888, eTextureFormat.BC1, 3, True, False, True),
    "normal_map": ("normal", eTextureFormat.RGB888, eTextureFormat.BC5, 2, False, True, False),
    "roughness_texture": ("rough", eTextureFormat.R8, eTextureFormat.BC4, 1, False, False, True),
    "metallic_texture": ("metal", eTextureFormat.R8, eTextureFormat.BC4, 1, False, False, True),
    "alpha_texture": ("alpha", eTextureFormat.R8, eTextureFormat.BC4, 1, False, False, False),
}


def add_needed_textures(scene, fields, textures, notes):
    out = dict(fields)
    for field, (role, plain, block, block_channels, srgb, is_normal, gen_mips) in ROLES.items():
        if field not in fields:
            continue
        name = textures[fields[field]["texture"]]
        if name.endswith(".dds"):
            dds = load_dds(name)
            if dds is not None:
                data, w, h, mips, channels = dds
                assert channels == block_channels, (name, channels)
                # (the base colour map keeps its mip chain, the others say mips_count = 1 and let the scene take level 0: as the reference's loader does)
                out[field] = scene.AddTexture(data, fmt=block, is_srgb=srgb, is_normalmap=is_normal, generate_mipmaps=gen_mips, size=(w, h),
                                              mips_count=mips if field == "base_texture" else 1, convention_dx=True, force_no_compression=False)
                continue
        img = None if name.endswith(".dds") else load_image(name)
        if img is None:
            notes.append(f"{name}: absent, procedural {role} map")
            img = stand_in(name, role)
        data = img if plain == eTextureFormat.RGB888 else img[..., 0]
        out[field] = scene.AddTexture(np.ascontiguousarray(data), fmt=plain, is_srgb=srgb, is_normalmap=is_normal, generate_mipmaps=gen_mips,
                                      convention_dx=name.endswith(".dds"), force_no_compression=False)
    return out


def material_of(entry, fields):
    if entry["desc"] == "shading_node_desc_t":
        kw = dict(fields)
        kw["type"] = eShadingNode[kw.get("type", "Diffuse")]
        return ShadingNode(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()})
    return PrincipledMat(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in fields.items()})


# ---- the stand-ins for env.bin / env_floor.bin -----------------------------------------------------------------------------------------
def _quad(p0, e1, e2, uv_scale=1.0):
    """two triangles p0, p0 + e1, p0 + e1 + e2, p0 + e2 with the normal e1 x e2"""
    p0, e1, e2 = (np.asarray(v, dtype=np.float32) for v in (p0, e1, e2))
    n = np.cross(e1, e2)
    n = n / np.linalg.norm(n)
    P = np.stack([p0, p0 + e1, p0 + e1 + e2, p0 + e2])
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32) * uv_scale
    attrs = np.concatenate([P, np.tile(n, (4, 1)), uv], axis=1).astype(np.float32)
    return attrs, np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)


def _block(lo, hi):
    """outward-facing axis-aligned box"""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    faces = [((x0, y0, z0), (0, 0, z1 - z0), (0, y1 - y0, 0)), ((x1, y0, z0), (0, y1 - y0, 0), (0, 0, z1 - z0)),
             ((x0, y0, z0), (x1 - x0, 0, 0), (0, 0, z1 - z0)), ((x0, y1, z0), (0, 0, z1 - z0), (x1 - x0, 0, 0)),
             ((x0, y0, z0), (0, y1 - y0, 0), (x1 - x0, 0, 0)), ((x0, y0, z1), (x1 - x0, 0, 0), (0, y1 - y0, 0))]
    parts = [_quad(*f) for f in faces]
    attrs = np.concatenate([a for a, _ in parts])
    idx = np.concatenate([i + 4 * k for k, (_, i) in enumerate(parts)])
    return attrs, idx


def _mesh(scene, parts):
    """parts: [(attrs, idx, front, back)] -> one mesh with one group per part"""
    attrs, idx, groups, nv, ni = [], [], [], 0, 0
    for a, i, front, back in parts:
        attrs.append(a)
        idx.append(i + nv)
        groups.append((front, back, ni, len(i)))
        nv, ni = nv + len(a), ni + len(i)
    return scene.AddMesh(np.concatenate(attrs), np.concatenate(idx), groups)


def _room(scene, floor, walls, dark, light, mid):
    """a closed room around the material ball (stand-in for env.bin: floor, walls, three grey trims), inward-facing"""
    R, Hh = 0.9, 0.8
    parts = [(*_quad((-R, 0, -R), (0, 0, 2 * R), (2 * R, 0, 0), 8.0), floor, floor)]
    wall_quads = [((-R, 0, -R), (2 * R, 0, 0), (0, Hh, 0)), ((R, 0, R), (-2 * R, 0, 0), (0, Hh, 0)),
                  ((-R, 0, R), (0, 0, -2 * R), (0, Hh, 0)), ((R, 0, -R), (0, 0, 2 * R), (0, Hh, 0))]
    wa = [_quad(*q) for q in wall_quads]
    parts.append((np.concatenate([a for a, _ in wa]), np.concatenate([i + 4 * k for k, (_, i) in enumerate(wa)]), walls, walls))
    parts.append((*_quad((-R, Hh, -R), (2 * R, 0, 0), (0, 0, 2 * R)), dark, dark))          # ceiling
    parts.append((*_block((-0.45, 0.0, -0.55), (-0.25, 0.12, -0.35)), light, light))        # two blocks behind the ball
    parts.append((*_block((0.25, 0.0, -0.6), (0.4, 0.2, -0.45)), mid, mid))
    return _mesh(scene, parts)


def _floor_stage(scene, floor, dark, mid):
    """a floor with two blocks under the open sky (stand-in for env_floor.bin: three groups)"""
    R = 3.0
    parts = [(*_quad((-R, 0, -R), (0, 0, 2 * R), (2 * R, 0, 0), 24.0), floor, floor),
             (*_block((-0.45, 0.0, -0.55), (-0.25, 0.12, -0.35)), dark, dark),
             (*_block((0.25, 0.0, -0.6), (0.4, 0.2, -0.45)), mid, mid)]
    return _mesh(scene, parts)


# ---- the scene ---------------------------------------------------------------------------------------------------------------------------
OPEN_SKY = ("Standard_DirLight", "Standard_SunLight", "Standard_MoonLight", "Standard_HDRLight")
IDENTITY = np.eye(4, dtype=np.float32)


def _translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[3, :3] = (x, y, z)  # (column-major 4 x 4 as the reference takes it: the translation is elements 12..14)
    return m


def build(scene, entry, use_staged_meshes=True):
    """-> {"notes": [...substitutions...]}; the scene is finalized"""
    variant = entry["scene"]
    notes = []
    # camera (test_scene.cpp:233-302)
    cam = dict(type=0, filter=api.ePixelFilter.Box, view_transform=VIEW_TRANSFORM["Standard"], up=(0.0, 1.0, 0.0), regularize_alpha=0.0,
               min_total_depth=4, min_samples=entry["min_samples"], variance_threshold=entry["variance_threshold"])
    if variant in ("Standard_SunLight", "Standard_MoonLight"):
        cam["view_transform"] = VIEW_TRANSFORM["AgX"]
    elif variant == "Standard_DirLight":
        cam["view_transform"] = VIEW_TRANSFORM["Filmic_HighContrast"]
    if variant == "Refraction_Plane":
        cam.update(origin=(-0.074711, 0.099348, -0.049506), fwd=(0.725718915, 0.492017448, 0.480885535), fov=45.1806)
    else:
        cam.update(origin=(0.16149, 0.294997, 0.332965), fwd=(-0.364128768, -0.555621922, -0.747458696), fov=18.1806)
    if variant == "Standard_Clipped":
        cam.update(clip_start=0.4, clip_end=0.5)
    if variant == "Standard_DOF0":
        cam.update(sensor_height=0.018, focus_distance=0.1, fstop=0.1, lens_blades=6, lens_rotation=30.0 * 3.141592653589 / 180.0, lens_ratio=2.0)
    elif variant == "Standard_DOF1":
        cam.update(sensor_height=0.018, focus_distance=0.4, fstop=0.1, lens_blades=0, lens_rotation=30.0 * 3.141592653589 / 180.0, lens_ratio=2.0)
    elif variant in ("Standard_GlassBall0", "Standard_GlassBall1"):
        cam.update(max_diff_depth=8, max_spec_depth=8, max_refr_depth=8, max_total_depth=9)
    elif variant == "Ray_Flags":
        cam["regularize_alpha"] = 0.1
    elif variant == "Standard_SunLight":
        cam["exposure"] = -14.0
    elif variant == "Standard_MoonLight":
        cam["exposure"] = 8.0
    scene.set_current_cam(scene.AddCamera(**cam))

    # materials (:304-470)
    fields = add_needed_textures(scene, entry["fields"], entry["textures"], notes)
    main_mat = scene.AddMaterial(material_of(entry, fields))

    def grey(v):
        return scene.AddMaterial(PrincipledMat(base_color=(v, v, v), roughness=0.0, specular=0.0))
    floor_mat, walls_mat, white_mat = grey(0.75), grey(0.5), grey(0.64)
    light_grey, mid_grey, dark_grey = grey(0.32), grey(0.16), grey(0.08)
    square_light_mat = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=20.3718, importance_sample=True, base_color=(1.0, 1.0, 1.0)))
    disc_light_mat = scene.AddMaterial(ShadingNode(type=eShadingNode.Emissive, strength=81.4873, importance_sample=True, base_color=(1.0, 1.0, 1.0)))
    glossy_red = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(1.0, 0.0, 0.0)))
    glossy_green = scene.AddMaterial(ShadingNode(type=eShadingNode.Glossy, base_color=(0.0, 1.0, 0.0)))
    refr_mat_flags = scene.AddMaterial(PrincipledMat(roughness=0.0, transmission=1.0, ior=2.3))
    if variant == "Standard_GlassBall0":
        glass0 = scene.AddMaterial(ShadingNode(type=eShadingNode.Refractive, base_color=(1.0, 1.0, 1.0), roughness=0.0, ior=1.45))
        glass1 = scene.AddMaterial(ShadingNode(type=eShadingNode.Refractive, base_color=(1.0, 1.0, 1.0), roughness=0.0, ior=1.0))
    else:
        glass0 = scene.AddMaterial(PrincipledMat(base_color=(1.0, 1.0, 1.0), roughness=0.0, ior=1.45, transmission=1.0))
        glass1 = scene.AddMaterial(PrincipledMat(base_color=(1.0, 1.0, 1.0), roughness=0.0, ior=1.0, transmission=1.0))
    two_sided_back = scene.AddMaterial(PrincipledMat(base_color=(0.0, 0.0, 0.5), roughness=0.0))

    # meshes (:472-690); like the reference, some are added twice and one copy is removed before Finalize (storage compaction)
    doomed = []

    def mesh(name, mats, twice=False):
        attrs, idx, g = load_bin(name)
        groups = [(front, back, g[2 * k], g[2 * k + 1]) for k, (front, back) in enumerate(mats)]
        h = scene.AddMesh(attrs, idx, groups)
        if twice:
            doomed.append(h)
            h = scene.AddMesh(attrs, idx, groups)
        return h
    base_mesh = mesh("base.bin", [(mid_grey, None)])
    model_mesh = mesh("refr_plane.bin" if variant == "Refraction_Plane" else "model.bin", [(main_mat, None)])
    core_mesh = mesh("core.bin", [(mid_grey, None)])
    subsurf_bar_mesh = mesh("subsurf_bar.bin", [(white_mat, None), (dark_grey, None)])
    text_mesh = mesh("text.bin", [(white_mat, None)], twice=True)
    two_sided_mesh = mesh("two_sided.bin", [(main_mat, two_sided_back)])
    if variant in OPEN_SKY:
        doomed.append(_floor_stage(scene, floor_mat, dark_grey, mid_grey))
        env_mesh = _floor_stage(scene, floor_mat, dark_grey, mid_grey)
    else:
        doomed.append(_room(scene, floor_mat, walls_mat, dark_grey, light_grey, mid_grey))
        env_mesh = _room(scene, floor_mat, walls_mat, dark_grey, light_grey, mid_grey)
    notes.append("env_floor.bin: absent, a floor with two blocks" if variant in OPEN_SKY else "env.bin: absent, a closed room with two blocks")
    square_light_mesh = mesh("square_light.bin", [(square_light_mat, None), (dark_grey, None)], twice=True)
    disc_light_mesh = mesh("disc_light.bin", [(disc_light_mat, None), (dark_grey, None)])
    glassball_mesh = mesh("glassball.bin", [(glass0, None), (glass1, None)])
    box_mesh = mesh("box.bin", [(glossy_red, None)])
    box2_mesh = mesh("box.bin", [(refr_mat_flags, None)])
    box3_mesh = mesh("box.bin", [(glossy_green, None)])

    # instances (:692-790)
    s45 = 0.707106769
    model_xform = np.array([[s45, 0, s45, 0], [0, 1, 0, 0], [-s45, 0, s45, 0], [0, 0.062, 0, 1]], dtype=np.float32)
    if variant == "Refraction_Plane":
        scene.AddMeshInstance(model_mesh, IDENTITY)
    elif variant in ("Standard_GlassBall0", "Standard_GlassBall1"):
        scene.AddMeshInstance(glassball_mesh, _translate(0.0, 0.05, 0.0))
    elif variant == "Ray_Flags":
        def box(mesh_h, x, y, z, **vis):
            m = np.diag([0.01, 0.05, 0.01, 1.0]).astype(np.float32)
            m[3, :3] = (x, y, z)
            scene.AddMeshInstance(mesh_h, m, **vis)
        box(box_mesh, -0.05, 0.05, 0.0, shadow=False)
        box(box2_mesh, 0.0, 0.051, 0.0, specular=False)
        box(box_mesh, 0.05, 0.05, 0.0, diffuse=False)
        box(box3_mesh, -0.05, 0.05, -0.05, camera=False)
        box(box3_mesh, 0.0, 0.05, -0.05, refraction=False)
        box(box3_mesh, 0.05, 0.05, -0.05)
    elif variant == "Two_Sided":
        scene.AddMeshInstance(two_sided_mesh, _translate(0.0, 0.04, 0.0))
        scene.AddMeshInstance(base_mesh, IDENTITY)
        scene.AddMeshInstance(text_mesh, IDENTITY)
    else:
        scene.AddMeshInstance(model_mesh, model_xform)
        scene.AddMeshInstance(base_mesh, IDENTITY)
        scene.AddMeshInstance(core_mesh, IDENTITY)
        scene.AddMeshInstance(subsurf_bar_mesh, IDENTITY)
        scene.AddMeshInstance(text_mesh, IDENTITY)
    scene.AddMeshInstance(env_mesh, IDENTITY)

    # lights (:791-950)
    rect_xform = [-0.425036609, 2.24262476e-06, -0.905176163, 0.0, -0.876228273, 0.250873595, 0.411444396, 0.0,
                  0.227085724, 0.968019843, -0.106628500, 0.0, -0.436484009, 0.187178999, 0.204932004, 1.0]
    disk_xform = [0.813511789, -0.536388099, -0.224691749, 0.0, 0.538244009, 0.548162937, 0.640164733, 0.0,
                  -0.220209062, -0.641720533, 0.734644651, 0.0, 0.360500991, 0.461762011, 0.431780994, 1.0]
    if variant in ("Standard_MeshLights", "Refraction_Plane"):
        if variant != "Refraction_Plane":
            scene.AddMeshInstance(square_light_mesh, IDENTITY)
        scene.AddMeshInstance(disc_light_mesh, IDENTITY)
    elif variant in ("Standard", "Standard_DOF0", "Standard_DOF1", "Standard_GlassBall0", "Standard_GlassBall1", "Standard_Clipped", "Two_Sided"):
        scene.AddLight("rect", color=(20.3718, 20.3718, 20.3718), width=0.162, height=0.162, xform=rect_xform)
        scene.AddLight("disk", color=(81.4873, 81.4873, 81.4873), width=0.1296, height=0.1296, xform=disk_xform)
    elif variant == "Standard_SphereLight":
        scene.AddLight("sphere", color=(7.95775, 7.95775, 7.95775), position=(-0.436484, 0.187179, 0.204932), radius=0.05)
        line_xform = list(disk_xform)
        line_xform[12:15] = [0.0, 0.461762, 0.0]
        scene.AddLight("line", color=(80.0, 80.0, 80.0), radius=0.005, height=0.2592, xform=line_xform)
    elif variant == "Standard_InsideLight":
        scene.AddLight("sphere", color=(2.53302956, 2.53302956, 2.53302956), position=(0.0, 0.17, 0.0), radius=0.1)
    elif variant == "Standard_SpotLight":
        scene.AddLight("spot", color=(10.1321182, 10.1321182, 10.1321182), position=(-0.436484, 0.187179, 0.204932),
                       direction=(0.699538708, -0.130918920, -0.702499688), radius=0.05, spot_size=45.0, spot_blend=0.15)
    elif variant == "Standard_DirLight":
        scene.AddLight("directional", direction=(0.541675210, -0.541675210, -0.642787635), color=(12.0, 12.0, 12.0), angle=10.0)
    elif variant == "Standard_SunLight":
        scene.AddLight("directional", direction=(0.454519480, -0.454519480, -0.766044438), color=(144809.859, 129443.617, 127098.890), angle=4.0)
    elif variant == "Ray_Flags":
        scene.AddLight("sphere", color=(0.0253302939,) * 3, position=(-0.05, 0.2, 0.075), radius=0.0)

    # environment (:952-998)
    if variant in ("Standard_HDRLight", "Standard_Clipped"):
        notes.append("studio_small_03_2k.hdr: absent, ray_amd.scenes.rgbe_sky")
        sky = scene.AddTexture(scenes.rgbe_sky(256, 128), fmt=eTextureFormat.RGBA8888, is_srgb=False, generate_mipmaps=False, force_no_compression=True)
        rot = 2.35619449019 if variant == "Standard_HDRLight" else 0.0
        scene.SetEnvironment(env_col=(0.25, 0.25, 0.25), back_col=(0.25, 0.25, 0.25), env_map=sky, back_map=sky, env_map_rotation=rot, back_map_rotation=rot)
    elif variant == "Standard_SunLight":
        scene.SetEnvironment(env_col=(1.0, 1.0, 1.0), back_col=(1.0, 1.0, 1.0), env_map=api.PhysicalSkyTexture, back_map=api.PhysicalSkyTexture)
    elif variant == "Standard_MoonLight":
        scene.SetEnvironment(env_col=(1.0, 1.0, 1.0), back_col=(1.0, 1.0, 1.0), env_map=api.PhysicalSkyTexture, back_map=api.PhysicalSkyTexture,
                             clouds_density=0.4)
    else:
        scene.SetEnvironment(env_col=(0.0, 0.0, 0.0), back_col=(0.0, 0.0, 0.0))

    for h in doomed:
        scene.RemoveMesh(h)
    scene.Finalize()
    return {"notes": notes}


# ---- one entry of the matrix: the oracle and a backend context on the same scene ----------------------------------------------------------
def checkerboard(w, h, bucket=16):
    """the tiles `partial = true` renders (tests/test_scene.cpp:1029-1043: every other 16 x 16 bucket)"""
    rects, skip = [], False
    for y in range(0, h, bucket):
        skip = not skip
        for x in range(0, w, bucket):
            skip = not skip
            if not skip:
                rects.append((x, y, min(w - x, bucket), min(h - y, bucket)))
    return rects


def run_entry(entry, make_context, w, h, spp_cap=None, batched=False, threads=1):
    """Renders `entry` with the live oracle (RendererRef) and with the backend `make_context(w, h, blob)` returns a context of (the host build of
    the kernel sources, or the GPU), as run_material_test schedules it: min(max_samples, spp_cap) iterations over the whole frame or over the
    checkerboard of the `partial` test, then the NLM filter if the test asks for it (a UNet test is rendered and left unfiltered: the network
    has its own tests).  -> ({buffer: frame_metrics}, notes)"""
    import oracle_lib as O
    import util
    from ray_amd import hip
    spp = entry["max_samples"] if spp_cap is None else min(entry["max_samples"], spp_cap)
    ref = O.create_renderer(w, h, "REF")
    rs = ref.CreateScene()
    notes = build(rs, entry)["notes"]
    rects = checkerboard(w, h) if entry["partial"] else [(0, 0, w, h)]
    if threads > 1 and not entry["partial"]:
        ref.render_tiled_mt(rs, 32, spp, threads)
        regions = [api.RegionContext((0, 0, w, h))]
        regions[0].iteration = spp
    else:
        regions = [api.RegionContext(r) for r in rects]
        for region in regions:
            for _ in range(spp):
                ref.RenderScene(rs, region)
    ctx = make_context(w, h, O.export_scene(rs))
    for r in rects:
        if batched:
            ctx.render_batch(1, spp, rect=r)
        else:
            for it in range(1, spp + 1):
                ctx.render(it, rect=r)
    if entry["denoise"] == "NLM":
        for region, r in zip(regions, rects):
            ref.DenoiseImage(region)
            ctx.denoise_nlm(spp, rect=r)
    elif entry["denoise"] == "UNet":
        notes.append("UNet filter not applied (tests/test_gpu_unet.py has the network)")
    if entry["caching"]:
        notes.append("use_spatial_cache is out of scope (SURVEY.md 8): rendered without the cache on both sides")
    got = {"raw": (ctx.readback(hip.BUF_RAW), ref.get_raw_pixels_ref()),
           "final": (ctx.readback(hip.BUF_FINAL), ref.get_pixels_ref()),
           "base_color": (ctx.readback(hip.BUF_BASE_COLOR), ref.get_aux_pixels_ref(api.eAUXBuffer.BaseColor)),
           "depth_normals": (ctx.readback(hip.BUF_DEPTH_NORMALS), ref.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))}
    return {k: dict(util.frame_metrics(a, b), equal=bool(np.array_equal(a, b))) for k, (a, b) in got.items()}, notes


def run_entry_through_the_api(entry, w, h, spp_cap=None):
    """The same entry as the reference's harness drives it (run_material_test, tests/test_shading.cpp:34-212): CreateRenderer -> CreateScene ->
    setup_test_scene -> Resize down and up -> RenderScene per region and sample -> DenoiseImage (NLM, or the sixteen UNet passes) -> get_pixels_ref,
    once with RendererRef and once with RendererHIP -- scene construction, export and upload are the product's here (SceneHIP), not the oracle's.
    -> ({buffer: frame_metrics}, the two raw frames)"""
    import oracle_lib as O
    import util
    spp = entry["max_samples"] if spp_cap is None else min(entry["max_samples"], spp_cap)
    rects = checkerboard(w, h) if entry["partial"] else [(0, 0, w, h)]
    frames = []
    for kind in ("REF", "HIP"):
        r = O.create_renderer(w, h, "REF") if kind == "REF" else api.CreateRenderer(api.Settings(w, h), "HIP")
        s = r.CreateScene()
        build(s, entry)
        r.Resize(w // 2, h // 2)  # ("test Resize robustness", test_shading.cpp:103-106)
        r.Resize(w, h)
        regions = [api.RegionContext(rc) for rc in rects]
        for region in regions:
            for _ in range(spp):
                r.RenderScene(s, region)
        if entry["denoise"] == "NLM":
            for region in regions:
                r.DenoiseImage(region)
        elif entry["denoise"] == "UNet":
            n = r.InitUNetFilter()
            for region in regions:
                for p in range(n):
                    r.DenoiseImageUNet(p, region)
        frames.append({"raw": r.get_raw_pixels_ref().copy(), "final": r.get_pixels_ref().copy(),
                       "base_color": r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor).copy(),
                       "depth_normals": r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals).copy(), "keep": (r, s)})
    ref, got = frames
    return {k: util.frame_metrics(got[k], ref[k]) for k in ("raw", "final", "base_color", "depth_normals")}, (got["raw"], ref["raw"])
