"""CPU gate: the kernel sources (ray_amd/csrc/rt_*.h) compiled for the host must reproduce RendererRef BIT FOR BIT.

tests/hostsim builds the very headers the HIP kernels are made of with g++ -msse2 -mno-avx (no fma, glibc libm --
the reference's own build flags), so any difference from the reference's golden vectors is a restatement error,
not a numerics artefact.  This is what lets the GPU tests attribute their (small) differences to the device.
Runs without a GPU.
"""
import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import hip

SCENES = ["cornell_basic", "cornell_principled", "cornell_lights", "cornell_env", "cornell_filmic", "cornell_instances"]


@pytest.fixture(scope="module")
def lib():
    if not O.have_hostsim():
        pytest.skip("tests/hostsim not built (run __graft_entry__.build())")
    return hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")


def test_rng_known_answers(lib):
    v = np.load(f"{util.GOLDEN}/rng_vectors.npz")
    ctx = hip.Context(0, lib)
    ctx.upload_static(util.pmj())
    out = ctx.k_scrambled_rand(v["dims"], v["seeds"], v["samples"])
    assert np.array_equal(out.view(np.uint32), v["xy"].view(np.uint32))


@pytest.mark.parametrize("name", SCENES)
def test_kernel_level_dumps(lib, name):
    g = util.golden_ref(name)
    ctx = util.make_context(lib, name)
    rays, hits = ctx.k_generate_primary_rays(1)
    assert rays.tobytes() == g["primary_rays"].tobytes()
    assert hits.tobytes() == g["primary_hits_in"].tobytes()
    rays2, hits2, tc = ctx.k_intersect_closest(rays, hits, 1)
    util.assert_hits_identical(hits2, g["primary_hits"])
    assert tc["rays"] == len(rays)
    rc, _ = ctx.k_intersect_shadow(g["shadow_rays"], 1)
    assert np.array_equal(rc, g["shadow_rc"])


@pytest.mark.parametrize("name", SCENES)
def test_frames_bit_exact(lib, name):
    g = util.golden_ref(name)
    ctx = util.make_context(lib, name)
    ctx.render(1)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp1"])
    for it in range(2, 9):
        ctx.render(it)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), g["final_spp8"])
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), g["base_color_spp8"])
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), g["depth_normals_spp8"])


def test_tile_sharding_and_rects(lib):
    name = "cornell_basic"
    w = h = 96
    full = util.render_frames(util.make_context(lib, name, w, h), 2)
    acc = np.zeros_like(full)
    for r in range(3):
        ctx = util.make_context(lib, name, w, h)
        ctx.set_shard(32, 3, r)
        acc += util.render_frames(ctx, 2)
    assert np.array_equal(acc, full)
    ctx = util.make_context(lib, name, w, h)
    for it in (1, 2):
        ctx.render(it, rect=(0, 0, w, 40))
        ctx.render(it, rect=(0, 40, w, h - 40))
    assert np.array_equal(ctx.readback(hip.BUF_RAW), full)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,w,h,spp", [("cornell_basic", 160, 96, 5), ("cornell_principled", 96, 160, 5),
                                              ("cornell_basic", 1, 1, 4), ("cornell_basic", 7, 3, 4), ("cornell_lights", 13, 9, 3),
                                              ("cornell_basic", 65, 1, 3)])
def test_live_reference_other_sizes(lib, name, w, h, spp):
    """non-square frames, single-pixel and ragged ones (not whole 8x8 ray-generation tiles), against the reference run
    live (not the fixtures)"""
    from ray_amd import api, scenes

    r, s = O.render_ref(scenes.SCENES[name], w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    img = util.render_frames(ctx, spp)
    assert np.array_equal(img, r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))


CAMERAS = {
    # thin-lens depth of field with a 6-blade rotated anamorphic aperture, shifted sensor (GeneratePrimaryRays :1493-1530)
    "dof_blades": dict(fstop=1.4, focus_distance=0.6, focal_length=0.05, lens_blades=6, lens_rotation=0.3, lens_ratio=1.5,
                       shift=(0.1, -0.05)),
    "dof_disk_flength": dict(fstop=2.0, focus_distance=0.7, ltype=1, focal_length=0.035, sensor_height=0.024),
    # pixel filters (Core.cpp filter tables through CDFInverted) and a clipped view range
    "gaussian_clip": dict(filter=1, filter_width=2.0, clip_start=0.3, clip_end=1.2),
    "blackman_harris": dict(filter=2, filter_width=1.5),
    # pass flags / depth limits of camera_desc_t
    "lighting_only": dict(lighting_only=1, max_diff_depth=2, max_total_depth=3, min_total_depth=1),
    "no_direct_no_bg": dict(skip_direct_lighting=1, no_background=1),
    "no_indirect": dict(skip_indirect_lighting=1),
    "clamped": dict(clamp_direct=2.0, clamp_indirect=1.0, regularize_alpha=0.1),
    # adaptive sampling: pixels drop out of the iteration once their variance estimate is below the threshold
    "adaptive": dict(min_samples=2, variance_threshold=0.05),
}


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("cam", sorted(CAMERAS))
def test_camera_features_against_live_reference(lib, cam):
    """camera_desc_t fields the fixtures do not cover, against the reference run live: lens model, sensor shift, pixel
    filters, clip range, lighting flags, clamps"""
    from ray_amd import api, scenes

    name = "cornell_principled" if cam in ("lighting_only", "clamped") else "cornell_basic"
    w, h, spp = 80, 64, (12 if cam == "adaptive" else 4)
    r, s = O.render_ref(scenes.SCENES[name], w, h, spp, **CAMERAS[cam])
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    img = util.render_frames(ctx, spp)
    assert np.array_equal(img, r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,cam", [("cornell_basic", {}), ("cornell_lights", {}), ("cornell_filmic", {}),
                                      ("cornell_basic", dict(min_samples=2, variance_threshold=0.02))])
def test_nlm_denoise_against_live_reference(lib, name, cam):
    """RendererBase::DenoiseImage(region): variance pre-filter + joint NLM + tonemap, on the full frame and
    on a sub-rect (the extended region then reaches into rendered pixels instead of clamping), with adaptive sampling
    (the filter re-arms required_samples) and with a look-up-table view transform"""
    from ray_amd import api, scenes

    w, h, spp = 72, 56, 6
    r, s = O.render_ref(scenes.SCENES[name], w, h, spp, **cam)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    util.render_frames(ctx, spp)
    noisy = ctx.readback(hip.BUF_RAW)
    for rect in ((0, 0, w, h), (10, 6, 40, 30)):
        region = api.RegionContext(rect)
        region._bind(O.ref_lib())
        region.iteration = spp  # what RenderScene calls on this region would have left
        r.DenoiseImage(region)
        ctx.denoise_nlm(spp, rect=rect)
        assert np.array_equal(ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref()), rect
        assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref()), rect
    assert not np.array_equal(noisy, ctx.readback(hip.BUF_RAW))
    # the adaptive-sampling flags the filter left behind steer the next iteration: render one more on both sides
    region = api.RegionContext((0, 0, w, h))
    region._bind(O.ref_lib())
    region.iteration = spp
    r.RenderScene(s, region)
    ctx.render(spp + 1)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_scene_mutation_against_live_reference(lib):
    """render, then RemoveMeshInstance / SetMeshInstanceTransform / RemoveLight / AddLight / SetEnvironment / Finalize, render
    again: the export of the mutated scene (sparse pools with freed slots, rebuilt TLAS and light tree) must reproduce the
    reference"""
    from ray_amd import api, scenes

    w, h = 64, 48
    r, s = O.render_ref(scenes.cornell_instances_mutable, w, h, 2)
    before = r.get_raw_pixels_ref().copy()
    scenes.mutate_instances_scene(s)
    region = api.RegionContext((0, 0, w, h))
    r.Clear()
    for _ in range(3):
        r.RenderScene(s, region)
    assert not np.array_equal(before, r.get_raw_pixels_ref())
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, 3), r.get_raw_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("wide", ["0", "1", "8"], ids=["bvh2", "bvh4", "bvh8"])
def test_instance_update_against_live_reference(lib, wide, monkeypatch):
    """the same mutation through the UPDATE path (rayhip_scene_update_instances; here the host build of its planning,
    scene_update.h, with the linear builder as host loops): the geometry of the first upload stays, instances / lights /
    environment are replaced and the top level is rebuilt -- frames must still be the reference's, including its habit of
    numbering top-level leaves by position among the live instances after a RemoveMeshInstance"""
    from ray_amd import api, scenes
    monkeypatch.setenv("HOSTSIM_BVH4", "1" if wide == "1" else "0")
    monkeypatch.setenv("HOSTSIM_BVH8", "1" if wide == "8" else "0")
    w, h = 64, 48
    r, s = O.render_ref(scenes.cornell_instances_mutable, w, h, 2)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, 2), r.get_raw_pixels_ref())
    scenes.mutate_instances_scene(s)
    region = api.RegionContext((0, 0, w, h))
    r.Clear()
    for _ in range(3):
        r.RenderScene(s, region)
    assert ctx.update_instances(O.export_scene(s)) == 0
    ctx.clear()
    assert np.array_equal(util.render_frames(ctx, 3), r.get_raw_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("wide", ["0", "1", "8"], ids=["bvh2", "bvh4", "bvh8"])
def test_instance_update_with_a_single_instance(lib, wide, monkeypatch):
    """the smallest top level: ONE instance (the linear builder has nothing to split: its root gets a far-away point box as
    second child), moved and rotated; then removed altogether (an empty scene with lights), then added again"""
    from ray_amd import api, scenes
    monkeypatch.setenv("HOSTSIM_BVH4", "1" if wide == "1" else "0")
    monkeypatch.setenv("HOSTSIM_BVH8", "1" if wide == "8" else "0")
    w, h, spp = 48, 48, 2
    handle = {}

    def one_instance(scene):
        scene.SetEnvironment(env_col=(0.2, 0.25, 0.3), back_col=(0.2, 0.25, 0.3))
        grey = scene.AddMaterial(api.ShadingNode(type=api.eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
        attrs, idx = scenes.cornell_mesh_arrays(scenes._block_quads("tall"))
        mesh = scene.AddMesh(attrs, idx, [(grey, None, 0, 30)])
        handle["mi"], handle["mesh"] = scene.AddMeshInstance(mesh), mesh
        scene.AddLight("sphere", color=(4.0, 4.0, 4.0), position=(-0.1, 0.5, 0.1), radius=0.03)
        scenes._cornell_camera(scene)
        scene.Finalize()

    r, s = O.render_ref(one_instance, w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())

    def check(mutate):
        mutate()
        s.Finalize()
        region = api.RegionContext((0, 0, w, h))
        r.Clear()
        for _ in range(spp):
            r.RenderScene(s, region)
        assert ctx.update_instances(O.export_scene(s)) == 0
        ctx.clear()
        assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())

    check(lambda: s.SetMeshInstanceTransform(handle["mi"], scenes._xform(translate=(0.1, 0.05, -0.05), rot_y_deg=30.0, scale=(1.2, 0.8, 1.0))))
    moved = r.get_raw_pixels_ref().copy()
    check(lambda: s.RemoveMeshInstance(handle["mi"]))
    assert not np.array_equal(moved, r.get_raw_pixels_ref())
    check(lambda: handle.update(mi=s.AddMeshInstance(handle["mesh"], scenes._xform(translate=(-0.1, 0.0, 0.1)))))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_instance_update_of_a_field_of_instances(lib):
    """200 instances (every 64th a lamp: its triangle lights move along), all moved, Finalize: update path against the live
    reference; and a scene with other geometry is turned away with 2 (= upload it)"""
    from ray_amd import api, scenes
    w, h, spp = 64, 48, 2
    r, s = O.render_ref(lambda sc: scenes.instance_field(sc, 200), w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    scenes.move_instance_field(s)
    region = api.RegionContext((0, 0, w, h))
    r.Clear()
    for _ in range(spp):
        r.RenderScene(s, region)
    assert ctx.update_instances(O.export_scene(s)) == 0
    ctx.clear()
    a, b = util.render_frames(ctx, spp), r.get_raw_pixels_ref()
    differing = int((np.abs(a - b).max(axis=-1) > 0).sum())
    print("pixels differing:", differing)
    assert differing <= 2  # (interpenetrating blocks: an exact-distance tie between two instances may resolve the other way round)
    assert ctx.update_instances(util.golden_scene("cornell_lights")) == 2
    ctx.clear()
    assert np.array_equal(util.render_frames(ctx, spp), a)  # ... and the scene on the "device" is untouched


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_clear_and_resize_against_live_reference(lib):
    """RendererBase::Clear (full / half <- colour, required_samples re-armed; RendererCPU.h:297-301) and Resize followed by
    more iterations"""
    from ray_amd import api, scenes

    w, h = 56, 40
    r, s = O.render_ref(scenes.cornell_basic, w, h, 3)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    util.render_frames(ctx, 3)
    r.Clear((0.25, 0.5, 0.75, 1.0))
    ctx.clear((0.25, 0.5, 0.75, 1.0))
    region = api.RegionContext((0, 0, w, h))  # a cleared region: iterations restart at 1
    for it in (1, 2):
        r.RenderScene(s, region)
        ctx.render(it)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())
    w2, h2 = 72, 48
    r.Resize(w2, h2)
    ctx.resize(w2, h2)
    assert r.size() == (w2, h2)
    region = api.RegionContext((0, 0, w2, h2))
    for it in (1, 2, 3):
        r.RenderScene(s, region)
        ctx.render(it)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("scene", ["empty_scene", "lights_only_scene"])
def test_degenerate_scenes_against_live_reference(lib, scene):
    """no geometry (no TLAS, empty BVH / triangle / vertex arrays), with and without lights"""
    from ray_amd import scenes

    w, h, spp = 48, 32, 3
    r, s = O.render_ref(getattr(scenes, scene), w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert float(np.abs(r.get_raw_pixels_ref()).max()) > 0.0


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_light_flags_against_live_reference(lib):
    """light_desc flags the fixtures leave at their defaults: cast_shadow, diffuse / specular / refraction visibility,
    multiple_importance (SceneBase.h light descriptors -> light_t flag bits)"""
    from functools import partial
    from ray_amd import scenes

    flags = {"sphere": dict(cast_shadow=False), "spot": dict(specular_visibility=False), "rect": dict(diffuse_visibility=False),
             "disk": dict(refraction_visibility=False, multiple_importance=False), "line": dict(multiple_importance=False),
             "directional": dict(cast_shadow=False, specular_visibility=False)}
    w, h, spp = 64, 64, 6
    r, s = O.render_ref(partial(scenes.cornell_lights, light_flags=flags), w, h, spp)
    r0, _ = O.render_ref(scenes.cornell_lights, w, h, spp)
    assert not np.array_equal(r.get_raw_pixels_ref(), r0.get_raw_pixels_ref())
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_material_zoo_against_live_reference(lib):
    """principled_mat_desc_t / shading_node_desc_t corners the fixtures do not reach (anisotropy, transmission, sheen and
    specular tints, emission inside Principled, alpha textures, mix_add, textured Glossy)"""
    from ray_amd import api, scenes

    w, h, spp = 72, 64, 6
    r, s = O.render_ref(scenes.cornell_principled_zoo, w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_delta_lights_and_texture_corners_against_live_reference(lib):
    """delta lights, emitters without importance sampling / with a texture, different front and back materials,
    non-power-of-two and 1x1 textures with mip chains"""
    from ray_amd import api, scenes

    w, h, spp = 72, 64, 6
    r, s = O.render_ref(scenes.cornell_delta_lights, w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("wide", ["0", "1", "8"])
def test_atrium_against_live_reference(lib, wide, monkeypatch):
    """the procedural atrium bench.py renders (Sponza- / Bistro-class at full detail), at a detail the CPU reference
    finishes in seconds: deep SAH BVH, thousands of emissive triangles in the light tree, textures; BVH2 and 4-wide walks"""
    from functools import partial
    from ray_amd import scenes

    monkeypatch.setenv("HOSTSIM_BVH4", "1" if wide == "1" else "0")
    monkeypatch.setenv("HOSTSIM_BVH8", "1" if wide == "8" else "0")
    monkeypatch.setenv("HOSTSIM_REFINE", "2" if wide == "8" else "0")  # (the 8-wide collapse is built over refined leaves)
    w, h, spp = 96, 54, 3
    r, s = O.render_ref(partial(scenes.atrium, detail=0.02), w, h, spp)
    assert s.triangle_count() > 5000
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("instanced", [False, True])
@pytest.mark.parametrize("wide", ["0", "1"])
def test_asset_street_against_live_reference(lib, instanced, wide, monkeypatch):
    """the Bistro-class street of bench.py's `bistro_assets` / `bistro_assets_inst` workloads (copies of the reference's own mat_test/model.bin
    along a street) with two copies: baked into one mesh / as instances under a top-level tree; host build of the kernel
    sources against the live reference bit for bit, BVH2 and 4-wide walks"""
    from functools import partial
    from ray_amd import scenes

    if not scenes.have_asset_meshes():
        pytest.skip("tests/assets/_ref/meshes.npz not staged (needs the reference tree: tests/golden/stage_ref_assets.py)")
    monkeypatch.setenv("HOSTSIM_BVH4", wide)
    monkeypatch.setenv("HOSTSIM_BVH8", "0")
    monkeypatch.setenv("HOSTSIM_REFINE", "0")
    w, h, spp = 96, 54, 2
    r, s = O.render_ref(partial(scenes.street_assets, copies=2, instanced=instanced), w, h, spp)
    assert s.triangle_count() > 100000
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert float(r.get_raw_pixels_ref()[..., :3].sum()) > 0.0


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_scenes_against_live_reference(lib, seed):
    """a small fuzzer: materials (all node types, Mix trees, every Principled parameter), lights (all types, delta and
    area, all flags) and camera / pass settings drawn from a seed; host build vs the reference, bit for bit"""
    from functools import partial
    from ray_amd import api, scenes

    w, h, spp = 56, 48, 3
    r, s = O.render_ref(partial(scenes.random_cornell, seed=seed), w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_random_instanced_scenes_against_live_reference(lib, seed):
    """the second fuzzer: 2-9 instances of shared meshes with random transforms and ray-type visibility, HDR environment
    with / without importance sampling, sky portals; frames and the NLM-denoised frames, bit for bit"""
    from functools import partial
    from ray_amd import api, scenes

    w, h, spp = 56, 48, 3
    r, s = O.render_ref(partial(scenes.random_instances, seed=seed), w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    region = api.RegionContext((0, 0, w, h))
    region._bind(O.ref_lib())
    region.iteration = spp
    r.DenoiseImage(region)
    ctx.denoise_nlm(spp)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())


# the pixels of random_instances(seed 6012), 64 x 48, 4 spp, whose primary or secondary ray runs exactly along the shared diagonal of two
# triangles of a quad: both triangles report the hit, at the same distance to the last bit, and whichever is tested SECOND takes it
# (IntersectTri accepts t_new <= t_old; CoreRef.cpp:24-50).  Which one that is depends on the order of the triangle records: the
# reference's own tree flavours disagree there, and so do its leaves and the refined ones.
TIE_SCENE = dict(seed=6012, w=64, h=48, spp=4, pixels={(7, 17), (47, 27)})


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_fresnel_weighted_mix_against_live_reference(lib):
    """a Mix node with an ior: its weight is multiplied by the dielectric Fresnel term for the medium outside the surface, read from
    the ray's ior stack (ShadeRef.cpp's mix loop) -- in a scene where nothing refracts (scenes.cornell_fresnel_mix)"""
    from ray_amd import scenes

    r, s = O.render_ref(scenes.cornell_fresnel_mix, 64, 64, 4)
    ctx = O.hostsim_context(64, 64, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, 4), r.get_raw_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_the_tie_pixels_of_the_refined_leaves_are_pinned(lib, monkeypatch):
    """round 3's device fuzzing found ONE scene in 240 where the default product (leaves refined to <= 2 triangles) leaves the
    reference: two pixels.  Pinned here: with the reference's leaves the wide walk equals RendererRef bit for bit; with refined
    leaves exactly those two pixels differ, nothing else (a third pixel would be a regression of the refinement, not a tie)"""
    from functools import partial
    from ray_amd import scenes

    t = TIE_SCENE
    r, s = O.render_ref(partial(scenes.random_instances, seed=t["seed"]), t["w"], t["h"], t["spp"])
    ref = r.get_raw_pixels_ref()
    blob = O.export_scene(s)
    monkeypatch.setenv("HOSTSIM_BVH4", "1")
    frames = {}
    for refine in ("0", "2"):
        monkeypatch.setenv("HOSTSIM_REFINE", refine)
        ctx = O.hostsim_context(t["w"], t["h"], blob)
        frames[refine] = util.render_frames(ctx, t["spp"]).copy()
    assert np.array_equal(frames["0"], ref)
    ys, xs = np.nonzero(np.abs(frames["2"] - ref).max(axis=-1) > 0)
    assert set(zip(xs.tolist(), ys.tolist())) == t["pixels"]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed,compress", [(1, False), (2, True), (3, False), (4, True), (5, True), (6, False)])
def test_random_textures_against_live_reference(lib, seed, compress):
    """the third fuzzer: texture slots of Principled / Diffuse / Emissive materials with random sizes (1-pixel and
    non-power-of-two included), channel formats, sRGB flags and mip chains, with and without settings_t::use_tex_compression"""
    from functools import partial
    from ray_amd import api, scenes

    w, h, spp = 56, 48, 3
    r, s = O.render_ref(partial(scenes.random_textures, seed=seed), w, h, spp, use_tex_compression=compress)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))


def test_physical_sky_golden_frames(lib):
    """the physical sky (environment = Ray::PhysicalSkyTexture, a directional light as the sun): narrow rays that leave
    the scene -- camera rays, the mirror block's reflections -- go through the analytic integrator (rt_sky.h: air, the cloud layer
    with its shadow marches, cirrus, sun disk, stars, moon), wide ones read the map the host baked.  Host build against the
    committed frames of RendererRef, bit for bit"""
    g = util.golden_ref("cornell_sky")
    ctx = util.make_context(lib, "cornell_sky")
    ctx.render(1)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp1"])
    for it in range(2, 9):
        ctx.render(it)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), g["final_spp8"])
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), g["base_color_spp8"])
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), g["depth_normals_spp8"])
    top = g["raw_spp8"][:8, :, :3]
    assert float(g["raw_spp8"][..., :3].max()) > 10.0 and ((top[..., 2] > 0.1) & (top[..., 2] > top[..., 0])).mean() > 0.3, "the top rows look into a blue sky"


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("night", [False, True])
def test_physical_sky_against_live_reference(lib, night):
    """... and against the live reference: by day (sun above the horizon) and by night (the sun below it: moonlight on the clouds,
    stars, the moon's textured disk), another frame size, the rect form"""
    from functools import partial
    from ray_amd import api, scenes

    w, h, spp = 48, 40, 3
    r, s = O.render_ref(partial(scenes.cornell_sky, night=night), w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))
    assert not np.isnan(r.get_raw_pixels_ref()).any()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_sky_portals_against_live_reference(lib):
    """rect / disk lights with sky_portal = true over an environment map"""
    from ray_amd import api, scenes

    w, h, spp = 64, 64, 6
    r, s = O.render_ref(scenes.cornell_portals, w, h, spp)
    r0, _ = O.render_ref(scenes.cornell_env, w, h, spp)
    assert not np.array_equal(r.get_raw_pixels_ref(), r0.get_raw_pixels_ref())
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("decode_at_export", ["0", "1"], ids=["blocks", "decoded"])
def test_compressed_textures_against_live_reference(lib, decode_at_export, monkeypatch):
    """settings_t::use_tex_compression (the reference's default): RGBA/RGB/R inputs and normal maps land in the BC3
    (YCoCg) / BC4 / BC5 storages.  They cross the boundary as the reference keeps them -- 4x4 blocks, decoded per fetch by
    rt_texture.h (RAYHIP_TEX_RAW_BC) -- or decoded at export with the reference's own block decoder; either way the frames
    must equal the reference's rendered from the compressed data"""
    from ray_amd import api, scenes
    monkeypatch.setenv("RAY_HIP_DECODE_BC", decode_at_export)

    w, h, spp = 64, 64, 4
    r, s = O.render_ref(scenes.cornell_textures, w, h, spp, use_tex_compression=True)
    r0, s0 = O.render_ref(scenes.cornell_textures, w, h, spp)
    ctx0 = O.hostsim_context(w, h, O.export_scene(s0))  # the same scene in the RGB / R / RG storages
    assert np.array_equal(util.render_frames(ctx0, spp), r0.get_raw_pixels_ref())
    assert not np.array_equal(r.get_raw_pixels_ref(), r0.get_raw_pixels_ref()), "compression did not change a texel: not exercised"
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    img = util.render_frames(ctx, spp)
    assert np.array_equal(img, r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))


@pytest.mark.parametrize("decode_at_export", ["0", "1"], ids=["blocks", "decoded"])
def test_block_compressed_inputs_against_live_reference(lib, decode_at_export, monkeypatch):
    """pre-compressed inputs (eTextureFormat::BC1 / BC3 / BC4 / BC5 -> TexStorageBCn<3|4|1|2>::AllocateRaw) made of random
    bytes -- every selector, both end-point orders, mip levels, sizes that are not multiples of four: the per-fetch block
    decoder of rt_texture.h against TexStorageBCn::Get (TextureStorageCPU.h:381-541), through whole frames"""
    from ray_amd import api, scenes
    monkeypatch.setenv("RAY_HIP_DECODE_BC", decode_at_export)
    w, h, spp = 64, 64, 4
    r, s = O.render_ref(scenes.cornell_block_textures, w, h, spp)
    blob = O.export_scene(s)
    ctx = O.hostsim_context(w, h, blob)
    assert np.array_equal(util.render_frames(ctx, spp), r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_BASE_COLOR), r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor))
    assert np.array_equal(ctx.readback(hip.BUF_DEPTH_NORMALS), r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals))
    if decode_at_export == "0":  # blocks are a quarter (BC1, BC4) / half... of one byte per texel and channel: 8 or 16 bytes per 16 texels
        monkeypatch.setenv("RAY_HIP_DECODE_BC", "1")
        assert len(blob) < len(O.export_scene(s))


@pytest.mark.parametrize("name", SCENES)
def test_layout_pass_is_exercised(lib, name):
    """librayhip re-orders nodes and triangles at upload (ray_amd/csrc/bvh_layout.h); the host build runs the same pass,
    so the bit-exact tests above cover it -- provided it did not silently fall back to the input order"""
    import ctypes
    ctx = util.make_context(lib, name)
    f = lib.lib.hostsim_layout_applied
    f.argtypes, f.restype = [ctypes.c_void_p], ctypes.c_int
    assert f(ctx._ctx) == 1


@pytest.mark.parametrize("width", ["4", "8"])
@pytest.mark.parametrize("name", SCENES)
def test_wide_bvh_is_bit_exact(lib, name, width, monkeypatch):
    """the quantised wide BLAS forms the GPU kernels walk -- 8-wide (ray_amd/csrc/rt_bvh8.h, the product default: octant-ordered
    slots, one stack entry per level, its own triangle order) and 4-wide (rt_bvh4.h): conservative boxes + the reference's own
    triangle test must reproduce RendererRef bit for bit, hits and frames"""
    monkeypatch.setenv("HOSTSIM_BVH4", "1" if width == "4" else "0")
    monkeypatch.setenv("HOSTSIM_BVH8", "1" if width == "8" else "0")
    if width == "8":
        monkeypatch.setenv("HOSTSIM_REFINE", "2")
    g = util.golden_ref(name)
    ctx = util.make_context(lib, name)
    _, hits, tc = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=0)
    assert tc["nodes4"] > 0, "the wide walk counts its 4-wide node visits: zero means the BVH2 path ran"
    util.assert_hits_identical(hits, g["primary_hits"])
    ctx.render(1)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp1"])
    for it in range(2, 9):
        ctx.render(it)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
