"""The reference's own material test matrix on the reference's own test meshes.

tests/test_shading.cpp of the reference holds ninety material tests, tests/test_aux_channels.cpp one more (Oren-Nayar, Principled diffuse / sheen / glossy / specular / anisotropic /
metal / plastic / tint / emission / clearcoat, refraction with and without MIS, transmission, alpha, two-sided, seven textured "complex"
materials under every light type, depth of field, clipping, regions, adaptive sampling, ray-visibility flags).  Their golden images cannot be
reproduced in this checkout (env.bin and most textures are absent), but the matrix itself -- descriptors, scene variants, sample
counts -- is extracted mechanically (tests/golden/make_material_matrix.py -> material_matrix.json) and every entry is rendered on the reference's
mat_test meshes by the live oracle and by this backend (tests/ref_material_scene.py):

  * CPU (`-m "not gpu"`): the committed matrix equals a fresh extraction (when /root/reference is there), and a cross-section of it is BIT-EQUAL
    between the host build of the kernel sources and RendererRef -- raw, tonemapped, base colour and depth-normals (tools/material_matrix.py host
    runs all of them: profiles/r05/material_matrix_host.txt);
  * GPU (`-m gpu`): all of them through the C ABI within the stated tolerance of the oracle (tests/util.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
import ref_material_scene as M
import util
from ray_amd import hip

ENTRIES = M.matrix()
NAMES = [e["name"] for e in ENTRIES]
# one of every family and every scene variant that does not bake a sky (those take 5-7 s each on the host: tests/test_sky_bake.py has the sky)
CPU_SECTION = ["oren_mat1", "sheen_mat3", "aniso_mat5", "tint_mat1", "emit_mat0", "coat_mat1", "refr_mis1", "trans_mat4", "alpha_mat1", "alpha_mat4",
               "two_sided_mat", "complex_mat3", "complex_mat5_clipped", "complex_mat5_adaptive", "complex_mat5_regions", "complex_mat5_dof",
               "complex_mat5_mesh_lights", "complex_mat5_sphere_light", "complex_mat5_spot_light", "complex_mat5_dir_light", "complex_mat5_hdri_light",
               "complex_mat7_refractive", "ray_flags", "aux_channels"]


@pytest.fixture(scope="module")
def assets():
    if not M.have_assets():
        if os.path.isdir("/root/reference/tests/test_data"):
            subprocess.run([sys.executable, os.path.join(util.GOLDEN, "stage_ref_assets.py")], check=True, stdout=subprocess.DEVNULL)
        else:
            pytest.skip("tests/assets/_ref is not staged (tests/golden/stage_ref_assets.py needs /root/reference)")
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")


def test_the_committed_matrix_is_what_the_reference_says():
    assert len(ENTRIES) == 91 and len(set(NAMES)) == 91  # (ninety of test_shading.cpp + test_aux_channels.cpp)
    assert set(CPU_SECTION) <= set(NAMES)
    src = "/root/reference/tests/test_shading.cpp"
    if not os.path.exists(src):
        pytest.skip("/root/reference is not here: the committed matrix stands")
    sys.path.insert(0, util.GOLDEN)
    import make_material_matrix
    assert make_material_matrix.parse_all(src) == ENTRIES
    # every scene variant of the reference's enum is in it, and every descriptor field is one the API mirror knows
    assert {e["scene"] for e in ENTRIES} == {"Standard", "Standard_NoLight", "Refraction_Plane", "Standard_MeshLights", "Two_Sided", "Standard_Clipped",
                                             "Standard_DOF0", "Standard_DOF1", "Standard_SphereLight", "Standard_InsideLight", "Standard_SpotLight",
                                             "Standard_DirLight", "Standard_SunLight", "Standard_MoonLight", "Standard_HDRLight", "Standard_GlassBall0",
                                             "Standard_GlassBall1", "Ray_Flags"}


@pytest.mark.parametrize("name", CPU_SECTION)
def test_host_build_is_bit_equal_to_the_oracle(assets, name):
    if not O.have_hostsim():
        pytest.skip("tests/hostsim not built")
    entry = ENTRIES[NAMES.index(name)]
    m, _ = M.run_entry(entry, O.hostsim_context, 48, 48, spp_cap=(10 if name == "complex_mat5_adaptive" else 2))
    for buf in ("raw", "final", "base_color", "depth_normals"):
        assert m[buf]["equal"], (name, buf, m[buf])


@pytest.mark.parametrize("name", ["two_sided_mat", "aux_channels", "ray_flags"])
def test_a_scene_built_by_scene_hip_is_the_oracles_scene(assets, name):
    """the same entry built twice -- by the oracle's scene class and by the product's (SceneHIP, no renderer and no GPU needed: scene construction
    is host work), meshes added twice and one copy removed on both -- and the product's blob rendered by the host build: the oracle's frame, bit
    for bit"""
    from ray_amd import api
    if not (O.have_hostsim() and os.path.exists(api.HIP_HOST_LIB)):
        pytest.skip("tests/hostsim or libray_hip.so not built")
    entry = ENTRIES[NAMES.index(name)]
    w = h = 48
    ref = O.create_renderer(w, h, "REF")
    rs = ref.CreateScene()
    M.build(rs, entry)
    region = api.RegionContext((0, 0, w, h))
    for _ in range(2):
        ref.RenderScene(rs, region)
    s = api.CreateSceneHIP()
    M.build(s, entry)
    assert s.triangle_count() == rs.triangle_count()
    ctx = O.hostsim_context(w, h, api.export_scene_blob(s))
    assert np.array_equal(util.render_frames(ctx, 2), ref.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), ref.get_pixels_ref())


TREE_FORMS = {"4-wide": {"HOSTSIM_BVH4": "1", "HOSTSIM_BVH8": "0"}, "8-wide, refined leaves": {"HOSTSIM_BVH4": "0", "HOSTSIM_BVH8": "1", "HOSTSIM_REFINE": "2"},
              "rebuilt by the LBVH builder": {"HOSTSIM_LBVH": "4"}, "4-wide, refined leaves (the product's default)": {"HOSTSIM_BVH4": "1", "HOSTSIM_REFINE": "1"}}


@pytest.mark.parametrize("name,form", [("complex_mat3", f) for f in TREE_FORMS] + [("ray_flags", "rebuilt by the LBVH builder"),
                                                                                   ("ray_flags", "4-wide, refined leaves (the product's default)")])
def test_every_tree_form_over_the_reference_meshes_is_bit_equal(assets, name, form, monkeypatch):
    """the derived acceleration structures (quantised 4- and 8-wide nodes, leaves refined, both levels rebuilt from the triangle records by the
    builder the device runs) over REAL asset meshes -- 77 762-triangle ball with its seams and slivers, non-uniformly scaled instances with
    visibility masks: the oracle's frame bit for bit (the synthetic scenes of tests/test_hostsim_parity.py pin the same on generated geometry)"""
    from ray_amd import api
    if not O.have_hostsim():
        pytest.skip("tests/hostsim not built")
    for k in ("HOSTSIM_BVH4", "HOSTSIM_BVH8", "HOSTSIM_REFINE", "HOSTSIM_LBVH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in TREE_FORMS[form].items():
        monkeypatch.setenv(k, v)
    entry = ENTRIES[NAMES.index(name)]
    w = h = 48
    ref = O.create_renderer(w, h, "REF")
    rs = ref.CreateScene()
    M.build(rs, entry)
    region = api.RegionContext((0, 0, w, h))
    for _ in range(2):
        ref.RenderScene(rs, region)
    ctx = O.hostsim_context(w, h, O.export_scene(rs))
    assert ctx.bvh_width() == (8 if form.startswith("8") else 4 if form.startswith("4") else 2)
    assert np.array_equal(util.render_frames(ctx, 2), ref.get_raw_pixels_ref())


@pytest.fixture(scope="module")
def gpu_lib():
    import torch  # noqa: F401  (first: its HIP runtime opens the device)
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path"
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_against_the_oracle(assets, gpu_lib, name):
    entry = ENTRIES[NAMES.index(name)]

    def make(w, h, blob):
        ctx = hip.Context(0, gpu_lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        return ctx
    m, _ = M.run_entry(entry, make, 96, 96, spp_cap=8, batched=True, threads=max(1, min(16, os.cpu_count() or 1)))
    spp = min(entry["max_samples"], 8)
    raw = m["raw"]
    assert raw["frac_within"] >= util.MIN_FRACTION, (name, raw)
    assert raw["psnr"] >= (util.MIN_PSNR_8SPP if spp >= 8 else util.MIN_PSNR_1SPP), (name, raw)
    if entry["denoise"] != "NLM":  # (the filter rewrites alpha with its weighted mean: float arithmetic like the colours)
        assert raw["alpha_equal"], (name, raw)
    for buf in ("base_color", "depth_normals"):
        assert m[buf]["frac_within"] >= util.MIN_FRACTION, (name, buf, m[buf])
    assert np.isfinite(raw["max_abs"])


API_SECTION = ["complex_mat5_regions", "complex_mat5_adaptive", "complex_mat5_sun_light", "two_sided_mat", "complex_mat7_principled", "aux_channels",
               "complex_mat5_unet_filter", "complex_mat6_unet_filter", "ray_flags"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", API_SECTION)
def test_renderer_hip_as_the_reference_harness_drives_it(assets, gpu_lib, name):
    """run_material_test's own sequence behind the Ray API on both sides (ref_material_scene.run_entry_through_the_api): the scene is built by
    SceneHIP here, the sky of the sun-light variant baked on the device, and the three tests that end in the UNet filter run it (generated
    weights on both sides: the network amplifies the last-bit differences of the two frames, and RendererHIP runs its f16 form -- the bar of
    tests/test_gpu_unet.py)."""
    from ray_amd import api
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.fail("ray_amd/host/_build/libray_hip.so is missing")
    entry = ENTRIES[NAMES.index(name)]
    m, (got, ref) = M.run_entry_through_the_api(entry, 96, 96, spp_cap=8)
    if entry["denoise"] == "UNet":
        err = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(1.0, np.abs(ref[..., :3]))
        assert err.max() <= 1e-1 and err.mean() <= 2e-3, (name, float(err.max()), float(err.mean()))
        assert m["final"]["psnr"] >= 45.0, (name, m["final"])
    else:
        assert m["raw"]["frac_within"] >= util.MIN_FRACTION and m["raw"]["psnr"] >= util.MIN_PSNR_1SPP, (name, m["raw"])
    for buf in ("base_color", "depth_normals"):
        assert m[buf]["frac_within"] >= util.MIN_FRACTION, (name, buf, m[buf])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["complex_mat3", "ray_flags"])
@pytest.mark.parametrize("mode,leaf_max", [("RAYHIP_REBUILD_BVH", "4"), ("RAYHIP_REFINE_LEAVES", "1")])
def test_device_built_trees_over_the_reference_meshes(assets, gpu_lib, name, mode, leaf_max, monkeypatch):
    """both levels rebuilt / the fat leaves refined BY THE DEVICE BUILDER (tests/test_gpu_bvh_build.py has it on generated scenes) over the
    reference's asset meshes, against the oracle"""
    monkeypatch.setenv(mode, leaf_max)
    monkeypatch.delenv("RAYHIP_BVH_BUILD_ON_HOST", raising=False)
    entry = ENTRIES[NAMES.index(name)]

    def make(w, h, blob):
        ctx = hip.Context(0, gpu_lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        return ctx
    m, _ = M.run_entry(entry, make, 96, 96, spp_cap=8, batched=True, threads=max(1, min(16, os.cpu_count() or 1)))
    assert m["raw"]["frac_within"] >= util.MIN_FRACTION and m["raw"]["psnr"] >= util.MIN_PSNR_8SPP, (name, mode, m["raw"])
