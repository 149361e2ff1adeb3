"""The acceleration-structure builder ON THE DEVICE (ray_amd/csrc/lbvh.hip.h behind rayhip_scene_upload): the GPU twin of
tests/test_bvh_build.py (which runs the same element functions as host loops).  SURVEY.md section 8f, N1.

RAYHIP_REBUILD_BVH=<leaf_max> throws the scene's trees away and builds both levels from the triangle records and the
instance transforms with the device builder; RAYHIP_REFINE_LEAVES=<leaf_max> keeps the trees and replaces fat leaves (the
default, leaf_max 2, which every other GPU test already runs through).  A BVH only culls: over the same triangle records
a correct tree reproduces the oracle's primary hits index for index and the golden frames within the frame tolerance --
on the device, in the wide walk the product launches and in the instrumented BVH2 walk.
"""
import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import api, hip, scenes

pytestmark = pytest.mark.gpu

SCENES = ["cornell_basic", "cornell_principled", "cornell_lights", "cornell_env", "cornell_instances"]


@pytest.fixture(scope="module")
def gpu_lib():
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path"
    return lib


@pytest.mark.parametrize("mode,leaf_max", [("RAYHIP_REBUILD_BVH", "2"), ("RAYHIP_REBUILD_BVH", "4"), ("RAYHIP_REBUILD_BVH", "8"),
                                           ("RAYHIP_REFINE_LEAVES", "1"), ("RAYHIP_REFINE_LEAVES", "4")])
@pytest.mark.parametrize("name", SCENES)
def test_device_built_trees_reproduce_the_oracle(gpu_lib, name, mode, leaf_max, monkeypatch):
    monkeypatch.setenv(mode, leaf_max)
    monkeypatch.delenv("RAYHIP_BVH_BUILD_ON_HOST", raising=False)
    g = util.golden_ref(name)
    ctx = util.make_context(gpu_lib, name)  # the upload reads the switches
    # kernel level: the reference's primary hits, index for index, by the product walk and by the instrumented BVH2 walk
    for flags in (0, hip.FLAG_COUNT_TRAVERSAL):
        _, hits, _ = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=flags)
        ref = g["primary_hits"]
        hit = ref["v"] >= 0
        assert np.array_equal(hits["obj_index"], ref["obj_index"])
        assert np.array_equal(hits["prim_index"][hit], ref["prim_index"][hit])
        for f in ("t", "u", "v"):
            np.testing.assert_allclose(hits[f][hit], ref[f][hit], rtol=1e-5, atol=1e-6)
    frame = util.render_frames(ctx, 8)
    m = util.frame_metrics(frame, g["raw_spp8"])
    print(name, mode, leaf_max, m)
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"]


@pytest.mark.parametrize("leaf_max", ["2", "4"])
def test_device_rebuild_equals_the_host_loops(gpu_lib, leaf_max, monkeypatch):
    """device builder vs the same element functions run as host loops (RAYHIP_BVH_BUILD_ON_HOST=1): the same trees, hence
    the same visit counters and the same bits in the frame"""
    name = "cornell_instances"
    g = util.golden_ref(name)
    monkeypatch.setenv("RAYHIP_REBUILD_BVH", leaf_max)
    out = {}
    for on_host in ("0", "1"):
        monkeypatch.setenv("RAYHIP_BVH_BUILD_ON_HOST", on_host)
        ctx = util.make_context(gpu_lib, name)
        _, hits, tc = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=hip.FLAG_COUNT_WIDE)
        out[on_host] = (tc, hits.tobytes(), util.render_frames(ctx, 4).tobytes())
    assert out["0"][0] == out["1"][0]
    assert out["0"][1] == out["1"][1] and out["0"][2] == out["1"][2]


@pytest.mark.parametrize("leaf_max", ["2", "4"])
def test_device_rebuild_of_the_atrium_against_renderer_ref(gpu_lib, leaf_max, monkeypatch):
    """the benchmarked scene family (one 0.27 M-triangle mesh, deep trees) with BOTH levels built on the device, against the
    live oracle at 480 x 270, 1 and 4 spp"""
    if not O.have_ref():
        pytest.fail("oracle/_ref/libray_ref.so is missing on the GPU box")
    import bench
    monkeypatch.setenv("RAYHIP_REBUILD_BVH", leaf_max)
    w, h = 480, 270
    wl = dict(bench.WORKLOADS["sponza"])
    ref = O.create_renderer(w, h, "REF")
    ref_scene = ref.CreateScene()
    bench.build_scene(ref_scene, wl)
    hs = api.CreateSceneHIP()
    bench.build_scene(hs, wl)
    ctx = hip.Context(0, gpu_lib)
    ctx.upload_static(api.pmj_table())
    ctx.resize(w, h)
    ctx.upload_scene_blob(api.export_scene_blob(hs))
    threads, _ = bench.usable_cpus()
    done = 0
    for spp, bar in ((1, util.MIN_PSNR_1SPP), (4, util.MIN_PSNR_1SPP)):
        ref.render_tiled_mt(ref_scene, 32, spp - done, threads, iterations_done=done)
        done = spp
        ctx.clear()
        ctx.render_batch(1, spp)
        m = util.frame_metrics(ctx.readback(hip.BUF_RAW), ref.get_raw_pixels_ref())
        print("atrium, device rebuild leaf_max", leaf_max, spp, "spp", m)
        assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= bar and m["alpha_equal"], m


def test_a_box_the_grid_cannot_hold_falls_back_to_the_bvh2(gpu_lib, monkeypatch):
    """a child box with an infinite coordinate cannot be quantised onto the 8-bit grid of a wide node: that is a property of the
    scene, not an error -- the upload succeeds, the kernels walk the BVH2 as handed over (rayhip_scene_bvh_width == 2) and the
    frame is the BVH2 walk's frame of the untouched scene (the widened box still contains its subtree: culling only)"""
    import struct
    import test_hostile_scenes as H
    monkeypatch.setenv("RAYHIP_REFINE_LEAVES", "0")  # (the patched node must reach the collapse as it is)
    blob = util.golden_scene("cornell_basic")
    _, moff, _ = H.sections(blob)["mesh_instances"]
    _, noff, nsize = H.sections(blob)["nodes"]
    nodes = np.frombuffer(blob, dtype=np.uint32, count=nsize // 4, offset=noff).reshape(-1, 16)
    # a LEAF child's box (an inner child is opened by the collapse and its own box never quantised): walk down from the instance's root
    todo, site = [int(struct.unpack_from("<I", blob, moff + 4)[0])], None
    while todo and site is None:
        n = todo.pop()
        for side in (0, 1):
            w = int(nodes[n, 12 + side])
            if w & (7 << 29):
                site = (n, side)
                break
            todo.append(w)
    n, side = site
    bad = H.patched(blob, "nodes", 64 * n + (4 if side == 0 else 16 + 4), "<f", float("inf"))  # ch_data0[1] / ch_data1[1]: that child's max x

    def frame(b, width=None):
        if width:
            monkeypatch.setenv("RAYHIP_BVH_WIDTH", width)
        ctx = hip.Context(0, gpu_lib)
        ctx.upload_static(util.pmj())
        ctx.resize(64, 64)
        ctx.upload_scene_blob(b)
        monkeypatch.delenv("RAYHIP_BVH_WIDTH", raising=False)
        ctx.render_batch(1, 4)
        return ctx.bvh_width(), ctx.readback(hip.BUF_RAW)

    w_bad, f_bad = frame(bad)
    w_ok, f_ok = frame(blob)
    w_2, f_2 = frame(blob, "2")
    assert (w_bad, w_ok, w_2) == (2, 4, 2)
    assert np.array_equal(f_bad, f_2) and np.array_equal(f_ok, f_2)
