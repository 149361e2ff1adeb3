"""The linear BVH builder (ray_amd/csrc/lbvh.h) and the two things it is used for (ray_amd/csrc/scene_rebuild.h), on the
host build of the sources: SURVEY.md section 8f, N1.

A BVH only culls, so a correct tree over the same triangle records must reproduce the oracle:
  * leaf refinement (the scene's trees, every leaf above `leaf_max` triangles replaced by a subtree; what librayhip does at
    upload) and
  * a full rebuild of both levels from triangles and instance transforms
are rendered through the BVH2 walk, the 4-wide walk and the 8-wide walk (rt_bvh8.h: its own collapse, octant-ordered slots,
its own triangle order) and compared with RendererRef's golden frames -- bit for bit on the
fixture scenes (instancing, transparency, every light kind).  What a different tree may legitimately change is the winner
of an EXACT tie between two triangles at the same distance (the reference's own tree flavours differ there, SURVEY
Appendix A.1): the material-zoo scene, which stacks coplanar surfaces, is allowed a handful of such pixels.
"""
import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import hip

SCENES = ["cornell_basic", "cornell_principled", "cornell_lights", "cornell_env", "cornell_instances"]
pytestmark = pytest.mark.skipif(not O.have_hostsim(), reason="tests/hostsim not built")


def _render(name, spp=8):
    ctx = O.hostsim_context(64, 64, util.golden_scene(name), pmj=util.pmj())
    for it in range(1, spp + 1):
        ctx.render(it)
    return ctx


@pytest.mark.parametrize("wide", ["0", "4", "8"], ids=["bvh2", "bvh4", "bvh8"])
@pytest.mark.parametrize("mode,leaf_max", [("HOSTSIM_REFINE", "1"), ("HOSTSIM_REFINE", "2"), ("HOSTSIM_REFINE", "4"),
                                           ("HOSTSIM_LBVH", "2"), ("HOSTSIM_LBVH", "4"), ("HOSTSIM_LBVH", "8")])
@pytest.mark.parametrize("name", SCENES)
def test_rebuilt_trees_reproduce_the_oracle(name, mode, leaf_max, wide, monkeypatch):
    monkeypatch.setenv(mode, leaf_max)
    monkeypatch.setenv("HOSTSIM_BVH4", "1" if wide == "4" else "0")
    monkeypatch.setenv("HOSTSIM_BVH8", "1" if wide == "8" else "0")
    g = util.golden_ref(name)
    ctx = _render(name)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
    # kernel level: the reference's primary hits, index for index
    _, hits, tc = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=0)
    util.assert_hits_identical(hits, g["primary_hits"])


def test_refinement_shortens_the_leaves(monkeypatch):
    """fewer triangle tests per ray, a few more node visits (the trade the upload makes): visit counters of the two trees"""
    name = "cornell_lights"
    g = util.golden_ref(name)
    counts = {}
    for leaf_max in ("0", "2"):
        monkeypatch.setenv("HOSTSIM_REFINE", leaf_max)
        ctx = O.hostsim_context(64, 64, util.golden_scene(name), pmj=util.pmj())
        _, _, tc = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=hip.FLAG_COUNT_TRAVERSAL)
        counts[leaf_max] = tc
    assert counts["2"]["tris"] < counts["0"]["tris"]
    assert counts["2"]["nodes"] >= counts["0"]["nodes"]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("wide8", ["0", "1"], ids=["bvh2", "bvh8"])
@pytest.mark.parametrize("mode,leaf_max", [("HOSTSIM_REFINE", "2"), ("HOSTSIM_LBVH", "4")])
def test_rebuilt_trees_on_live_scenes(mode, leaf_max, wide8, monkeypatch):
    """bigger meshes (the atrium at test size), many instances, and the scene with coplanar overlaps"""
    from ray_amd import scenes
    monkeypatch.setenv(mode, leaf_max)
    monkeypatch.setenv("HOSTSIM_BVH8", wide8)
    for fn, exact in ((scenes.atrium_small, True), (scenes.cornell_principled_zoo, False)):
        r, s = O.render_ref(fn, 64, 64, 4)
        ctx = O.hostsim_context(64, 64, O.export_scene(s))
        for it in range(1, 5):
            ctx.render(it)
        a, b = ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref()
        differing = int((np.abs(a - b).max(axis=-1) > 0).sum())
        print(fn.__name__, mode, leaf_max, "pixels differing:", differing)
        if exact:
            assert differing == 0
        else:  # exact-t ties between coplanar surfaces may resolve the other way round
            assert differing <= 8 and util.frame_metrics(a, b)["frac_within"] >= util.MIN_FRACTION


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("mode,leaf_max", [("HOSTSIM_REFINE", "2"), ("HOSTSIM_LBVH", "2"), ("HOSTSIM_BVH8", "1")])
@pytest.mark.parametrize("seed", range(6))
def test_rebuilt_trees_on_fuzzed_instance_scenes(seed, mode, leaf_max, monkeypatch):
    """random two-level scenes (2-9 instances of shared meshes, rotations, non-uniform scales, ray-type visibility masks,
    transparency, environment lights): the linear builder's trees -- groups of one or two primitives, meshes that are a single
    leaf, top levels of few instances -- against RendererRef"""
    from ray_amd import scenes
    monkeypatch.setenv(mode, leaf_max)
    if mode == "HOSTSIM_BVH8":
        monkeypatch.setenv("HOSTSIM_REFINE", "2")  # (what the upload does before the 8-wide collapse)
    w, h, spp = 48, 48, 3
    r, s = O.render_ref(lambda sc: scenes.random_instances(sc, seed), w, h, spp)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    for it in range(1, spp + 1):
        ctx.render(it)
    a, b = ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref()
    differing = int((np.abs(a - b).max(axis=-1) > 0).sum())
    print("seed", seed, mode, "pixels differing:", differing)
    # interpenetrating instances: an exact-distance tie between two surfaces may resolve the other way round
    assert differing <= 4 and util.frame_metrics(a, b)["frac_within"] >= util.MIN_FRACTION


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_instances"])
def test_small_group_sah_refinement_reproduces_the_oracle(name, monkeypatch):
    """scene_rebuild.h: SmallSahBuilder -- leaf refinement with surface-area-heuristic splits instead of the linear builder's
    Morton order (an alternative kept for comparison: on the atrium it saves 0.7 % of the node visits and 6 % of the triangle
    tests, not enough to take the refinement off the device builder)"""
    monkeypatch.setenv("HOSTSIM_REFINE", "2")
    monkeypatch.setenv("HOSTSIM_REFINE_SAH", "1")
    monkeypatch.setenv("HOSTSIM_BVH4", "1")
    g = util.golden_ref(name)
    ctx = _render(name)
    assert np.array_equal(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
