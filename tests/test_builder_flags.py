"""mesh_desc_t::allow_spatial_splits / use_fast_bvh_build (SceneBase.h:130-131; the reference's builder: internal/BVHSplit.cpp:148, 323-470).

Both flags are passed through to the reference's host-side build (ray_amd/host/ray_capi.cpp); what they change is the tree this tree's
upload path then works on: spatial splits DUPLICATE triangle references into several leaves -- which the leaf refinement
(scene_rebuild.h), the layout permutation (bvh_layout.h) and the 4- / 8-wide collapses (bvh4_build.h, bvh8_build.h) have to survive.  The
scene is built to provoke them (scenes.cornell_needles: long planks over a carpet of small triangles, one mesh).  VERDICT round 4, weak 1.
"""
import functools

import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import hip, scenes

FLAGS = [(True, False), (False, True), (True, True)]
WALKS = {
    "bvh2": {},
    "bvh4": {"HOSTSIM_BVH4": "1"},
    "bvh4_refined": {"HOSTSIM_BVH4": "1", "HOSTSIM_REFINE": "2"},
    "bvh8_refined": {"HOSTSIM_BVH8": "1", "HOSTSIM_REFINE": "2"},
}


@pytest.fixture(scope="module")
def gpu_lib():
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path, -m gpu tests cannot run here"
    return lib


def needles(splits, fast):
    return functools.partial(scenes.cornell_needles, spatial_splits=splits, fast_bvh_build=fast)


@pytest.mark.skipif(not (O.have_ref() and O.have_hostsim()), reason="oracle/_ref or tests/hostsim not built")
def test_spatial_splits_really_duplicate_references():
    """the test scene does what it is for: with allow_spatial_splits the reference's builder emits more triangle references and nodes"""
    _, plain = O.render_ref(needles(False, False), 8, 8, 1)
    _, split = O.render_ref(needles(True, False), 8, 8, 1)
    assert split.triangle_count() > plain.triangle_count() and split.node_count() > plain.node_count()


@pytest.mark.skipif(not (O.have_ref() and O.have_hostsim()), reason="oracle/_ref or tests/hostsim not built")
@pytest.mark.parametrize("walk", sorted(WALKS))
@pytest.mark.parametrize("splits,fast", FLAGS)
def test_host_build_is_bit_exact_under_the_builder_flags(splits, fast, walk, monkeypatch):
    """every form of the walk (the reference's BVH2 as it comes, 4-wide, 4- and 8-wide over refined leaves) over a tree with duplicated
    references against the live reference built with the same flags: all images bit for bit"""
    w, h, spp = 72, 56, 3
    r, s = O.render_ref(needles(splits, fast), w, h, spp)
    for k, v in WALKS[walk].items():
        monkeypatch.setenv(k, v)
    ctx = O.hostsim_context(w, h, O.export_scene(s))
    assert ctx.bvh_width() == {"bvh2": 2, "bvh4": 4, "bvh4_refined": 4, "bvh8_refined": 8}[walk]
    img = util.render_frames(ctx, spp)
    assert np.array_equal(img, r.get_raw_pixels_ref())
    assert np.array_equal(ctx.readback(hip.BUF_FINAL), r.get_pixels_ref())


@pytest.mark.gpu
@pytest.mark.parametrize("splits,fast", FLAGS)
def test_device_walk_under_the_builder_flags(gpu_lib, splits, fast):
    """the product's upload path (leaf refinement on the device, layout pass, 4-wide collapse) over the same trees: the closest hits of the
    oracle's primary rays are the oracle's (object, triangle, t, u, v) exactly, the frames within the stated tolerance"""
    w, h, spp = 96, 80, 8
    r, s = O.render_ref(needles(splits, fast), w, h, spp)
    blob = O.export_scene(s)
    rays, hits_in = O.ref_generate_primary_rays(s, w, h, 1)
    _, ref_hits = O.ref_intersect_closest(s, rays, hits_in, 1)
    ctx = hip.Context(0, gpu_lib)
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(blob)
    _, got, _ = ctx.k_intersect_closest(rays, hits_in, 1, flags=0)
    util.assert_hits_identical(got, ref_hits)
    ctx.render_batch(1, spp)
    m = util.frame_metrics(ctx.readback(hip.BUF_RAW), r.get_raw_pixels_ref())
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
