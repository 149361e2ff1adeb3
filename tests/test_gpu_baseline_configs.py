"""The BASELINE.json configurations at their stated frame sizes, HIP path against the LIVE oracle on the GPU box.

The oracle library (oracle/_ref/libray_ref.so: the reference itself, built by oracle/Makefile) travels to the GPU box, so
RendererRef renders the very scene the HIP path renders -- built twice from the same ray_amd/scenes.py function, once
into the reference's CPU scene and once into SceneHIP -- on all usable host cores (32x32 tiles from a queue), and the raw
linear frames are compared in the stated tolerance (tests/util.py):

    >= 99.5 % of the pixels within 1e-3 * max(1, |ref|);  PSNR (linear, clamped) >= 55 dB at 1 spp, >= 70 dB at >= 64 spp

  config 2   samples/00_basic Cornell box, 1024 x 1024, 64 spp in ONE batched pass (the stated 256 spp would cost the
             scalar oracle minutes of CPU for no extra information: the 64-spp bar is the one the tolerance names)
  config 3   Sponza-class atrium (0.27 M triangles), 1920 x 1080: 1 spp, and 20 spp in one 20-layer batched pass
  config 4   Bistro-class atrium (3.0 M triangles, THE benchmarked scene), 1920 x 1080: the same, and the stated 64 spp in
             one 64-layer pass against the 70 dB bar
  config 5   samples/03_principled, 2048 x 2048: 1, 8 and 64 spp (the stated 512 would cost the scalar oracle ten minutes)
plus, kernel level, on the benchmarked 3.0 M-triangle scene: the closest-hit kernel on the reference's own bounce-0
(coherent) and bounce-2 (incoherent) rays -- rays the oracle generated, traced and shaded itself -- must return the
reference's (obj_index, prim_index) exactly (SURVEY.md section 8d "value distributions"), exact-distance ties aside.
"""
import os
import sys
import time

import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import api, hip, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload table + usable_cpus: the tests measure what bench.py measures)

pytestmark = pytest.mark.gpu


def _need_oracle():
    if not O.have_ref():
        pytest.fail("oracle/_ref/libray_ref.so is missing on the GPU box (it is built by __graft_entry__.build() and ships "
                    "with the snapshot)")
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.fail("ray_amd/host/_build/libray_hip.so is missing (SceneHIP)")


class Workload:
    """one BASELINE scene, built into the oracle's scene and into the HIP path"""

    def __init__(self, name, w=None, h=None):
        _need_oracle()
        self.wl = dict(bench.WORKLOADS[name])
        if w:
            self.wl["w"], self.wl["h"] = w, h
        self.w, self.h = self.wl["w"], self.wl["h"]
        self.threads, _ = bench.usable_cpus()
        t0 = time.time()
        self.ref = O.create_renderer(self.w, self.h, "REF")
        self.ref_scene = self.ref.CreateScene()
        bench.build_scene(self.ref_scene, self.wl)
        hs = api.CreateSceneHIP()
        bench.build_scene(hs, self.wl)
        self.ctx = hip.Context(0, hip.Library())
        self.ctx.upload_static(api.pmj_table())
        self.ctx.resize(self.w, self.h)
        self.ctx.upload_scene_blob(api.export_scene_blob(hs))
        self.ref_spp = 0
        print(f"[{name}] scenes built + uploaded in {time.time() - t0:.1f} s; oracle threads: {self.threads}")

    def ref_frame(self, spp):
        """RendererRef after `spp` iterations (continues from where the last call stopped)"""
        assert spp >= self.ref_spp
        if spp > self.ref_spp:
            t = self.ref.render_tiled_mt(self.ref_scene, 32, spp - self.ref_spp, self.threads, iterations_done=self.ref_spp)
            print(f"  RendererRef: {spp - self.ref_spp} spp in {t:.1f} s")
            self.ref_spp = spp
        return self.ref.get_raw_pixels_ref()


def _check(img, ref, min_psnr, what):
    m = util.frame_metrics(img, ref)
    print(what, m)
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= min_psnr and m["alpha_equal"], (what, m)


@pytest.mark.parametrize("name", ["sponza", "bistro", "bistro_tex"])
def test_atrium_1080p_against_renderer_ref(name):
    """configs 3 and 4 at 1920 x 1080: iteration 1, then 20 iterations in one 20-layer pass (the shape bench.py times);
    bistro_tex: config 4 with a texture set (mip-mapped base-colour / normal / roughness maps on every large surface)"""
    wk = Workload(name)
    wk.ctx.render(1)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(1), util.MIN_PSNR_1SPP, f"{name} 1080p 1 spp")
    wk.ctx.clear()
    assert wk.ctx.max_batch() >= 20
    wk.ctx.render_batch(1, 20)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(20), util.MIN_PSNR_8SPP, f"{name} 1080p 20 spp (one batched pass)")
    if name == "bistro":
        # THE headline configuration at its stated sample count: 64 spp in one 64-layer pass against RendererRef continued to
        # 64 iterations -- the ">= 70 dB at >= 64 spp" half of the stated tolerance (reference bars: tests/test_shading.cpp:351-353)
        wk.ctx.clear()
        assert wk.ctx.max_batch() >= 64
        wk.ctx.render_batch(1, 64)
        _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(64), util.MIN_PSNR_64SPP, f"{name} 1080p 64 spp (one 64-layer pass)")


@pytest.mark.parametrize("name", ["bistro_assets", "bistro_assets_inst"])
def test_asset_street_1080p_against_renderer_ref(name):
    """config 4 as BASELINE.md 4.3 (4) specifies it -- 38 copies of the reference's own asset mesh (mat_test/model.bin, 3.0 M triangles of real
    geometry) along a street, baked into one mesh / as 38 mesh instances under a top-level tree -- at 1920 x 1080: iteration 1, then the
    stated 64 spp in one 64-layer pass against RendererRef continued to 64 iterations"""
    if not scenes.have_asset_meshes():
        pytest.fail("tests/assets/_ref/meshes.npz is missing on the GPU box (staged by __graft_entry__.build(), ships with the snapshot)")
    wk = Workload(name)
    wk.ctx.render(1)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(1), util.MIN_PSNR_1SPP, f"{name} 1080p 1 spp")
    wk.ctx.clear()
    assert wk.ctx.max_batch() >= 64
    wk.ctx.render_batch(1, 64)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(64), util.MIN_PSNR_64SPP, f"{name} 1080p 64 spp (one 64-layer pass)")


def test_cornell_1024_64spp_against_renderer_ref():
    """config 2 at 1024 x 1024: 1 spp, then 64 spp -- the '>= 70 dB at >= 64 spp' half of the stated tolerance"""
    wk = Workload("cornell")
    wk.ctx.render(1)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(1), util.MIN_PSNR_1SPP, "cornell 1024^2 1 spp")
    wk.ctx.clear()
    wk.ctx.render_batch(1, 64)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(64), util.MIN_PSNR_64SPP, "cornell 1024^2 64 spp")


def test_principled_2048_against_renderer_ref():
    """config 5 at 2048 x 2048 (textured Principled, NEE through the light tree): 1 spp and 8 spp"""
    wk = Workload("principled")
    wk.ctx.render(1)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(1), util.MIN_PSNR_1SPP, "03_principled 2048^2 1 spp")
    wk.ctx.clear()
    wk.ctx.render_batch(1, 8)
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(8), util.MIN_PSNR_8SPP, "03_principled 2048^2 8 spp")
    wk.ctx.clear()
    n = min(64, wk.ctx.max_batch())
    for first in range(1, 65, n):
        wk.ctx.render_batch(first, min(n, 65 - first))
    _check(wk.ctx.readback(hip.BUF_RAW), wk.ref_frame(64), util.MIN_PSNR_64SPP, "03_principled 2048^2 64 spp")


def _default_hits(n):
    h = np.zeros(n, dtype=hip.HIT_DTYPE)
    h["obj_index"], h["prim_index"], h["t"], h["v"] = -1, -1, 3.402823466e+30, -1.0
    return h


def test_closest_hit_kernel_on_reference_rays_of_the_benchmarked_scene():
    """K2 on the 3.0 M-triangle scene: bounce-0 and bounce-2 rays produced by the oracle's own GeneratePrimaryRays /
    IntersectScene / ShadePrimary / ShadeSecondary chain; the product kernel (4-wide quantised BLAS) must return the
    reference's (obj_index, prim_index) for every ray, and t / u / v to 1e-5"""
    w, h = 960, 540  # a quarter of the rays of the full frame through the same scene (the oracle chain is single-threaded)
    wk = Workload("bistro", w, h)
    s = wk.ref_scene
    t0 = time.time()
    rays0, hits0_in = O.ref_generate_primary_rays(s, w, h, 1)
    rays0_t, hits0 = O.ref_intersect_closest(s, rays0, hits0_in, 1)
    _, rays1, _ = O.ref_shade(s, w, h, 0, 1, rays0_t, hits0, np.zeros((h, w, 4), np.float32))
    rays1_t, hits1 = O.ref_intersect_closest(s, rays1, _default_hits(len(rays1)), 1)
    _, rays2, _ = O.ref_shade(s, w, h, 1, 1, rays1_t, hits1, np.zeros((h, w, 4), np.float32))
    rays2_t, hits2 = O.ref_intersect_closest(s, rays2, _default_hits(len(rays2)), 1)
    print(f"oracle chain: {len(rays0)} / {len(rays1)} / {len(rays2)} rays at bounce 0 / 1 / 2 in {time.time() - t0:.1f} s")
    assert len(rays2) > 50_000
    for label, rays_in, hits_in, ref_rays, ref_hits in (("bounce 0", rays0, hits0_in, rays0_t, hits0),
                                                        ("bounce 2", rays2, _default_hits(len(rays2)), rays2_t, hits2)):
        got_rays, got_hits, _ = wk.ctx.k_intersect_closest(rays_in, hits_in, 1, flags=0)
        hit = ref_hits["v"] >= 0
        same_obj = got_hits["obj_index"] == ref_hits["obj_index"]
        same_prim = (got_hits["prim_index"] == ref_hits["prim_index"]) | ~hit
        bad = ~(same_obj & same_prim)
        print(f"{label}: {len(rays_in)} rays, {int(hit.sum())} hits, index mismatches: {int(bad.sum())}")
        # The only legitimate difference is an EXACT tie: two triangles met at bit-identical t (a ray through a shared edge of
        # coplanar neighbours), where the reference keeps whichever it tests last and the order of the
        # tests is the order of the leaves -- which the leaf refinement of the upload changes.  Such a ray must carry the
        # reference's t to the bit, and there may be a few per million at most.
        assert int(bad.sum()) <= max(2, len(rays_in) // 100_000), f"{label}: {int(bad.sum())} rays with a different (obj_index, prim_index)"
        assert np.array_equal(got_hits["t"][bad].view(np.uint32), ref_hits["t"][bad].view(np.uint32)), f"{label}: a mismatch that is not an exact tie"
        same = hit & ~bad
        for f in ("t", "u", "v"):
            np.testing.assert_allclose(got_hits[f][same], ref_hits[f][same], rtol=1e-5, atol=1e-6)
        assert np.array_equal(got_rays["depth"], ref_rays["depth"])
