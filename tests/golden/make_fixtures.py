#!/usr/bin/env python3
"""Generates tests/golden/* from the REAL reference (oracle/_ref/libray_ref.so, built by oracle/Makefile).

Run in the build container (needs /root/reference to have been compiled once):
    python tests/golden/make_fixtures.py
The GPU box has no reference tree; these committed files are what the `-m gpu` tests and smoke() fall back on,
and what pins the hostsim/HIP restatement to RendererRef even if the oracle library is absent.

Files
  pmj02_samples.npy                      the reference's PMJ02 table (Core.h:363-368) as uint32[262144]
  <scene>.rayscene                       flat scene + camera + filter table (ray_amd/csrc/scene_blob.h)
  <scene>_ref.npz                        RendererRef outputs: raw_spp1, raw_spp8 (get_raw_pixels_ref), final_spp8,
                                         base_color_spp8, depth_normals_spp8; kernel-level dumps for iteration 1:
                                         primary rays/hits (GeneratePrimaryRays), hits after IntersectScene,
                                         ShadePrimary's image + secondary rays + shadow rays (shade0_color,
                                         secondary_rays, shadow_rays), the shadow rays' IntersectScene(shadow) results;
                                         bounce 1: the secondary rays after IntersectScene (+ their hits) and
                                         ShadeSecondary's image / rays (shade1_color, secondary_rays1, shadow_rays1)
  cornell_sky.rayscene / _ref.npz        the physical sky (scenes.FRAME_SCENES): frames only -- raw_spp1 / raw_spp8 / final / aux
  rng_vectors.npz                        get_scrambled_2d_rand known answers
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from ray_amd import api, scenes  # noqa: E402

W = H = 64


def main():
    np.save(os.path.join(HERE, "pmj02_samples.npy"), O.pmj_table())

    rs = np.random.RandomState(1234)
    dims = rs.randint(0, 2 + 8 * 9, size=4096).astype(np.uint32)
    seeds = rs.randint(0, 2**32, size=4096, dtype=np.uint64).astype(np.uint32)
    samples = rs.randint(0, 8192, size=4096).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "rng_vectors.npz"), dims=dims, seeds=seeds, samples=samples,
                        xy=O.ref_scrambled_rand(dims, seeds, samples))

    for name, fn in scenes.SCENES.items():
        r = O.create_renderer(W, H, "REF")
        s = r.CreateScene()
        fn(s)
        with open(os.path.join(HERE, f"{name}.rayscene"), "wb") as f:
            f.write(O.export_scene(s))
        out = {}
        # kernel-level dumps, iteration 1
        rays, hits = O.ref_generate_primary_rays(s, W, H, 1)
        out["primary_rays"], out["primary_hits_in"] = rays, hits
        rays2, hits2 = O.ref_intersect_closest(s, rays, hits, 1)
        out["primary_hits"] = hits2
        out["primary_rays_traced"] = rays2  # throughput / depth as IntersectScene left them (transparent surfaces crossed)
        color, sec, sh = O.ref_shade(s, W, H, 0, 1, rays2, hits2, np.zeros((H, W, 4), np.float32))
        out["shade0_color"], out["secondary_rays"], out["shadow_rays"] = color, sec, sh
        out["shadow_rc"] = O.ref_intersect_shadow(s, sh, 1)
        # bounce 1 of the same iteration: the secondary rays traced (hits preset like RendererCPU.h:533-535) and shaded by
        # ShadeSecondary on top of the bounce-0 image
        hits_in = np.zeros(len(sec), dtype=hits.dtype)
        hits_in["obj_index"], hits_in["prim_index"], hits_in["t"], hits_in["v"] = -1, -1, 3.402823466e+30, -1.0
        sec1, hits1 = O.ref_intersect_closest(s, sec, hits_in, 1)
        out["secondary_rays_traced"], out["secondary_hits"] = sec1, hits1
        color1, sec2, sh1 = O.ref_shade(s, W, H, 1, 1, sec1, hits1, color)
        out["shade1_color"], out["secondary_rays1"], out["shadow_rays1"] = color1, sec2, sh1
        # frames
        region = api.RegionContext((0, 0, W, H))
        for it in range(1, 9):
            r.RenderScene(s, region)
            if it == 1:
                out["raw_spp1"] = r.get_raw_pixels_ref()
        out["raw_spp8"] = r.get_raw_pixels_ref()
        out["final_spp8"] = r.get_pixels_ref()
        out["base_color_spp8"] = r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor)
        out["depth_normals_spp8"] = r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals)
        np.savez_compressed(os.path.join(HERE, f"{name}_ref.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})

    # scenes pinned by their FRAMES only (no kernel-level dumps): the physical sky -- its rays are finished by a pass of their own
    # (ShadeSkyPrimary / ShadeSkySecondary, RendererCPU.h:484-486, 555-557), which the kernel-level shade hook does not model
    for name, fn in scenes.FRAME_SCENES.items():
        r = O.create_renderer(W, H, "REF")
        s = r.CreateScene()
        fn(s)
        with open(os.path.join(HERE, f"{name}.rayscene"), "wb") as f:
            f.write(O.export_scene(s))
        out = {}
        region = api.RegionContext((0, 0, W, H))
        for it in range(1, 9):
            r.RenderScene(s, region)
            if it == 1:
                out["raw_spp1"] = r.get_raw_pixels_ref()
        out["raw_spp8"] = r.get_raw_pixels_ref()
        out["final_spp8"] = r.get_pixels_ref()
        out["base_color_spp8"] = r.get_aux_pixels_ref(api.eAUXBuffer.BaseColor)
        out["depth_normals_spp8"] = r.get_aux_pixels_ref(api.eAUXBuffer.DepthNormals)
        np.savez_compressed(os.path.join(HERE, f"{name}_ref.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
