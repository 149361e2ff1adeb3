#!/usr/bin/env python3
"""Render one of the built-in scenes through the drop-in Ray API mirror (RendererHIP / SceneHIP) and write a TGA, the way
the reference's samples do (samples/00_basic/main.cpp:188-206).

    python examples/render.py [scene] [--size 1024 1024] [--spp 64] [--denoise] [--out image.tga]

scene: any key of ray_amd.scenes.SCENES (cornell_basic, cornell_principled, cornell_lights, cornell_env, ...), or
"sponza" / "bistro" for the procedural atria of bench.py.  Needs an AMD GPU (the HIP backend has no CPU path).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_amd import api, scenes  # noqa: E402


def write_tga(path: str, rgba: np.ndarray):
    """8-bit uncompressed true-colour TGA, top-left origin (the reference's WriteTGA, samples/utils)"""
    h, w = rgba.shape[:2]
    px = np.clip(rgba[..., [2, 1, 0]] * 255.0 + 0.5, 0, 255).astype(np.uint8)  # BGR
    header = bytearray(18)
    header[2] = 2
    header[12:14] = w.to_bytes(2, "little")
    header[14:16] = h.to_bytes(2, "little")
    header[16] = 24
    header[17] = 0x20
    with open(path, "wb") as f:
        f.write(header)
        f.write(px.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="cornell_basic")
    ap.add_argument("--size", nargs=2, type=int, default=(1024, 1024))
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--denoise", action="store_true", help="RendererBase::DenoiseImage (NLM) after the last sample")
    ap.add_argument("--out", default="image.tga")
    args = ap.parse_args()
    w, h = args.size

    r = api.CreateRenderer(api.Settings(w, h), "HIP")  # Ray::CreateRenderer(s, log, ..., eRendererType::HIP)
    print("device:", r.device_name())
    s = r.CreateScene()
    t0 = time.perf_counter()
    if args.scene in ("sponza", "bistro"):
        scenes.atrium(s, 0.36 if args.scene == "sponza" else 4.3)
    else:
        scenes.SCENES[args.scene](s)
    print(f"scene: {s.triangle_count()} triangles, built in {time.perf_counter() - t0:.2f} s")

    region = api.RegionContext((0, 0, w, h))
    t0 = time.perf_counter()
    for _ in range(args.spp):
        r.RenderScene(s, region)  # queued; RendererHIP renders them in a few wavefront passes
    if args.denoise:
        r.DenoiseImage(region)
    img = r.get_pixels_ref()  # tone-mapped RGBA (forces the pending iterations out)
    dt = time.perf_counter() - t0
    print(f"{args.spp} spp of {w}x{h} in {dt * 1e3:.1f} ms = {w * h * args.spp / dt / 1e6:.1f} Msamples/s")
    write_tga(args.out, img)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
