#!/usr/bin/env python3
"""Generate the build-time files the reference's CPU backends need but /root/reference lacks.

BUILD TOOL shared by the oracle build (oracle/Makefile), the host library (ray_amd/host/Makefile) and the drop-in build: it writes the
generated directory it is given (all git-ignored), nothing else.  It is a generator, not oracle code: it lives under tools/.

* Config.h -- what CMake's configure_file would emit from Config.h.in (reference
  CMakeLists.txt:204) with ENABLE_REF_IMPL and ENABLE_SIMD_IMPL on, VK/DX off.
* stand-ins for the three headers named in /root/reference/.MISSING_LARGE_BLOBS:
  internal/precomputed/__3d_noise_tex.inl, __cirrus_tex.inl (physical-sky clouds, AtmosphereRef.cpp:8-9):
  deterministic tileable value noise, 32^3 and 64^2 x 2 channels (round 4: the physical sky is part of the path; rounds 1-3
  wrote zeros here).  The real textures are not in the tree; a host that has them hands them over through the same
  rayhip_scene_desc fields;
  __oidn_weights_hdr_alb_nrm.inl (UNet denoiser, UNetFilter.cpp:12-14): the trained network is not in the
  tree, so the 32 arrays are filled with DETERMINISTIC pseudo-random half-precision weights of the right
  shapes (uniform in +-sqrt(6 / fan_in): activations stay O(1) through the sixteen passes).  The filter then
  computes a well-defined function of its inputs -- not a denoised picture, but exactly the arithmetic the
  HIP restatement (ray_amd/csrc/unet_kernels.hip) has to reproduce pass by pass, with the same file feeding
  the oracle (oracle/_ref) and the product's host library (ray_amd/host: RendererHIP::InitUNetFilter takes
  the weights from the reference's own SetupUNetWeights).
"""
import os
import re
import sys


def main(ref: str, gen: str) -> None:
    os.makedirs(os.path.join(gen, "internal", "precomputed"), exist_ok=True)
    with open(os.path.join(gen, "Config.h"), "w") as f:
        f.write("#pragma once\n\n#define ENABLE_REF_IMPL\n#define ENABLE_SIMD_IMPL\n")

    pre = os.path.join(gen, "internal", "precomputed")
    import numpy as np
    # tileable value noise (a few octaves of a periodic random lattice, trilinearly smoothed): not the reference's Perlin-Worley
    # volume, but a texture of the same kind -- smooth, tileable, full 8-bit range -- so that the cloud code it feeds (GetCloudsDensity,
    # the cirrus layer: AtmosphereRef.cpp:321-357, 765-822) does real work.  Both sides read the same bytes: the oracle through the
    # compiled-in array, the product through rayhip_scene_desc::sky_noise3d_tex / sky_cirrus_tex.
    def tileable_noise(shape, octaves, seed):
        rng = np.random.Generator(np.random.PCG64(seed))
        out = np.zeros(shape, dtype=np.float64)
        amp, total = 1.0, 0.0
        for o in range(octaves):
            cells = 4 << o
            lattice = rng.uniform(0.0, 1.0, size=(cells,) * len(shape))
            idx = [np.arange(n) * cells / n for n in shape]
            acc = lattice
            for axis, x in enumerate(idx):
                i0 = np.floor(x).astype(int) % cells
                i1 = (i0 + 1) % cells
                t = x - np.floor(x)
                t = t * t * (3.0 - 2.0 * t)
                sh = [1] * len(shape)
                sh[axis] = len(x)
                acc = np.take(acc, i0, axis=axis) * (1.0 - t.reshape(sh)) + np.take(acc, i1, axis=axis) * t.reshape(sh)
            out += amp * acc
            total += amp
            amp *= 0.5
        out /= total
        out = (out - out.min()) / (out.max() - out.min())
        return np.clip(np.rint(out * 255.0), 0, 255).astype(np.uint8)

    noise_res, cirrus_res = 32, 64
    noise = tileable_noise((noise_res,) * 3, 3, 20240901)
    with open(os.path.join(pre, "__3d_noise_tex.inl"), "w") as f:
        f.write(f"extern const int NOISE_3D_RES = {noise_res};\nextern const uint8_t __3d_noise_tex[{noise.size}] = {{")
        f.write(",".join(str(int(v)) for v in noise.ravel()))
        f.write("};\n")
    cirrus = np.stack([tileable_noise((cirrus_res, cirrus_res), 4, 20240902), tileable_noise((cirrus_res, cirrus_res), 4, 20240903)], axis=-1)
    with open(os.path.join(pre, "__cirrus_tex.inl"), "w") as f:
        f.write(f"extern const int CIRRUS_TEX_RES = {cirrus_res};\nextern const uint8_t __cirrus_tex[{cirrus.size}] = {{")
        f.write(",".join(str(int(v)) for v in cirrus.ravel()))
        f.write("};\n")

    # weight array names are whatever UNetFilter.cpp pulls out of the namespace; shapes: the (out, in) channel counts of the
    # sixteen convolutions as SetupUNetWeights reorders them (UNetFilter.cpp:412-570), 3x3 kernels
    src = open(os.path.join(ref, "internal", "UNetFilter.cpp"), encoding="utf-8", errors="ignore").read()
    names = sorted(set(re.findall(r"unet_weights_hdr_alb_nrm::(\w+)", src)))
    shapes = {"enc_conv0": (32, 9), "enc_conv1": (32, 32), "enc_conv2": (48, 32), "enc_conv3": (64, 48), "enc_conv4": (80, 64),
              "enc_conv5a": (96, 80), "enc_conv5b": (96, 96), "dec_conv4a": (112, 96 + 64), "dec_conv4b": (112, 112),
              "dec_conv3a": (96, 112 + 48), "dec_conv3b": (96, 96), "dec_conv2a": (64, 96 + 32), "dec_conv2b": (64, 64),
              "dec_conv1a": (64, 64 + 9), "dec_conv1b": (32, 64), "dec_conv0": (3, 32)}
    with open(os.path.join(pre, "__oidn_weights_hdr_alb_nrm.inl"), "w") as f:
        for k, n in enumerate(names):
            layer, kind = n.rsplit("_", 1)
            out_ch, in_ch = shapes[layer]
            rng = np.random.Generator(np.random.PCG64(1234567 + k))  # (bit-reproducible across numpy versions)
            if kind == "weight":
                a = (6.0 / (9.0 * in_ch)) ** 0.5
                vals = rng.uniform(-a, a, size=out_ch * in_ch * 9)
            else:
                vals = rng.uniform(-0.1, 0.1, size=out_ch)
            bits = vals.astype(np.float16).view(np.uint16)
            f.write(f"const uint16_t {n}[{len(bits)}] = {{")
            f.write(",".join(str(int(b)) for b in bits))
            f.write("};\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
