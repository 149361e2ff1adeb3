#!/usr/bin/env python3
"""Generate the build-time files the reference's CPU backends need but /root/reference lacks.

TEST INFRASTRUCTURE (oracle build).  Writes ONLY under oracle/_ref/gen (git-ignored):

* Config.h -- what CMake's configure_file would emit from Config.h.in (reference
  CMakeLists.txt:204) with ENABLE_REF_IMPL and ENABLE_SIMD_IMPL on, VK/DX off.
* zero-filled stand-ins for the three headers named in /root/reference/.MISSING_LARGE_BLOBS:
  internal/precomputed/__3d_noise_tex.inl, __cirrus_tex.inl (physical-sky clouds,
  AtmosphereRef.cpp:8-9) and __oidn_weights_hdr_alb_nrm.inl (UNet denoiser, UNetFilter.cpp:12-14).
  Neither subsystem is on the hot path (OUT OF SCOPE) and neither is exercised
  by any scene used here; the stubs only satisfy the linker.
"""
import os
import re
import sys


def main(ref: str, gen: str) -> None:
    os.makedirs(os.path.join(gen, "internal", "precomputed"), exist_ok=True)
    with open(os.path.join(gen, "Config.h"), "w") as f:
        f.write("#pragma once\n\n#define ENABLE_REF_IMPL\n#define ENABLE_SIMD_IMPL\n")

    pre = os.path.join(gen, "internal", "precomputed")
    with open(os.path.join(pre, "__3d_noise_tex.inl"), "w") as f:
        f.write("extern const int NOISE_3D_RES = 2;\nextern const uint8_t __3d_noise_tex[8] = {0};\n")
    with open(os.path.join(pre, "__cirrus_tex.inl"), "w") as f:
        f.write("extern const int CIRRUS_TEX_RES = 2;\nextern const uint8_t __cirrus_tex[8] = {0};\n")

    # weight array names are whatever UNetFilter.cpp pulls out of the namespace
    src = open(os.path.join(ref, "internal", "UNetFilter.cpp"), encoding="utf-8", errors="ignore").read()
    names = sorted(set(re.findall(r"unet_weights_hdr_alb_nrm::(\w+)", src)))
    with open(os.path.join(pre, "__oidn_weights_hdr_alb_nrm.inl"), "w") as f:
        for n in names:
            f.write(f"const uint16_t {n}[1] = {{0}};\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
