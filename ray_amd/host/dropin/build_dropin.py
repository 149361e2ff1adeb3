#!/usr/bin/env python3
"""Compile the drop-in FOR REAL: the reference tree with the INTEGRATION.md patch applied, its own sample unchanged.

    python ray_amd/host/dropin/build_dropin.py [REF=/root/reference]

What it does (everything lands under ray_amd/host/_build/dropin/, which is git-ignored; the scratch copy of the reference
is also kept off the GPU box by .gpurunignore -- only the linked binary travels):
  1. copies the reference's library sources (root *.h / *.cpp, internal/, third-party/renderdoc) and samples/00_basic/main.cpp
     to _build/dropin/Ray/ -- a scratch copy, never committed;
  2. applies the registration patch of INTEGRATION.md section 2, edit by edit, each anchored on the reference text it
     replaces (an anchor that is not found is an error: the patch in the document and the tree have drifted apart):
        RendererBase.h     enum entry eRendererType::HIP, RendererGPU mask
        RendererBase.cpp   "HIP" <-> eRendererType::HIP in both name tables
        Ray.h              DefaultEnabledRenderTypes |= HIP
        Ray.cpp            #include "internal/RendererHIP.h", factory branch in front of the Vulkan one
        Config.h           ENABLE_REF_IMPL + ENABLE_HIP_IMPL (what CMake would write from Config.h.in + the new option)
     and drops in internal/RendererHIP.h (ray_amd/host/RendererHIP.h);
  3. compiles libRay's translation units (Reference backend + scene code; no SIMD / Vulkan / DX), ray_amd/host/RendererHIP.cpp
     from where it lies, and samples/00_basic/main.cpp UNCHANGED, and links them with librayhip.so into
     _build/dropin/sample_00_basic.
`Ray::CreateRenderer(s, &Ray::g_stdout_log)` in that sample -- default arguments, not a line touched -- then returns the
HIP renderer on a box with an MI355X and falls through to the Reference renderer without one (constructor throws, the
factory logs and goes on: Ray.cpp:58-63's convention).  tests/test_gpu_dropin.py runs the binary on the GPU box and compares its
TGA with the ctypes path; tests/test_dropin_build.py checks here that the patch applies and that the binary, on a box
without a GPU, falls back and says so.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.join(ROOT, "ray_amd", "host", "_build", "dropin")

EDITS = {
    "RendererBase.h": [
        ("    Vulkan,\n    DirectX12\n};", "    Vulkan,\n    DirectX12,\n    HIP // AMD Instinct (gfx950) through librayhip\n};"),
        ("const Bitmask<eRendererType> RendererGPU = Bitmask<eRendererType>{eRendererType::Vulkan} | eRendererType::DirectX12;",
         "const Bitmask<eRendererType> RendererGPU =\n    Bitmask<eRendererType>{eRendererType::Vulkan} | eRendererType::DirectX12 | eRendererType::HIP;"),
    ],
    "RendererBase.cpp": [
        ("    case eRendererType::DirectX12:\n        return \"DX\";\n", "    case eRendererType::DirectX12:\n        return \"DX\";\n    case eRendererType::HIP:\n        return \"HIP\";\n"),
        ("    } else if (name == \"DX\") {\n        return eRendererType::DirectX12;\n    }",
         "    } else if (name == \"DX\") {\n        return eRendererType::DirectX12;\n    } else if (name == \"HIP\") {\n        return eRendererType::HIP;\n    }"),
    ],
    "Ray.h": [
        ("eRendererType::SIMD_AVX2 | eRendererType::SIMD_NEON | eRendererType::Vulkan | eRendererType::DirectX12;",
         "eRendererType::SIMD_AVX2 | eRendererType::SIMD_NEON | eRendererType::Vulkan | eRendererType::DirectX12 |\n    eRendererType::HIP;"),
    ],
    "Ray.cpp": [
        ("#ifdef ENABLE_VK_IMPL\n#include \"internal/RendererVK.h\"\n#endif // ENABLE_VK_IMPL\n",
         "#ifdef ENABLE_VK_IMPL\n#include \"internal/RendererVK.h\"\n#endif // ENABLE_VK_IMPL\n#ifdef ENABLE_HIP_IMPL\n#include \"internal/RendererHIP.h\"\n#endif // ENABLE_HIP_IMPL\n"),
        ("                                       const Bitmask<eRendererType> enabled_types) {\n#if defined(ENABLE_VK_IMPL)\n",
         "                                       const Bitmask<eRendererType> enabled_types) {\n"
         "#if defined(ENABLE_HIP_IMPL)\n"
         "    if (enabled_types & eRendererType::HIP) {\n"
         "        log->Info(\"Ray: Creating HIP renderer %ix%i\", s.w, s.h);\n"
         "        try {\n"
         "            return Hip::CreateRenderer(s, log);\n"
         "        } catch (std::exception &e) {\n"
         "            log->Info(\"Ray: Failed to create HIP renderer, %s\", e.what());\n"
         "        }\n"
         "    }\n"
         "#endif // ENABLE_HIP_IMPL\n"
         "#if defined(ENABLE_VK_IMPL)\n"),
    ],
}

INTERNAL = ["AtmosphereRef", "BVHSplit", "CDFUtils", "Core", "CoreRef", "DenoiseRef", "FreelistAlloc", "PMJ", "RadCacheRef", "SamplingParams",
            "SceneCommon", "SceneCPU", "ShadeRef", "TextureParams", "TextureSplitter", "TextureStorageCPU", "TextureUtils", "TextureUtilsSSE2",
            "Time", "TonemapRef", "UNetFilter", "RendererRef"]
CXXFLAGS = ["-O3", "-DNDEBUG", "-std=c++17", "-msse2", "-mno-avx", "-fno-strict-aliasing", "-fPIC", "-w"]


def patch_tree(ref: str, tree: str) -> None:
    if os.path.isdir(tree):
        shutil.rmtree(tree)
    os.makedirs(tree)
    for name in os.listdir(ref):  # the library's own sources: root files + internal/ (+ the one third-party header Ray.cpp includes)
        p = os.path.join(ref, name)
        if os.path.isfile(p) and name.endswith((".h", ".cpp", ".inl")):
            shutil.copy(p, os.path.join(tree, name))
    shutil.copytree(os.path.join(ref, "internal"), os.path.join(tree, "internal"),
                    ignore=shutil.ignore_patterns("shaders", "Vk", "Dx", "*.glsl", "*.spv", "*.cso"))
    shutil.copytree(os.path.join(ref, "third-party", "renderdoc"), os.path.join(tree, "third-party", "renderdoc"))
    os.makedirs(os.path.join(tree, "samples", "00_basic"))
    shutil.copy(os.path.join(ref, "samples", "00_basic", "main.cpp"), os.path.join(tree, "samples", "00_basic", "main.cpp"))
    for name, edits in EDITS.items():
        path = os.path.join(tree, name)
        with open(path, encoding="utf-8") as f:
            text = f.read()
        for old, new in edits:
            if text.count(old) != 1:
                raise SystemExit(f"INTEGRATION patch: anchor not found exactly once in {name}:\n{old}")
            text = text.replace(old, new)
        with open(path, "w", encoding="utf-8") as f:
            f.write(text)
    shutil.copy(os.path.join(ROOT, "ray_amd", "host", "RendererHIP.h"), os.path.join(tree, "internal", "RendererHIP.h"))
    # Config.h + the three stand-ins for the blobs the tree lacks (tools/gen_ref_blobs.py), then the configuration of THIS build
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_ref_blobs
    gen_ref_blobs.main(ref, tree)
    with open(os.path.join(tree, "Config.h"), "w") as f:
        f.write("#pragma once\n\n#define ENABLE_REF_IMPL\n#define ENABLE_HIP_IMPL\n")


def build(ref: str) -> str:
    tree = os.path.join(OUT, "Ray")
    obj = os.path.join(OUT, "obj")
    patch_tree(ref, tree)
    os.makedirs(obj, exist_ok=True)
    inc = ["-I" + tree, "-I" + os.path.join(tree, "internal")]
    jobs = []
    for n in INTERNAL:
        jobs.append((os.path.join(tree, "internal", n + ".cpp"), os.path.join(obj, "internal_" + n + ".o"), []))
    for n in ("Ray", "RendererBase"):
        jobs.append((os.path.join(tree, n + ".cpp"), os.path.join(obj, "root_" + n + ".o"), []))
    jobs.append((os.path.join(tree, "internal", "simd", "detect.cpp"), os.path.join(obj, "simd_detect.o"), ["-mxsave"]))
    jobs.append((os.path.join(ROOT, "ray_amd", "host", "RendererHIP.cpp"), os.path.join(obj, "RendererHIP.o"), ["-I" + os.path.join(ROOT, "ray_amd", "host")]))
    jobs.append((os.path.join(tree, "samples", "00_basic", "main.cpp"), os.path.join(obj, "sample_00_basic.o"), []))

    def cc(job):
        src, dst, extra = job
        subprocess.run(["g++", *CXXFLAGS, *inc, *extra, "-c", src, "-o", dst], check=True)
        return dst

    with cf.ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, jobs))
    exe = os.path.join(OUT, "sample_00_basic")
    csrc_build = os.path.join(ROOT, "ray_amd", "csrc", "_build")
    subprocess.run(["g++", "-o", exe, *objs, "-L" + csrc_build, "-lrayhip", "-Wl,-rpath,$ORIGIN/../../../csrc/_build", "-lpthread"], check=True)
    return exe


if __name__ == "__main__":
    print(build(sys.argv[1] if len(sys.argv) > 1 else "/root/reference"))
