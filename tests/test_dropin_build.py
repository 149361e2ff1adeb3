"""The drop-in compiled for real (ray_amd/host/dropin/build_dropin.py): the reference tree with the INTEGRATION.md registration
patch applied and its own samples/00_basic/main.cpp, unchanged, linked against librayhip.so.  Here (no GPU): the patch must
apply to the reference as it is -- every edit anchored on the text it replaces -- and the built sample must do what the
factory convention says on a box without a device: try HIP, log the failure, fall back to the Reference renderer."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RAY_REFERENCE_DIR", "/root/reference")
EXE = os.path.join(ROOT, "ray_amd", "host", "_build", "dropin", "sample_00_basic")
sys.path.insert(0, os.path.join(ROOT, "ray_amd", "host", "dropin"))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "internal")), reason="needs the reference tree")
def test_the_integration_patch_applies_to_the_reference(tmp_path):
    import build_dropin
    tree = os.path.join(str(tmp_path), "Ray")
    build_dropin.patch_tree(REF, tree)
    text = {n: open(os.path.join(tree, n)).read() for n in ("RendererBase.h", "RendererBase.cpp", "Ray.h", "Ray.cpp", "Config.h")}
    assert "HIP // AMD Instinct" in text["RendererBase.h"] and "| eRendererType::HIP;" in text["RendererBase.h"]
    assert 'return "HIP";' in text["RendererBase.cpp"] and 'name == "HIP"' in text["RendererBase.cpp"]
    assert "eRendererType::HIP;" in text["Ray.h"]
    assert "Hip::CreateRenderer(s, log)" in text["Ray.cpp"] and '#include "internal/RendererHIP.h"' in text["Ray.cpp"]
    assert text["Ray.cpp"].index("ENABLE_HIP_IMPL)") < text["Ray.cpp"].index("if (enabled_types & eRendererType::Vulkan)")
    assert "ENABLE_HIP_IMPL" in text["Config.h"]
    # the sample is the reference's file, byte for byte
    with open(os.path.join(tree, "samples", "00_basic", "main.cpp"), "rb") as a, open(os.path.join(REF, "samples", "00_basic", "main.cpp"), "rb") as b:
        assert a.read() == b.read()
    # a drifted anchor is an error, not a silent no-op
    build_dropin.EDITS["Ray.h"].append(("this text is not in Ray.h", "x"))
    try:
        with pytest.raises(SystemExit):
            build_dropin.patch_tree(REF, os.path.join(str(tmp_path), "Ray2"))
    finally:
        build_dropin.EDITS["Ray.h"].pop()


def test_the_unchanged_sample_falls_back_without_a_device(tmp_path):
    import torch
    if not os.path.exists(EXE):
        pytest.skip("the drop-in sample is not built (needs the reference tree at build time)")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_dropin.py runs the sample on it")
    r = subprocess.run([EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "Ray: Creating HIP renderer 256x256" in out
    assert "Ray: Failed to create HIP renderer, no HIP device" in out
    assert "Ray: Creating Reference renderer 256x256" in out
    assert os.path.getsize(os.path.join(str(tmp_path), "00_basic.tga")) == 18 + 256 * 256 * 3 + 26  # header, BGR8 pixels, footer
