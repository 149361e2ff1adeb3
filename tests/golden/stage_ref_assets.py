"""Stages the reference's own test assets for tests/test_material_matrix.py -- the mat_test meshes and those textures of its material tests
that the checkout has -- into tests/assets/_ref/ (git-ignored: nothing of it enters the history; NOT gpurun-ignored: it travels to the GPU box
like oracle/_ref, because /root/reference does not exist there).

    python tests/golden/stage_ref_assets.py            # needs /root/reference; __graft_entry__.build() runs it when that exists

Meshes (tests/test_data/meshes/mat_test/*.bin) and .dds files are byte copies of DATA files.  .tga files are decoded here once (run-length
packets, tests/utils.cpp:116-159 + internal/TextureUtils.cpp:1753-1860 describe the layout the reference's loader produces) into the rows
LoadTGA(flip_y = true) hands to AddTexture and stored as compressed .npz: a 2048 x 2048 run-length image takes seconds to decode in Python."""
import json
import os
import shutil
import struct
import sys

import numpy as np

REF = "/root/reference/tests/test_data"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "assets", "_ref")
MATRIX = os.path.join(ROOT, "tests", "golden", "material_matrix.json")


def decode_tga(path):
    """-> [h, w, 3] u8 RGB in the row order of the FILE.  (ReadTGAFile turns a bottom-up file top-down, LoadTGA(flip_y = true) turns it back:
    what reaches AddTexture is the file's own row order.)"""
    with open(path, "rb") as f:
        data = f.read()
    id_len, cmap, kind = data[0], data[1], data[2]
    w, h, bpp, desc = struct.unpack_from("<HHBB", data, 12)
    assert cmap == 0 and kind in (2, 10) and bpp in (24, 32), (path, kind, bpp)
    top_down = (desc & 0x20) != 0
    px = bpp // 8
    body = np.frombuffer(data, dtype=np.uint8, offset=18 + id_len)
    n = w * h
    if kind == 2:
        out = body[:n * px].reshape(n, px)
    else:
        out = np.empty((n, px), dtype=np.uint8)
        pos = filled = 0
        while filled < n:
            head = int(body[pos])
            count = (head & 0x7F) + 1
            pos += 1
            if head & 0x80:
                out[filled:filled + count] = body[pos:pos + px]
                pos += px
            else:
                out[filled:filled + count] = body[pos:pos + count * px].reshape(count, px)
                pos += count * px
            filled += count
    rgb = out[:, [2, 1, 0]].reshape(h, w, 3)
    # file order bottom-up -> the reference's two flips cancel; a top-down file would be flipped once by LoadTGA
    return np.ascontiguousarray(rgb[::-1] if top_down else rgb)


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (this container); on the GPU box the staged files arrive with the snapshot")
    os.makedirs(os.path.join(OUT, "meshes"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "textures"), exist_ok=True)
    staged = []
    for name in sorted(os.listdir(os.path.join(REF, "meshes", "mat_test"))):
        dst = os.path.join(OUT, "meshes", name)
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(REF, "meshes", "mat_test", name), dst)
        staged.append("meshes/" + name)
    with open(MATRIX) as f:
        wanted = sorted({t for e in json.load(f)["tests"] for t in e["textures"]})
    absent = []
    for name in wanted:
        src = os.path.join(REF, "textures", name)
        if not os.path.exists(src):
            absent.append(name)
            continue
        if name.endswith(".dds"):
            dst = os.path.join(OUT, "textures", name)
            if not os.path.exists(dst):
                shutil.copyfile(src, dst)
        else:
            dst = os.path.join(OUT, "textures", name + ".npz")
            if not os.path.exists(dst):
                np.savez_compressed(dst, rgb=decode_tga(src))
        staged.append("textures/" + name)
    print(f"staged {len(staged)} files under {OUT}; absent from the checkout (procedural stand-ins in the tests): {absent}")


if __name__ == "__main__":
    main()
