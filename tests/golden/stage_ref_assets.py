"""Stages the reference's own test assets for tests/test_material_matrix.py -- the mat_test meshes and those textures of its material tests
that the checkout has -- into tests/assets/_ref/ (git-ignored: nothing of it enters the history; NOT gpurun-ignored: it travels to the GPU box
like oracle/_ref, because /root/reference does not exist there).

    python tests/golden/stage_ref_assets.py            # needs /root/reference; __graft_entry__.build() runs it when that exists

Everything is DATA, parsed here once and stored as the arrays the tests hand to the API, in compressed .npz containers (smaller to ship, and
nothing to parse on the GPU box): meshes (tests/test_data/meshes/mat_test/*.bin, layout of tests/utils.cpp:72-114) -> meshes.npz with
<name>.attrs / .indices / .groups; .dds -> the block data behind the 128-byte header + (w, h, mips, channels) (tests/utils.cpp:161-201); .tga
-> RGB rows in the order LoadTGA(flip_y = true) hands to AddTexture (run-length packets, tests/utils.cpp:116-159 +
internal/TextureUtils.cpp:1753-1860: a 2048 x 2048 run-length image takes seconds to decode in Python)."""
import json
import os
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
    os.makedirs(os.path.join(OUT, "textures"), exist_ok=True)
    staged = []
    meshes = {}
    for name in sorted(os.listdir(os.path.join(REF, "meshes", "mat_test"))):
        with open(os.path.join(REF, "meshes", "mat_test", name), "rb") as f:
            data = f.read()
        n_attrs, n_idx, n_groups = struct.unpack_from("<III", data, 0)
        stem = name[:-4]
        meshes[stem + ".attrs"] = np.frombuffer(data, dtype=np.float32, count=n_attrs, offset=12)
        meshes[stem + ".indices"] = np.frombuffer(data, dtype=np.uint32, count=n_idx, offset=12 + 4 * n_attrs)
        meshes[stem + ".groups"] = np.frombuffer(data, dtype=np.uint32, count=n_groups, offset=12 + 4 * n_attrs + 4 * n_idx)
        staged.append("meshes/" + name)
    if not os.path.exists(os.path.join(OUT, "meshes.npz")):
        np.savez_compressed(os.path.join(OUT, "meshes.npz"), **meshes)
    with open(MATRIX) as f:
        wanted = sorted({t for e in json.load(f)["tests"] for t in e["textures"]})
    absent = []
    for name in wanted:
        src = os.path.join(REF, "textures", name)
        if not os.path.exists(src):
            absent.append(name)
            continue
        if name.endswith(".dds"):
            dst = os.path.join(OUT, "textures", name + ".npz")
            if not os.path.exists(dst):
                with open(src, "rb") as f:
                    data = f.read()
                h, w = struct.unpack_from("<II", data, 12)
                mips = struct.unpack_from("<I", data, 28)[0]
                channels = {b"DXT1": 3, b"DXT5": 4, b"ATI1": 1, b"BC4U": 1, b"ATI2": 2}[data[84:88]]
                np.savez_compressed(dst, blocks=np.frombuffer(data, dtype=np.uint8, offset=128), shape=np.array([w, h, max(1, mips), channels]))
        else:
            dst = os.path.join(OUT, "textures", name + ".npz")
            if not os.path.exists(dst):
                np.savez_compressed(dst, rgb=decode_tga(src))
        staged.append("textures/" + name)
    print(f"staged {len(staged)} files under {OUT}; absent from the checkout (procedural stand-ins in the tests): {absent}")


if __name__ == "__main__":
    main()
