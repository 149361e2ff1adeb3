"""A scene that does not hold together must be refused at the boundary, with a message, before any kernel follows an index
out of its array (ray_amd/csrc/scene_blob.h: section bounds, alignment, element sizes; ray_amd/csrc/scene_validate.h: every
index a kernel dereferences without a bound of its own, by reachability from the top-level tree).  The checks are shared
source between librayhip and the host build, so they are exercised here without a GPU."""
import struct

import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import hip

pytestmark = pytest.mark.skipif(not O.have_hostsim(), reason="tests/hostsim not built")

HEADER, SECTION = 16, 40  # scene_blob.h: Header {magic[8], count, pad}, Section {name[24], offset u64, size u64}


def sections(blob: bytes) -> dict:
    count = struct.unpack_from("<I", blob, 8)[0]
    out = {}
    for i in range(count):
        name, off, size = struct.unpack_from("<24sQQ", blob, HEADER + i * SECTION)
        out[name.rstrip(b"\0").decode()] = (i, off, size)
    return out


def upload(blob: bytes):
    ctx = hip.Context(0, hip.Library(O.HOSTSIM_LIB, prefix="hostsim_"))
    ctx.upload_static(util.pmj())
    ctx.resize(32, 32)
    ctx.upload_scene_blob(blob)
    return ctx


def patched(blob: bytes, section: str, byte_offset: int, fmt: str, value) -> bytes:
    _, off, size = sections(blob)[section]
    assert byte_offset + struct.calcsize(fmt) <= size
    b = bytearray(blob)
    struct.pack_into(fmt, b, off + byte_offset, value)
    return bytes(b)


def first_leaf_entry(blob: bytes) -> int:
    """index into tris[] / tri_indices[] of a triangle entry that a bottom-level leaf reaches (the arrays are sparse pools)"""
    _, off, size = sections(blob)["nodes"]
    nodes = np.frombuffer(blob, dtype=np.uint32, count=size // 4, offset=off).reshape(-1, 16)
    _, moff, msize = sections(blob)["mesh_instances"]
    node_index = struct.unpack_from("<I", blob, moff + 4)[0]
    stack = [int(node_index)]
    while stack:
        w = stack.pop()
        if w & (7 << 29):
            return w & ~(7 << 29)
        stack += [int(nodes[w, 12]), int(nodes[w, 13])]
    raise AssertionError("no leaf")


def test_the_untouched_scene_uploads():
    upload(util.golden_scene("cornell_lights")).render(1)


@pytest.mark.parametrize("what", ["tri_indices", "vtx_indices", "node_link", "leaf_range", "li_indices", "light_tree_link", "material",
                                  "instance_root", "tlas_root", "env_light"])
def test_an_index_out_of_its_array_is_refused(what):
    blob = util.golden_scene("cornell_lights")
    s = sections(blob)
    if what == "tri_indices":
        bad = patched(blob, "tri_indices", 4 * first_leaf_entry(blob), "<I", 0x0fffffff)
    elif what == "vtx_indices":
        _, off, _ = s["tri_indices"]
        tri = struct.unpack_from("<I", blob, off + 4 * first_leaf_entry(blob))[0]
        bad = patched(blob, "vtx_indices", 12 * tri, "<I", 0x0fffffff)
    elif what == "node_link":
        _, moff, _ = s["mesh_instances"]
        root = struct.unpack_from("<I", blob, moff + 4)[0]
        bad = patched(blob, "nodes", 64 * root + 48, "<I", 0x00ffffff)  # left_child of a bottom-level root: far outside
    elif what == "leaf_range":
        _, moff, _ = s["mesh_instances"]
        root = struct.unpack_from("<I", blob, moff + 4)[0]
        bad = patched(blob, "nodes", 64 * root + 48, "<I", (7 << 29) | 0x0ffffff0)  # a leaf of 8 entries beyond tris[]
    elif what == "li_indices":
        bad = patched(blob, "li_indices", 0, "<I", 0x00ffffff)
    elif what == "light_tree_link":
        bad = patched(blob, "light_cwnodes", 80, "<I", 0x00ffffff)  # child[0] of the root (after two boxes and 48 quantised bytes): no such node
    elif what == "material":
        _, off, _ = s["tri_indices"]
        tri = struct.unpack_from("<I", blob, off + 4 * first_leaf_entry(blob))[0]
        bad = patched(blob, "tri_materials", 4 * tri, "<H", 0x3fff)  # front material index 16383
    elif what == "instance_root":
        bad = patched(blob, "mesh_instances", 4, "<I", 0x00ffffff)
    elif what == "tlas_root":
        bad = patched(blob, "scalars", 32 + 64, "<I", 0x00ffffff)  # Scalars: tex_table[8], environment (64 B), tlas_root
    else:
        bad = patched(blob, "scalars", 32 + 40, "<I", 0x00ffffff)  # environment.light_index
    with pytest.raises(RuntimeError) as e:
        upload(bad)
    print(what, "->", e.value)
    assert len(str(e.value)) > 20


def test_malformed_containers_are_refused():
    blob = util.golden_scene("cornell_basic")
    s = sections(blob)
    cases = {}
    i, off, size = s["tris"]
    b = bytearray(blob)
    struct.pack_into("<QQ", b, HEADER + i * SECTION + 24, 2 ** 64 - 8, 16)  # offset + size wraps around
    cases["wrapping section"] = bytes(b)
    b = bytearray(blob)
    struct.pack_into("<QQ", b, HEADER + i * SECTION + 24, off + 4, size - 16)  # not 16-byte aligned
    cases["misaligned section"] = bytes(b)
    b = bytearray(blob)
    struct.pack_into("<QQ", b, HEADER + i * SECTION + 24, off, size - 8)  # not a whole number of 48-byte records
    cases["ragged section"] = bytes(b)
    b = bytearray(blob)
    struct.pack_into("<QQ", b, HEADER + i * SECTION + 24, off, len(blob))  # runs past the end
    cases["section past the end"] = bytes(b)
    cases["truncated"] = blob[:HEADER + 3 * SECTION]
    cases["bad magic"] = b"NOTASCENE" + blob[9:]
    b = bytearray(blob)
    struct.pack_into("<I", b, 8, 0x7fffffff)  # section count
    cases["absurd section count"] = bytes(b)
    for name, bad in cases.items():
        with pytest.raises(RuntimeError) as e:
            upload(bad)
        print(name, "->", e.value)


def test_a_texture_outside_the_texel_pool_is_refused():
    blob = util.golden_scene("cornell_principled")
    _, off, size = sections(blob)["textures"]
    assert size >= 144  # rayhip_texture: width[12], height[12], offset[12]
    bad = patched(blob, "textures", 96, "<I", 0x7ffffff0)  # offset[0] of the first texture
    with pytest.raises(RuntimeError):
        upload(bad)


def test_a_texture_handle_beyond_the_eight_storages_is_refused():
    """handle >> 28 selects one of EIGHT storages (tex_table[8]); values 8..15 would index behind the table -- on the host into
    the fields that follow it in the desc, on the device into other fields of the scene view, so a sum that happens to pass on
    the host bounds nothing on the device.  Every material whose slot the shade stage reads is patched (the pool is sparse)."""
    blob = util.golden_scene("cornell_principled")
    _, off, size = sections(blob)["materials"]
    hit = 0
    bad = bytearray(blob)
    for m in range(size // 76):
        for k in range(5):
            h = struct.unpack_from("<I", blob, off + 76 * m + 4 * k)[0]
            if h != 0xffffffff and (h >> 28) < 8 and struct.unpack_from("<I", blob, off + 76 * m + 36)[0] != 4:  # not a mix node
                struct.pack_into("<I", bad, off + 76 * m + 4 * k, (h & 0x0fffffff) | (0xb << 28))
                hit += 1
    assert hit > 0
    with pytest.raises(RuntimeError) as e:
        upload(bytes(bad))
    print(e.value)
    assert "texture" in str(e.value)
    # and the environment map handles (rayhip_environment: env_col[3], env_map, back_col[3], back_map, ...)
    for word in (3, 7):
        with pytest.raises(RuntimeError) as e:
            upload(patched(blob, "scalars", 32 + 4 * word, "<I", 0x90000000))
        assert "environment map" in str(e.value)


def test_a_mesh_tree_that_links_into_the_top_level_is_refused():
    """an instance whose BLAS root IS the top-level root: the device would read instance leaves as triangle ranges"""
    blob = util.golden_scene("cornell_instances")
    s = sections(blob)
    tlas_root = struct.unpack_from("<I", blob, s["scalars"][1] + 32 + 64)[0]
    _, off, size = s["nodes"]
    nodes = np.frombuffer(blob, dtype=np.uint32, count=size // 4, offset=off).reshape(-1, 16)
    if int(nodes[tlas_root, 12]) & (7 << 29) and int(nodes[tlas_root, 13]) & (7 << 29):
        pytest.skip("the top level is a single node of two leaves")
    # first live instance (reachable from the top level)
    stack, mi = [int(tlas_root)], None
    while stack and mi is None:
        w = stack.pop()
        if w & (7 << 29):
            mi = w & ~(7 << 29)
        else:
            stack += [int(nodes[w, 12]), int(nodes[w, 13])]
    bad = patched(blob, "mesh_instances", 144 * mi + 4, "<I", tlas_root)
    with pytest.raises(RuntimeError) as e:
        upload(bad)
    print(e.value)
    assert "top-level" in str(e.value) or "not a tree" in str(e.value)


def test_a_physical_sky_must_bring_its_tables():
    """sky_map_spread_angle > 0 without rayhip_scene_desc::sky, or with a texture whose size does not match the stated dimensions"""
    blob = util.golden_scene("cornell_sky")
    upload(blob).render(1)
    secs = sections(blob)
    i, _, _ = secs["sky"]
    b = bytearray(blob)
    b[HEADER + i * SECTION: HEADER + i * SECTION + 3] = b"xky"
    with pytest.raises(Exception, match="physical sky"):
        upload(bytes(b))
    _, off, _ = secs["sky"]
    with pytest.raises(Exception, match="sky"):
        upload(patched(blob, "sky", 208 + 12, "<i", 100))  # weather_res: not a power of two, and not the size of the section
    with pytest.raises(Exception, match="directional"):
        upload(patched(blob, "sky_dir_lights", 0, "<I", 0x00ffffff))
