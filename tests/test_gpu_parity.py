"""Parity of the HIP path (through the librayhip C ABI) against RendererRef.  Needs a real MI355X.

Every test goes through include/rayhip.h entry points (ctypes, ray_amd/hip.py); the checker is
  * the committed golden vectors in tests/golden/ (dumped from the real reference by make_fixtures.py), and
  * tests/hostsim (the kernel sources compiled for the host, itself bit-exact against RendererRef on the CPU --
    tests/test_hostsim_parity.py), which isolates what is specific to the device: libm, LDS stack, wave-level
    compaction, SoA memory layout.
Bar: integer/index results exact; fp32 images within the stated tolerance (tests/util.py).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import util
from ray_amd import hip

pytestmark = pytest.mark.gpu

SCENES = ["cornell_basic", "cornell_principled", "cornell_lights", "cornell_env", "cornell_filmic", "cornell_instances"]


@pytest.fixture(scope="module")
def gpu_lib():
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path, -m gpu tests cannot run here"
    return lib


@pytest.fixture(scope="module")
def hostsim_lib():
    if not O.have_hostsim():
        pytest.skip("tests/hostsim not built")
    return hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")


def test_device_is_gfx950(gpu_lib):
    ctx = hip.Context(0, gpu_lib)
    name = ctx.device_name()
    print(name)
    assert "gfx950" in name


def test_rng_is_bit_exact(gpu_lib):
    """integer Owen-scrambled PMJ02 stream: exact u32 -> identical floats"""
    v = np.load(f"{util.GOLDEN}/rng_vectors.npz")
    ctx = hip.Context(0, gpu_lib)
    ctx.upload_static(util.pmj())
    out = ctx.k_scrambled_rand(v["dims"], v["seeds"], v["samples"])
    assert np.array_equal(out.view(np.uint32), v["xy"].view(np.uint32))


@pytest.mark.parametrize("name", SCENES)
def test_primary_rays(gpu_lib, name):
    """K1 vs Ref::GeneratePrimaryRays: pixel set and integer fields exact, origins/directions within 2 ulp-ish"""
    g = util.golden_ref(name)
    ctx = util.make_context(gpu_lib, name)
    rays, hits = ctx.k_generate_primary_rays(1)
    ref = util.sort_by_xy(g["primary_rays"])
    order = np.argsort(rays["xy"], kind="stable")
    rays, hits = rays[order], hits[order]
    assert np.array_equal(rays["xy"], ref["xy"]) and np.array_equal(rays["depth"], ref["depth"])
    for f in ("o", "d", "c", "ior"):
        np.testing.assert_allclose(rays[f], ref[f], rtol=0, atol=2e-7)
    assert np.array_equal(rays["pdf"], ref["pdf"]) and np.array_equal(rays["cone_spread"], ref["cone_spread"])
    ref_h = g["primary_hits_in"][np.argsort(g["primary_rays"]["xy"], kind="stable")]
    np.testing.assert_allclose(hits["t"], ref_h["t"], rtol=1e-6)


@pytest.mark.parametrize("name", SCENES)
def test_closest_hit_on_reference_rays(gpu_lib, name):
    """K2 on the reference's own rays: exact (obj_index, prim_index); |dt|,|du|,|dv| <= 1e-5 rel; throughput/depth of
    rays that crossed transparent surfaces as the reference left them"""
    g = util.golden_ref(name)
    ctx = util.make_context(gpu_lib, name)
    rays, hits, tc = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1)
    # the kernel rayhip_render launches (the 4-wide quantised BLAS of rt_bvh4.h by default) must find the very same hits
    rays_w, hits_w, _ = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=0)
    assert hits_w.tobytes() == hits.tobytes() and rays_w.tobytes() == rays.tobytes()
    ref = g["primary_hits"]
    hit = ref["v"] >= 0
    assert np.array_equal(hits["obj_index"], ref["obj_index"])
    assert np.array_equal(hits["prim_index"][hit], ref["prim_index"][hit])  # (misses: see util.assert_hits_identical)
    for f in ("t", "u", "v"):
        np.testing.assert_allclose(hits[f][hit], ref[f][hit], rtol=1e-5, atol=1e-6)
    assert tc["rays"] == len(rays) and tc["nodes"] > 0 and tc["tris"] > 0


@pytest.mark.parametrize("name", SCENES)
def test_shadow_rays_on_reference_rays(gpu_lib, name):
    """K3 on the reference's own shadow rays: visibility identical"""
    g = util.golden_ref(name)
    ctx = util.make_context(gpu_lib, name)
    rc, tc = ctx.k_intersect_shadow(g["shadow_rays"], 1)
    np.testing.assert_allclose(rc, g["shadow_rc"], rtol=1e-6, atol=0)
    assert tc["rays"] == len(g["shadow_rays"])


@pytest.mark.parametrize("name", SCENES)
def test_traversal_work_counters_match_host(gpu_lib, hostsim_lib, name, monkeypatch):
    """the visit counts that feed the algorithmic-bytes formula are the same on device and in the host build"""
    monkeypatch.setenv("HOSTSIM_REFINE", "2")     # the trees librayhip walks: leaves refined to <= 2 triangles (scene_rebuild.h),
    monkeypatch.setenv("HOSTSIM_NO_LAYOUT", "1")  # in the order the refinement leaves them (no layout pass since round 3),
    monkeypatch.setenv("HOSTSIM_BVH4", "1")       # collapsed four wide (on the device there: bvh4_build.hip.h, here the host driver)
    g = util.golden_ref(name)
    gpu = util.make_context(gpu_lib, name)
    host = util.make_context(hostsim_lib, name)
    _, hg, tc_g = gpu.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1)
    _, hh, tc_h = host.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1)
    assert tc_g == tc_h
    assert np.array_equal(hg["prim_index"], hh["prim_index"])  # same triangle order on both sides: equal even for misses
    # the product walk (4-wide, its tree built by the device driver of the collapse) with counters: node visits and triangle
    # tests, device against the host build (the host driver of the collapse: the same tree in another node order)
    _, hgw, tw_g = gpu.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=hip.FLAG_COUNT_WIDE)
    _, hhw, tw_h = host.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=hip.FLAG_COUNT_WIDE)
    assert tw_g == tw_h and tw_g["nodes4"] > 0
    assert hgw.tobytes() == hg.tobytes()
    # the shadow-ray hook walks the reference's BVH2 on the device; so does the host build when no wide form is selected
    monkeypatch.setenv("HOSTSIM_BVH4", "0")
    host2 = util.make_context(hostsim_lib, name)
    _, sc_g = gpu.k_intersect_shadow(g["shadow_rays"], 1)
    _, sc_h = host2.k_intersect_shadow(g["shadow_rays"], 1)
    assert sc_g == sc_h


@pytest.mark.parametrize("name", SCENES)
def test_frame_vs_reference(gpu_lib, name):
    """whole RenderScene loop vs RendererRef raw buffer at 1 and 8 spp (golden fixtures)"""
    g = util.golden_ref(name)
    ctx = util.make_context(gpu_lib, name)
    ctx.render(1)
    m1 = util.frame_metrics(ctx.readback(hip.BUF_RAW), g["raw_spp1"])
    for it in range(2, 9):
        ctx.render(it)
    m8 = util.frame_metrics(ctx.readback(hip.BUF_RAW), g["raw_spp8"])
    print(name, "1spp", m1, "8spp", m8)
    assert m1["frac_within"] >= util.MIN_FRACTION and m1["psnr"] >= util.MIN_PSNR_1SPP and m1["alpha_equal"]
    assert m8["frac_within"] >= util.MIN_FRACTION and m8["psnr"] >= util.MIN_PSNR_8SPP and m8["alpha_equal"]
    # tonemapped + aux buffers
    mf = util.frame_metrics(ctx.readback(hip.BUF_FINAL), g["final_spp8"])
    assert mf["frac_within"] >= util.MIN_FRACTION
    np.testing.assert_allclose(ctx.readback(hip.BUF_BASE_COLOR), g["base_color_spp8"], atol=2e-3)
    dn = ctx.readback(hip.BUF_DEPTH_NORMALS)
    assert (np.abs(dn - g["depth_normals_spp8"]).max(axis=-1) <= 2e-3).mean() >= util.MIN_FRACTION


@pytest.mark.parametrize("name", SCENES)
def test_frame_vs_hostsim_larger(gpu_lib, hostsim_lib, name):
    """256x256, 4 spp: device vs the host build of the same kernel source (bit-exact vs RendererRef on the CPU)"""
    w = h = 256
    gpu = util.make_context(gpu_lib, name, w, h)
    host = util.make_context(hostsim_lib, name, w, h)
    a = util.render_frames(gpu, 4)
    b = util.render_frames(host, 4)
    m = util.frame_metrics(a, b)
    print(name, m)
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_1SPP


def test_instrumented_render_matches_plain(gpu_lib):
    """RAYHIP_FLAG_COUNT_TRAVERSAL must not change the image; counters equal the host build's"""
    name = "cornell_basic"
    a = util.make_context(gpu_lib, name)
    b = util.make_context(gpu_lib, name)
    ia = util.render_frames(a, 2)
    ib = util.render_frames(b, 2, flags=hip.FLAG_COUNT_TRAVERSAL)
    assert np.array_equal(ia, ib)
    c0, c1 = b.trav_counters()
    assert c0["rays"] > 64 * 64 and c1["rays"] > 0 and c0["nodes"] > c0["rays"]


def test_render_is_deterministic(gpu_lib):
    """wave-level compaction reorders rays between runs; per-pixel results must not change"""
    name = "cornell_principled"
    imgs = [util.render_frames(util.make_context(gpu_lib, name), 3) for _ in range(2)]
    assert np.array_equal(imgs[0], imgs[1])


def test_tile_sharding_is_bit_identical(gpu_lib):
    """multi-GPU decomposition on one GPU: sum of the shards' RAW buffers == unsharded render, exactly"""
    name = "cornell_basic"
    w = h = 128
    full = util.render_frames(util.make_context(gpu_lib, name, w, h), 3)
    acc = np.zeros_like(full)
    n = 3
    for r in range(n):
        ctx = util.make_context(gpu_lib, name, w, h)
        ctx.set_shard(32, n, r)
        part = util.render_frames(ctx, 3)
        acc += part
    assert np.array_equal(acc, full)


@pytest.mark.parametrize("w,h,tile,n", [(100, 72, 32, 3), (100, 72, 64, 2), (65, 65, 64, 2), (96, 80, 12, 2), (160, 96, 8, 5)])
def test_sharded_batches_walk_owned_tiles_only(gpu_lib, w, h, tile, n):
    """a rank of a tile-sharded render generates rays (and sizes its wavefront state) for its own shard tiles only,
    frames that are not whole tiles and tile sizes that are not whole 8x8 ray-generation tiles included; batched, with a
    rect, the shards must still add up to the unsharded frame bit for bit"""
    name = "cornell_basic"
    one = util.make_context(gpu_lib, name, w, h)
    for it in range(1, 6):
        one.render(it)
    one.render(6, rect=(8, 16, w - 20, h - 24))
    full = one.readback(hip.BUF_RAW)
    acc = np.zeros_like(full)
    for r in range(n):
        ctx = util.make_context(gpu_lib, name, w, h)
        ctx.set_shard(tile, n, r)
        ctx.render_batch(1, 5)
        ctx.render_batch(6, 1, rect=(8, 16, w - 20, h - 24))
        acc += ctx.readback(hip.BUF_RAW)
    assert np.array_equal(acc, full)


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_instances"])
def test_ray_sort_is_bit_identical(gpu_lib, name):
    """SURVEY a10 (SortRays: key = direction octant | Morton code of the origin cell, radix sort, gather): opt-in through
    RAYHIP_FLAG_SORT_RAYS; ray order must not change a pixel -- full frame, a rect, and on a rank of a sharded frame"""
    w, h = 96, 72
    plain = util.make_context(gpu_lib, name, w, h)
    sorted_ = util.make_context(gpu_lib, name, w, h)
    for it in range(1, 5):
        plain.render(it)
        sorted_.render(it, flags=hip.FLAG_SORT_RAYS)
    plain.render(5, rect=(8, 8, 64, 40))
    sorted_.render(5, rect=(8, 8, 64, 40), flags=hip.FLAG_SORT_RAYS)
    for buf in (hip.BUF_RAW, hip.BUF_FINAL, hip.BUF_BASE_COLOR, hip.BUF_DEPTH_NORMALS):
        assert np.array_equal(plain.readback(buf), sorted_.readback(buf)), buf
    a = util.make_context(gpu_lib, name, w, h)
    b = util.make_context(gpu_lib, name, w, h)
    a.set_shard(32, 3, 1), b.set_shard(32, 3, 1)
    for it in range(1, 4):
        a.render(it)
        b.render(it, flags=hip.FLAG_SORT_RAYS)
    assert np.array_equal(a.readback(hip.BUF_RAW), b.readback(hip.BUF_RAW))
    b.render_batch(4, 3, flags=hip.FLAG_SORT_RAYS)  # (a batch with the sort flag falls back to one pass per iteration)
    for it in range(4, 7):
        a.render(it)
    assert np.array_equal(a.readback(hip.BUF_RAW), b.readback(hip.BUF_RAW))


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (65, 1), (13, 9)])
def test_tiny_and_ragged_frames(gpu_lib, hostsim_lib, w, h):
    """frames smaller than one 8x8 ray-generation tile / one wave, and ragged ones, stacked into one pass: GPU vs the host
    build (which equals the reference on these sizes, tests/test_hostsim_parity.py::test_live_reference_other_sizes)"""
    spp = 5
    host = util.render_frames(util.make_context(hostsim_lib, "cornell_basic", w, h), spp)
    ctx = util.make_context(gpu_lib, "cornell_basic", w, h)
    ctx.render_batch(1, spp)
    m = util.frame_metrics(ctx.readback(hip.BUF_RAW), host)
    assert m["frac_within"] == 1.0 and m["alpha_equal"], m


def test_error_paths_are_loud(gpu_lib):
    """misuse and unsupported content return an error (rayhip_last_error) instead of rendering something else"""
    import copy
    ctx = hip.Context(0, gpu_lib)
    with pytest.raises(RuntimeError, match="resize"):
        ctx.render_batch(1, 2, rect=(0, 0, 8, 8), cam=hip.Camera())  # nothing set up
    ctx = util.make_context(gpu_lib, "cornell_basic")
    with pytest.raises(RuntimeError, match="rect"):
        ctx.render(1, rect=(32, 32, 64, 64))
    with pytest.raises(RuntimeError, match="1-based"):
        ctx.render(0)
    cam = copy.copy(ctx.cam)
    cam.view_transform = 8  # Filmic_HighContrast without its table
    with pytest.raises(RuntimeError, match="look-up table"):
        ctx.render(1, cam=cam)
    with pytest.raises(RuntimeError, match="look-up table"):
        ctx.denoise_nlm(1, cam=cam)
    cam = copy.copy(ctx.cam)
    cam.type = 1  # eCamType::Ortho
    with pytest.raises(RuntimeError, match="perspective"):
        ctx.render(1, cam=cam)
    with pytest.raises(RuntimeError, match="bad frame size"):
        ctx.resize(0, 10)
    with pytest.raises(RuntimeError, match="bad shard"):
        ctx.set_shard(64, 2, 2)
    ctx.resize(64, 64)
    ctx.render(1)  # still usable after the errors
    assert float(ctx.readback(hip.BUF_RAW)[..., :3].max()) > 0.0


def test_rect_render(gpu_lib):
    """RegionContext rect: rendering two half-frame rects == rendering the full frame"""
    name = "cornell_basic"
    full = util.render_frames(util.make_context(gpu_lib, name), 2)
    ctx = util.make_context(gpu_lib, name)
    for it in (1, 2):
        ctx.render(it, rect=(0, 0, 64, 32))
        ctx.render(it, rect=(0, 32, 64, 32))
    assert np.array_equal(ctx.readback(hip.BUF_RAW), full)


def test_stats_and_timing(gpu_lib):
    ctx = util.make_context(gpu_lib, "cornell_basic")
    st = hip.Stats()
    ctx.render(1, stats=st)
    d = st.as_dict()
    assert d["primary_trace"] >= 0 and sum(d.values()) > 0
    (ms0, n0), (ms1, n1) = ctx.trav_timing()
    assert n0 >= 2 and n1 >= 1 and ms0 > 0.0


def test_full_size_properties(gpu_lib):
    """BASELINE-size frame (1920x1080): properties that do not need a reference image --
    tile sharding sums to the unsharded frame bit for bit, two half-frame rects equal the full frame, the run is
    deterministic, and the frame agrees statistically with the 64x64 fixture of the same scene (same camera)."""
    name = "cornell_basic"
    w, h, spp = 1920, 1080, 2
    full = util.render_frames(util.make_context(gpu_lib, name, w, h), spp)
    again = util.render_frames(util.make_context(gpu_lib, name, w, h), spp)
    assert np.array_equal(full, again)
    acc = np.zeros_like(full)
    for r in range(4):
        ctx = util.make_context(gpu_lib, name, w, h)
        ctx.set_shard(64, 4, r)
        acc += util.render_frames(ctx, spp)
    assert np.array_equal(acc, full)
    ctx = util.make_context(gpu_lib, name, w, h)
    for it in range(1, spp + 1):
        ctx.render(it, rect=(0, 0, w, 500))
        ctx.render(it, rect=(0, 500, w, h - 500))
    assert np.array_equal(ctx.readback(hip.BUF_RAW), full)
    assert np.isfinite(full).all() and (full[..., :3] >= 0).all()
    # wide-aspect frame of the same camera: the central square shows what the 64x64 fixture shows
    g = util.golden_ref(name)
    centre = full[:, 420:1500]  # the 1080x1080 square the fixture's field of view covers
    assert float(centre[..., 3].mean()) > 0.9  # the box fills (almost all of) it: alpha ~ 1
    assert 0.2 < float(centre[..., :3].mean()) / max(float(g["raw_spp8"][..., :3].mean()), 1e-6) < 5.0


@pytest.mark.parametrize("name", SCENES)
def test_iteration_batching_is_bit_identical(gpu_lib, name):
    """rayhip_render_batch: up to max_batch() iterations share one wavefront pass (layered virtual frame); every buffer must equal
    what the same iterations give one by one"""
    w, h = 96, 80
    one = util.make_context(gpu_lib, name, w, h)
    for it in range(1, 12):
        one.render(it)
    bat = util.make_context(gpu_lib, name, w, h)
    bat.render(1)                 # plain
    bat.render_batch(2, 3)        # 2..4
    bat.render_batch(5, 7)        # 5..11
    for buf in (hip.BUF_RAW, hip.BUF_FINAL, hip.BUF_BASE_COLOR, hip.BUF_DEPTH_NORMALS):
        assert np.array_equal(one.readback(buf), bat.readback(buf)), buf
    # rect + shard inside a batch
    a = util.make_context(gpu_lib, name, w, h)
    b = util.make_context(gpu_lib, name, w, h)
    a.set_shard(32, 2, 1), b.set_shard(32, 2, 1)
    for it in range(1, 6):
        a.render(it, rect=(8, 16, 80, 48))
    b.render_batch(1, 5, rect=(8, 16, 80, 48))
    assert np.array_equal(a.readback(hip.BUF_RAW), b.readback(hip.BUF_RAW))


def test_refill_kernel_is_bit_identical(gpu_lib, monkeypatch):
    """the persistent ray-refill form of the closest-hit kernel (kernels.hip.h) performs the same node visits and triangle
    tests per ray, only interleaved differently between lanes -- hits and frames must be the same bits whether no bounce
    (RAYHIP_REFILL=0), every bounce (1) or the secondary bounces (2, the default) go through it: transparency rounds
    (cornell_principled), analytic lights (cornell_lights) and a TLAS with seven instances and visibility masks
    (cornell_instances) included.  "4" / "4any": the pooled form (prepared rays handed from lane to lane through LDS) where the
    scene suits it (one instance) and forced onto any scene -- the path of rays that cannot change lanes, taken by most rays
    of cornell_instances"""
    for name in ("cornell_principled", "cornell_lights", "cornell_instances"):
        g = util.golden_ref(name)
        ctxs = {}
        for mode in ("0", "1", "2", "3", "4"):
            monkeypatch.setenv("RAYHIP_REFILL", mode)
            ctxs[mode] = util.make_context(gpu_lib, name)
        monkeypatch.setenv("RAYHIP_POOL_ANY", "1")
        ctxs["4any"] = util.make_context(gpu_lib, name)
        monkeypatch.delenv("RAYHIP_POOL_ANY")
        monkeypatch.delenv("RAYHIP_REFILL")
        ctxs["default"] = util.make_context(gpu_lib, name)
        hits, frames = {}, {}
        for mode, ctx in ctxs.items():
            _, hits[mode], _ = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=0)
            ctx.render_batch(1, 6)
            frames[mode] = ctx.readback(hip.BUF_RAW)
        for mode in ("1", "2", "3", "4", "4any", "default"):
            assert hits[mode].tobytes() == hits["0"].tobytes(), (name, mode)
            assert np.array_equal(frames[mode], frames["0"]), (name, mode)


@pytest.mark.parametrize("name", SCENES)
def test_sized_launches_and_dynamic_chunks_are_bit_identical(gpu_lib, name, monkeypatch):
    """round 5: (1) every launch of a pass is sized from the queue census of the pass before (RAYHIP_CENSUS=0: the full grid every time),
    (2) the shadow rays of a bounce are traced on a second stream next to the closest-hit launch of the next bounce
    (RAYHIP_OVERLAP_SHADOW=0: one stream), (3) opt-in, the persistent kernels with lane refill -- K2's secondary bounces, K3, the light pick --
    take their chunks from a shared counter (wavefront.hip.h: ChunkWalk with a work counter; RAYHIP_DYNAMIC=1).  Which block walks which
    chunk, how many blocks there are and which stream a launch sits on is invisible in every image: layered and single passes, a census
    that has arrived (sync between passes) or not, one live chunk per block, four blocks per wave slot, shards and rects must all give
    the bits of the round-4 schedule."""
    w, h = 96, 80
    monkeypatch.setenv("RAYHIP_CENSUS", "0"), monkeypatch.setenv("RAYHIP_OVERLAP_SHADOW", "0")
    old = util.make_context(gpu_lib, name, w, h)  # the round-4 schedule
    monkeypatch.delenv("RAYHIP_CENSUS")
    census_only = util.make_context(gpu_lib, name, w, h)
    monkeypatch.delenv("RAYHIP_OVERLAP_SHADOW")
    default = util.make_context(gpu_lib, name, w, h)  # census + K3 next to K2 on a second stream
    monkeypatch.setenv("RAYHIP_DYNAMIC", "1")
    dynamic = util.make_context(gpu_lib, name, w, h)
    monkeypatch.setenv("RAYHIP_DYN_MULT", "4"), monkeypatch.setenv("RAYHIP_CHUNKS_PER_BLOCK", "1")
    many_blocks = util.make_context(gpu_lib, name, w, h)
    monkeypatch.delenv("RAYHIP_DYN_MULT"), monkeypatch.delenv("RAYHIP_DYNAMIC"), monkeypatch.setenv("RAYHIP_CHUNKS_PER_BLOCK", "1000")
    few_blocks = util.make_context(gpu_lib, name, w, h)
    monkeypatch.delenv("RAYHIP_CHUNKS_PER_BLOCK")
    ctxs = (old, census_only, default, dynamic, many_blocks, few_blocks)
    for ctx in ctxs:
        ctx.render_batch(1, 6)
        ctx.sync()  # (the census of the first pass is there when the second starts)
        ctx.render_batch(7, 6)
        for it in range(13, 16):  # single-layer passes, census in flight
            ctx.render(it)
        ctx.sync()
        ctx.render_batch(16, 40)  # a pass eight times the size of the one its census comes from
    for buf in (hip.BUF_RAW, hip.BUF_FINAL, hip.BUF_BASE_COLOR, hip.BUF_DEPTH_NORMALS, hip.BUF_VARIANCE):
        ref = old.readback(buf)
        for k, ctx in enumerate(ctxs[1:]):
            assert np.array_equal(ctx.readback(buf), ref), (buf, k)
    a = util.make_context(gpu_lib, name, w, h)
    monkeypatch.setenv("RAYHIP_OVERLAP_SHADOW", "0"), monkeypatch.setenv("RAYHIP_CENSUS", "0")
    b = util.make_context(gpu_lib, name, w, h)
    a.set_shard(32, 2, 1), b.set_shard(32, 2, 1)
    for first in (1, 6):
        a.render_batch(first, 5, rect=(8, 16, 80, 48)), b.render_batch(first, 5, rect=(8, 16, 80, 48))
        a.sync()
    assert np.array_equal(a.readback(hip.BUF_RAW), b.readback(hip.BUF_RAW))


@pytest.mark.parametrize("name", SCENES)
def test_light_pick_with_lane_refill_is_bit_identical(gpu_lib, name, monkeypatch):
    """the light pick as a persistent kernel whose lanes take the next point when their descent is over (k_light_pick_refill, the
    default) against the chunk-at-a-time kernel: per point the same descent, so every image is the same bits (the list of lit
    points comes out in another order, which no pixel can see)"""
    monkeypatch.setenv("RAYHIP_SHADE_SPLIT", "5")  # (against round 6's default too: the pick runs before the surface stage there)
    chunked = util.make_context(gpu_lib, name)
    monkeypatch.delenv("RAYHIP_SHADE_SPLIT")
    default = util.make_context(gpu_lib, name)
    chunked.render_batch(1, 6), default.render_batch(1, 6)
    for buf in (hip.BUF_RAW, hip.BUF_VARIANCE):
        assert np.array_equal(default.readback(buf), chunked.readback(buf)), buf
    for it in range(7, 9):  # single-layer passes
        chunked.render(it), default.render(it)
    assert np.array_equal(default.readback(hip.BUF_RAW), chunked.readback(hip.BUF_RAW))


@pytest.mark.parametrize("name", ["cornell_principled", "cornell_instances"])
def test_several_samples_of_a_pixel_in_one_wavefront_are_bit_identical(gpu_lib, name, monkeypatch):
    """the ray generator of a layered pass puts S samples of each of 64 / S pixels into a wavefront (S = 4 by default,
    RAYHIP_RAYGEN_SAMPLES = 1 / 16 / 64: one sample of 64 pixels ... 64 samples of one) -- another order of the same rays, so every
    image is the same bits (passes of 16 and of 12 layers, a rect, and a pass whose layer count is not a multiple of S, which
    falls back to the plain order)"""
    ctxs = {}
    for s in ("1", "4", "16", "64"):
        monkeypatch.setenv("RAYHIP_RAYGEN_SAMPLES", s)
        ctxs[s] = util.make_context(gpu_lib, name)
    monkeypatch.delenv("RAYHIP_RAYGEN_SAMPLES")
    ctxs["default"] = util.make_context(gpu_lib, name)
    for ctx in ctxs.values():
        ctx.render_batch(1, 16)
        ctx.render_batch(17, 12, rect=(8, 16, 40, 24))
        ctx.render_batch(29, 6)
        ctx.render_batch(35, 64)
    for s, ctx in ctxs.items():
        for buf in (hip.BUF_RAW, hip.BUF_VARIANCE, hip.BUF_BASE_COLOR, hip.BUF_DEPTH_NORMALS):
            assert np.array_equal(ctx.readback(buf), ctxs["1"].readback(buf)), (s, buf)


@pytest.mark.parametrize("name", SCENES)
def test_unused_ior_plane_is_bit_identical(gpu_lib, name, monkeypatch):
    """a scene without refractive surfaces: its passes neither write nor read the rays' stacks of refractive indices
    (ShadeParams::plain_ior) -- every image the same bits as with the plane in use; scenes WITH refraction (cornell_lights)
    keep the plane, whatever the switch says"""
    monkeypatch.setenv("RAYHIP_NO_PLAIN_IOR", "1")
    with_plane = util.make_context(gpu_lib, name)
    monkeypatch.delenv("RAYHIP_NO_PLAIN_IOR")
    default = util.make_context(gpu_lib, name)
    with_plane.render_batch(1, 6), default.render_batch(1, 6)
    for buf in (hip.BUF_RAW, hip.BUF_VARIANCE, hip.BUF_BASE_COLOR, hip.BUF_DEPTH_NORMALS):
        assert np.array_equal(default.readback(buf), with_plane.readback(buf)), buf


def test_a_stale_ior_plane_is_never_read(gpu_lib, hostsim_lib):
    """a context that rendered a scene WITH refraction (its rays' ior planes hold real stacks) and is then handed one without: its
    passes leave the plane alone (plain_ior), so nothing of that scene may read it -- the Fresnel weight of a mix node takes the medium
    outside from the stack -- and the frames must equal a fresh context's, and the host build's"""
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    r = api.CreateRenderer(api.Settings(64, 64), "HIP")
    s = r.CreateScene()
    scenes.cornell_fresnel_mix(s)
    blob = api.export_scene_blob(s)
    used = util.make_context(gpu_lib, "cornell_lights")
    used.render_batch(1, 8)  # (a Refractive block: positive entries in the ior planes of both ray buffers)
    used.upload_scene_blob(blob)
    fresh = hip.Context(0, gpu_lib)
    host = hip.Context(0, hostsim_lib)
    for ctx in (fresh, host):
        ctx.upload_static(util.pmj())
        ctx.resize(64, 64)
        ctx.upload_scene_blob(blob)
    used.clear(), fresh.clear()
    used.render_batch(1, 8), fresh.render_batch(1, 8)
    util.render_frames(host, 8)
    for buf in (hip.BUF_RAW, hip.BUF_VARIANCE):
        assert np.array_equal(used.readback(buf), fresh.readback(buf)), buf
    m = util.frame_metrics(fresh.readback(hip.BUF_RAW), host.readback(hip.BUF_RAW))
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP, m


@pytest.mark.parametrize("name", SCENES)
def test_flat_shadow_kernel_is_bit_identical(gpu_lib, name, monkeypatch):
    """K3 as the flat persistent kernel (k_trace_shadow_refill, the default over the 4-wide tree) against the nested form: the
    throughput of the oracle's own shadow rays (hook) and whole frames -- transparent surfaces between a point and its light
    (cornell_principled), analytic blockers (cornell_lights), instances with visibility masks (cornell_instances)"""
    g = util.golden_ref(name)
    monkeypatch.setenv("RAYHIP_SHADOW_REFILL", "0")
    nested = util.make_context(gpu_lib, name)
    monkeypatch.delenv("RAYHIP_SHADOW_REFILL")
    flat = util.make_context(gpu_lib, name)
    rc_nested, _ = nested.k_intersect_shadow(g["shadow_rays"], 1)
    monkeypatch.setenv("RAYHIP_HOOK_SHADOW_REFILL", "1")
    rc_flat, _ = flat.k_intersect_shadow(g["shadow_rays"], 1)
    monkeypatch.delenv("RAYHIP_HOOK_SHADOW_REFILL")
    assert rc_flat.tobytes() == rc_nested.tobytes()
    np.testing.assert_allclose(rc_flat, g["shadow_rc"], rtol=1e-6, atol=0)
    nested.render_batch(1, 6), flat.render_batch(1, 6)
    assert np.array_equal(flat.readback(hip.BUF_RAW), nested.readback(hip.BUF_RAW))
    assert np.array_equal(flat.readback(hip.BUF_VARIANCE), nested.readback(hip.BUF_VARIANCE))


@pytest.mark.parametrize("name", SCENES)
def test_wide_forms_agree_bit_for_bit(gpu_lib, name, monkeypatch):
    """the three acceleration-structure forms the kernels can walk -- the reference's BVH2 (RAYHIP_BVH_WIDTH=2), the 4-wide
    collapse (4, the default) and the 8-wide one (8: its own child order, its own triangle order, one stack entry per level)
    -- cull differently and find the same hits: hit records and frames are the same bits, through the plain kernels and
    through the persistent one"""
    g = util.golden_ref(name)
    out = {}
    for width in ("2", "4", "8"):
        for refill in ("0", "2"):
            monkeypatch.setenv("RAYHIP_BVH_WIDTH", width)
            monkeypatch.setenv("RAYHIP_REFILL", refill)
            ctx = util.make_context(gpu_lib, name)
            _, hits, _ = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=0)
            rc, _ = ctx.k_intersect_shadow(g["shadow_rays"], 1)
            ctx.render_batch(1, 6)
            out[(width, refill)] = (hits, rc, ctx.readback(hip.BUF_RAW))
    ref = out[("2", "0")]
    for key, (hits, rc, frame) in out.items():
        util.assert_hits_identical(hits, ref[0])
        assert np.array_equal(frame, ref[2]), key
    # shadow visibilities are taken by the instrumented BVH2 kernel in every mode (test hook): equal by construction
    assert np.array_equal(out[("8", "2")][1], ref[1])


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_principled", "cornell_env"])
def test_shade_forms_agree_bit_for_bit(gpu_lib, name, monkeypatch):
    """the shade stage's launch forms (RAYHIP_SHADE_SPLIT: pick folded into the surface kernel / its own kernel, next-event
    estimation and continuation as one launch / two / two with the NEE over the compacted list of lit points, the form then
    chosen on the device) group the same per-point arithmetic differently: frames, aux images and the kernel-level shade
    outputs must be the same bits"""
    frames = {}
    for split in ("0", "1", "3", "5", "13", "29"):  # (29: round 6's default -- pick first, surface + continuation fused, NEE over dense records)
        monkeypatch.setenv("RAYHIP_SHADE_SPLIT", split)
        ctx = util.make_context(gpu_lib, name)
        ctx.render_batch(1, 5)
        frames[split] = (ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_BASE_COLOR), ctx.readback(hip.BUF_DEPTH_NORMALS))
    for split, f in frames.items():
        for a, b in zip(f, frames["0"]):
            assert np.array_equal(a, b), (name, split)


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_env"])
def test_light_pick_from_memory_and_from_lds_agree(gpu_lib, name, monkeypatch):
    """round 6's form pinned (RAYHIP_SHADE_SPLIT=29): k_light_pick_first with the light table in LDS (the default while it fits) against the
    same kernel reading the table from memory (RAYHIP_PICK_LDS=0: the path of scenes whose table does not fit) -- the same descent per ray"""
    monkeypatch.setenv("RAYHIP_SHADE_SPLIT", "29")
    frames = {}
    for lds in ("1", "0"):
        monkeypatch.setenv("RAYHIP_PICK_LDS", lds)
        ctx = util.make_context(gpu_lib, name)
        ctx.render_batch(1, 6)
        ctx.render(7)
        frames[lds] = (ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_VARIANCE))
    for a, b in zip(frames["1"], frames["0"]):
        assert np.array_equal(a, b), name
    g = util.golden_ref(name)
    monkeypatch.delenv("RAYHIP_PICK_LDS")
    m = util.frame_metrics(util.render_frames(util.make_context(gpu_lib, name), 8), g["raw_spp8"])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP, m


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_principled"])
def test_the_form_chosen_from_the_census_changes_nothing(gpu_lib, name, monkeypatch):
    """default settings: the first pass of a scene takes round 6's form, later passes the form the queue census of the pass before asks for (every
    point of these scenes is lit: the three-kernel form) -- three passes of a context left to itself against both forms pinned"""
    frames = {}
    for split in (None, "13", "29"):
        if split is None:
            monkeypatch.delenv("RAYHIP_SHADE_SPLIT", raising=False)
        else:
            monkeypatch.setenv("RAYHIP_SHADE_SPLIT", split)
        ctx = util.make_context(gpu_lib, name)
        for first in (1, 5, 9):
            ctx.render_batch(first, 4)
            ctx.sync()  # (the census of a pass has arrived when the next one starts)
        frames[split] = (ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_VARIANCE))
    for split in ("13", "29"):
        for a, b in zip(frames[None], frames[split]):
            assert np.array_equal(a, b), (name, split)


def test_sparse_lights_take_the_split_form(gpu_lib, monkeypatch):
    """a scene where most shade points end their light-tree descent without a light (small emitters facing away from most
    of the scene): the device picks the split form; same frame as the combined kernel"""
    from functools import partial
    from ray_amd import api, scenes

    s = api.CreateSceneHIP(use_tex_compression=False)
    scenes.atrium(s, 0.02)
    blob = api.export_scene_blob(s)
    frames = {}
    for split in ("1", "5", "29"):
        monkeypatch.setenv("RAYHIP_SHADE_SPLIT", split)
        ctx = hip.Context(0, gpu_lib)
        ctx.upload_static(util.pmj())
        ctx.resize(160, 90)
        ctx.upload_scene_blob(blob)
        ctx.render_batch(1, 4)
        frames[split] = ctx.readback(hip.BUF_RAW)
    assert np.array_equal(frames["1"], frames["5"])
    assert np.array_equal(frames["1"], frames["29"])
    assert float(frames["5"].sum()) > 0.0


def test_triangle_pitch_is_invisible(gpu_lib, monkeypatch):
    """RAYHIP_TRI_PITCH=64 re-pitches the triangle records on the device (one 64-byte sector each): same hits, same frame"""
    name = "cornell_instances"
    g = util.golden_ref(name)
    out = {}
    for pitch in ("48", "64"):
        monkeypatch.setenv("RAYHIP_TRI_PITCH", pitch)
        ctx = util.make_context(gpu_lib, name)
        _, hits, _ = ctx.k_intersect_closest(g["primary_rays"], g["primary_hits_in"], 1, flags=0)
        ctx.render_batch(1, 4)
        out[pitch] = (hits, ctx.readback(hip.BUF_RAW))
    util.assert_hits_identical(out["64"][0], out["48"][0])
    assert np.array_equal(out["64"][1], out["48"][1])


def test_maximal_batch_and_row_limit_split(gpu_lib):
    """a pass of more layers than one accumulate launch folds (64), a frame tall enough that the layers are stacked in
    several columns of the virtual frame, and one so large that the 16-bit coordinate limit cuts the pass
    (rayhip_max_batch): all must equal the iterations rendered one by one"""
    name = "cornell_basic"
    for (w, h, n) in ((64, 48, 70), (16, 2000, 40), (24, 30000, 5)):
        one = util.make_context(gpu_lib, name, w, h)
        for it in range(1, n + 1):
            one.render(it)
        bat = util.make_context(gpu_lib, name, w, h)
        assert bat.max_batch() == min(512, (65535 // w) * (65535 // h))
        bat.render_batch(1, n)
        assert np.array_equal(one.readback(hip.BUF_RAW), bat.readback(hip.BUF_RAW)), (w, h)
        assert np.array_equal(one.readback(hip.BUF_FINAL), bat.readback(hip.BUF_FINAL)), (w, h)


@pytest.mark.parametrize("cam", ["dof_blades", "dof_disk_flength", "gaussian_clip", "lighting_only", "no_direct_no_bg", "clamped",
                                 "adaptive"])
def test_camera_features_match_the_host_build(gpu_lib, hostsim_lib, cam):
    """camera_desc_t fields the fixtures do not cover (lens model, sensor shift, pixel filters, clip range, lighting flags,
    clamps, adaptive sampling).  The host build of the same kernel sources reproduces the reference bit for bit on these
    (tests/test_hostsim_parity.py::test_camera_features_against_live_reference, where the reference tree is); here the
    GPU must agree with the host build within the image tolerance.  Scenes are built through the Ray API mirror."""
    import os
    from ray_amd import api, scenes
    from test_hostsim_parity import CAMERAS
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    name = "cornell_principled" if cam in ("lighting_only", "clamped") else "cornell_basic"
    w, h, spp = 80, 64, (12 if cam == "adaptive" else 4)
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.SCENES[name](s, **CAMERAS[cam])
    blob = api.export_scene_blob(s)
    imgs = []
    for lib in (hostsim_lib, gpu_lib):
        ctx = hip.Context(0, lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        if lib is gpu_lib and cam != "adaptive":
            ctx.render_batch(1, spp)
        else:
            util.render_frames(ctx, spp)
        imgs.append((ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_FINAL)))
    m = util.frame_metrics(imgs[1][0], imgs[0][0])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    m = util.frame_metrics(imgs[1][1], imgs[0][1])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP, m


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_filmic"])
def test_nlm_denoise_matches_the_host_build(gpu_lib, hostsim_lib, name):
    """RendererBase::DenoiseImage(region) (rt_denoise.h; host build == reference, tests/test_hostsim_parity.py): the GPU's
    filtered RAW / FINAL images against the host build's, full frame and a sub-rect, batched render in front of it; and
    through RendererHIP (DenoiseImage forces the pending iterations out first)"""
    import os
    from ray_amd import api, scenes
    w, h, spp = 64, 64, 6
    outs = []
    for lib in (hostsim_lib, gpu_lib):
        ctx = util.make_context(lib, name, w, h)
        if lib is gpu_lib:
            ctx.render_batch(1, spp)
        else:
            util.render_frames(ctx, spp)
        ctx.denoise_nlm(spp)
        full = (ctx.readback(hip.BUF_RAW), ctx.readback(hip.BUF_FINAL))
        ctx.denoise_nlm(spp, rect=(8, 12, 40, 30))  # (filters the already filtered frame again inside the rect: fine for a comparison)
        outs.append(full + (ctx.readback(hip.BUF_RAW),))
    for a, b in zip(outs[1], outs[0]):
        m = util.frame_metrics(a, b)
        assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP, m
    if os.path.exists(api.HIP_HOST_LIB):
        r = api.CreateRenderer(api.Settings(w, h), "HIP")
        s = r.CreateScene()
        scenes.SCENES[name](s)
        region = api.RegionContext((0, 0, w, h))
        for _ in range(spp):
            r.RenderScene(s, region)
        r.DenoiseImage(region)
        assert np.array_equal(r.get_raw_pixels_ref(), outs[1][0])
        assert np.array_equal(r.get_pixels_ref(), outs[1][1])


@pytest.mark.parametrize("scene", ["cornell_portals", "cornell_textures", "cornell_principled_zoo", "cornell_delta_lights", "empty_scene",
                                   "lights_only_scene", "atrium_small"])
def test_live_only_scenes_match_the_host_build(gpu_lib, hostsim_lib, scene):
    """scenes whose parity with the reference is established on the host build against the live reference
    (tests/test_hostsim_parity.py: sky portals; RGB888 / R8 / normal-map textures with mip chains): GPU vs host build"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    w, h, spp = 64, 64, 6
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    getattr(scenes, scene)(s)
    blob = api.export_scene_blob(s)
    imgs = []
    for lib in (hostsim_lib, gpu_lib):
        ctx = hip.Context(0, lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        imgs.append(util.render_frames(ctx, spp))
    m = util.frame_metrics(imgs[1], imgs[0])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m


@pytest.mark.parametrize("fn,seed,compress", [("random_cornell", 3, False), ("random_cornell", 7, False), ("random_cornell", 215, False),
                                              ("random_instances", 2, False), ("random_instances", 1013, False),
                                              ("random_textures", 2, True), ("random_textures", 3, False), ("random_textures", 5, True)])
def test_random_scenes_match_the_host_build(gpu_lib, hostsim_lib, fn, seed, compress):
    """the scene fuzzers of tests/test_hostsim_parity.py on the device (batched render, then the NLM filter): GPU vs the host
    build of the same kernel sources, which equals the reference on these scenes (tools/gpu_fuzz.py runs more of them)"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    w, h, spp = 64, 48, 4
    r = api.CreateRenderer(api.Settings(w, h, use_tex_compression=compress), "HIP")
    s = r.CreateScene()
    getattr(scenes, fn)(s, seed=seed)
    blob = api.export_scene_blob(s)
    imgs = []
    for lib in (hostsim_lib, gpu_lib):
        ctx = hip.Context(0, lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        if lib is gpu_lib:
            ctx.render_batch(1, spp)
        else:
            util.render_frames(ctx, spp)
        raw = ctx.readback(hip.BUF_RAW)
        ctx.denoise_nlm(spp)
        imgs.append((raw, ctx.readback(hip.BUF_RAW)))
    for a, b in zip(imgs[1], imgs[0]):
        m = util.frame_metrics(a, b)
        assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP, m


@pytest.mark.parametrize("seed", [10017, 7017])
def test_the_worst_fuzz_scenes_against_the_oracle_directly(gpu_lib, seed):
    """the two scenes of the 480-scene device fuzz of round 4 with the lowest scores against the HOST BUILD (profiles/r04/gpu_fuzz.txt:
    random_instances 10017: 99.935 % within tolerance, 7017: 87.2 dB) against the live reference itself -- one link instead of two
    (VERDICT round 4, weak 2): render + NLM filter as the fuzzer does, RendererRef doing the same on its own frame"""
    from ray_amd import api, scenes
    w, h, spp = 64, 48, 4
    compress = bool(seed & 1)
    ref = O.create_renderer(w, h, "REF", use_tex_compression=compress)
    rs = ref.CreateScene()
    scenes.random_instances(rs, seed=seed)
    region = api.RegionContext((0, 0, w, h))
    for _ in range(spp):
        ref.RenderScene(rs, region)
    ctx = hip.Context(0, gpu_lib)
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(O.export_scene(rs))
    ctx.render_batch(1, spp)
    m = util.frame_metrics(ctx.readback(hip.BUF_RAW), ref.get_raw_pixels_ref())
    print(f"random_instances {seed}: device against RendererRef after {spp} spp:", m)
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    ctx.denoise_nlm(spp)
    ref.DenoiseImage(region)
    m = util.frame_metrics(ctx.readback(hip.BUF_RAW), ref.get_raw_pixels_ref())
    print(f"random_instances {seed}: ... and after the NLM filter:", m)
    assert m["frac_within"] >= 0.99 and m["psnr"] >= 60.0, m


def test_compressed_textures_through_the_ray_api(gpu_lib, hostsim_lib):
    """settings_t::use_tex_compression = true (the reference's default): SceneHIP keeps the BCn storages and the exporter
    decodes them (host build == reference on this, tests/test_hostsim_parity.py); the GPU must agree with the host build"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    w, h, spp = 64, 64, 4
    blobs = []
    for compress in (True, False):
        r = api.CreateRenderer(api.Settings(w, h, use_tex_compression=compress), "HIP")
        s = r.CreateScene()
        scenes.cornell_textures(s)
        blobs.append(api.export_scene_blob(s))
    imgs = []
    for lib, blob in ((hostsim_lib, blobs[0]), (gpu_lib, blobs[0]), (gpu_lib, blobs[1])):
        ctx = hip.Context(0, lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        imgs.append(util.render_frames(ctx, spp))
    m = util.frame_metrics(imgs[1], imgs[0])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    assert not np.array_equal(imgs[1], imgs[2]), "compression did not change a texel: not exercised"


def test_a_host_written_in_c(gpu_lib, tmp_path):
    """examples/c_abi_render.c: blob -> rayhip_* calls -> PPM, compiled as C99 (no Python, no C++ in the host): the picture
    must be the tone-mapped frame the ctypes path reads back"""
    import subprocess
    import test_abi
    exe = test_abi.build_c_host(tmp_path)
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    w, h, spp = 96, 64, 6
    out = os.path.join(str(tmp_path), "o.ppm")
    r = subprocess.run([exe, os.path.join(golden, "cornell_lights.rayscene"), os.path.join(golden, "pmj02_samples.npy"), str(w), str(h), str(spp), out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    print(r.stdout.strip())
    with open(out, "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == f"{w} {h}\n".encode() and f.readline() == b"255\n"
        ppm = np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)
    ctx = util.make_context(gpu_lib, "cornell_lights", w, h)
    ctx.render_batch(1, spp)
    final = np.clip(ctx.readback(hip.BUF_FINAL)[..., :3], 0.0, 1.0)
    assert np.array_equal(ppm, (final * 255.0 + 0.5).astype(np.uint8))


def test_block_textures_are_decoded_on_the_device(gpu_lib, hostsim_lib):
    """the four block-compressed storages cross the boundary as blocks (RAYHIP_TEX_RAW_BC) and a fetch decodes its texel on
    the device: random BC1 / BC3 / BC4 / BC5 blocks (host build of the same decoder: bit-exact against the reference's
    TexStorageBCn::Get, test_hostsim_parity.py); the base-colour image of the first hits must match the host build's to
    rounding, the frame within tolerance"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    w, h, spp = 96, 96, 4
    s = api.CreateSceneHIP()
    scenes.cornell_block_textures(s)
    blob = api.export_scene_blob(s)
    imgs, base = [], []
    for lib in (gpu_lib, hostsim_lib):
        ctx = hip.Context(0, lib)
        ctx.upload_static(util.pmj())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        imgs.append(util.render_frames(ctx, spp))
        base.append(ctx.readback(hip.BUF_BASE_COLOR))
    m = util.frame_metrics(imgs[0], imgs[1])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    # base colour: texel values through srgb_to_linear (device powf): tight, not exact
    assert np.abs(base[0] - base[1]).max() <= 2e-6, np.abs(base[0] - base[1]).max()


def test_renderer_hip_reuploads_a_mutated_scene(gpu_lib, hostsim_lib):
    """scene mutators between RenderScene calls (with iterations still pending): RendererHIP must flush, notice the new
    scene version and upload the mutated arrays -- sparse pools with freed slots, rebuilt TLAS / light tree"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    w, h = 64, 48
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.cornell_instances_mutable(s)
    region = api.RegionContext((0, 0, w, h))
    for _ in range(2):
        r.RenderScene(s, region)  # (pending)
    scenes.mutate_instances_scene(s)
    r.Clear()
    region = api.RegionContext((0, 0, w, h))
    for _ in range(3):
        r.RenderScene(s, region)
    ctx = hip.Context(0, hostsim_lib)
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(api.export_scene_blob(s))
    m = util.frame_metrics(r.get_raw_pixels_ref(), util.render_frames(ctx, 3))
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m


def test_renderer_hip_camera_switch(gpu_lib, hostsim_lib):
    """two cameras in one scene, switched between RenderScene calls on the SAME region (iterations pending): the pending
    batch must be rendered with the camera it was queued with, the next iterations with the new one"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    w, h = 64, 48
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.cornell_basic(s)
    cam_b = s.AddCamera(type=0, origin=(-0.10, 0.40, 0.75), fwd=(-0.25, -0.2, -1.0), fov=50.0, gamma=2.2)
    region = api.RegionContext((0, 0, w, h))
    blobs = [api.export_scene_blob(s)]
    for _ in range(3):
        r.RenderScene(s, region)
    s.set_current_cam(cam_b)
    blobs.append(api.export_scene_blob(s))
    for _ in range(2):
        r.RenderScene(s, region)  # iterations 4, 5 of the same region, through the other camera
    ctx = hip.Context(0, hostsim_lib)
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(blobs[0])
    for it in (1, 2, 3):
        ctx.render(it)
    ctx.upload_scene_blob(blobs[1])  # (same scene, the blob's camera is the current one)
    for it in (4, 5):
        ctx.render(it)
    m = util.frame_metrics(r.get_raw_pixels_ref(), ctx.readback(hip.BUF_RAW))
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    m = util.frame_metrics(r.get_pixels_ref(), ctx.readback(hip.BUF_FINAL))
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP, m


def test_stage_times_partition_the_pass(gpu_lib):
    """RendererBase::stats_t of the HIP backend: the eleven stage entries are exclusive intervals of the pass and add up to its time (the
    reference's GPU backends report per-stage timestamps, RendererVK.cpp:452-487) -- also with the shadow launch of a bounce running on the
    second stream beside the closest-hit launch of the next one, whose elapsed time must not be booked twice"""
    import time
    ctx = util.make_context(gpu_lib, "cornell_lights", w=512, h=512)
    ctx.reserve_batch(16)
    ctx.render_batch(1, 16)  # (set-up: buffers, first launches)
    ctx.sync()
    ctx.stage_times(reset=True)
    t0 = time.perf_counter()
    for k in range(4):
        ctx.render_batch(17 + 16 * k, 16, flags=hip.FLAG_TIME_STAGES)
    ctx.sync()
    wall_us = (time.perf_counter() - t0) * 1e6
    st = ctx.stage_times(reset=True)
    total = sum(st.values())
    assert st["primary_trace"] > 0 and st["secondary_shade"] > 0 and st["secondary_shadow"] > 0, st
    # every interval lies inside the timed region (<= wall, with 2 % for the clocks) and the passes fill most of it (launch gaps aside)
    assert total <= 1.02 * wall_us, (total, wall_us, st)
    assert total >= 0.80 * wall_us, (total, wall_us, st)


def test_renderer_hip_clear_resize_stats(gpu_lib, hostsim_lib):
    """RendererHIP::Clear / Resize / GetStats / ResetStats behind the Ray API (with iterations pending in the batch queue
    when they are called): pixels against the host build (which equals the reference on this sequence,
    tests/test_hostsim_parity.py::test_clear_and_resize_against_live_reference)"""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    name, w, h = "cornell_basic", 56, 40
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.SCENES[name](s)
    ctx = hip.Context(0, hostsim_lib)
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(api.export_scene_blob(s))
    region = api.RegionContext((0, 0, w, h))
    for it in (1, 2, 3):
        r.RenderScene(s, region)  # (left pending)
        ctx.render(it)
    r.Clear((0.25, 0.5, 0.75, 1.0))
    ctx.clear((0.25, 0.5, 0.75, 1.0))
    region = api.RegionContext((0, 0, w, h))
    for it in (1, 2):
        r.RenderScene(s, region)
        ctx.render(it)
    m = util.frame_metrics(r.get_raw_pixels_ref(), ctx.readback(hip.BUF_RAW))
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    st = r.GetStats()
    assert st["primary_trace"] > 0 and st["secondary_shade"] > 0, st
    r.ResetStats()
    assert sum(r.GetStats().values()) == 0
    w2, h2 = 72, 48
    r.Resize(w2, h2)
    ctx.resize(w2, h2)
    assert r.size() == (w2, h2)
    region = api.RegionContext((0, 0, w2, h2))
    for it in (1, 2, 3):
        r.RenderScene(s, region)
        ctx.render(it)
    m = util.frame_metrics(r.get_raw_pixels_ref(), ctx.readback(hip.BUF_RAW))
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP and m["alpha_equal"], m
    assert r.get_pixels_ref().shape == (h2, w2, 4)


@pytest.mark.parametrize("name", ["cornell_lights", "cornell_filmic"])
def test_renderer_hip_through_the_ray_api(gpu_lib, name):
    """the drop-in itself: Ray::CreateRenderer(HIP) -> SceneHIP mutators -> RenderScene x N -> get_*_pixels_ref.  The live
    scene's arrays reach librayhip unserialised (alignment as the reference allocates them), and consecutive RenderScene
    calls are batched behind the API -- the pixels must equal the low-level path's, bit for bit."""
    import os
    from ray_amd import api, scenes
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")
    # (cornell_filmic: RendererHIP hands the reference's view-transform table to rayhip_set_tonemap_lut)
    r = api.CreateRenderer(api.Settings(64, 64), "HIP")
    assert r.type() == "HIP"
    s = r.CreateScene()
    scenes.SCENES[name](s)
    region = api.RegionContext((0, 0, 64, 64))
    for _ in range(3):
        r.RenderScene(s, region)
    first3 = r.get_raw_pixels_ref().copy()   # forces the 3 pending iterations out
    for _ in range(5):
        r.RenderScene(s, region)
    ctx = util.make_context(gpu_lib, name)
    a = util.render_frames(ctx, 3)
    assert np.array_equal(first3, a)
    for it in range(4, 9):
        ctx.render(it)
    assert np.array_equal(r.get_raw_pixels_ref(), ctx.readback(hip.BUF_RAW))
    assert np.array_equal(r.get_pixels_ref(), ctx.readback(hip.BUF_FINAL))
    g = util.golden_ref(name)
    m = util.frame_metrics(r.get_raw_pixels_ref(), g["raw_spp8"])
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP
    m = util.frame_metrics(r.get_pixels_ref(), g["final_spp8"])  # the tone-mapped image (8-bit display range)
    assert m["frac_within"] >= util.MIN_FRACTION and m["psnr"] >= util.MIN_PSNR_8SPP


def test_the_tie_pixels_on_the_device(gpu_lib, hostsim_lib, monkeypatch):
    """the scene of tests/test_hostsim_parity.py::test_the_tie_pixels_of_the_refined_leaves_are_pinned on the device: with the
    reference's leaves (RAYHIP_REFINE_LEAVES=0) every pixel is within tolerance of the host build walking the same leaves (which is
    RendererRef bit for bit); with the default refined leaves the two tie pixels -- and only they -- may leave it"""
    from test_hostsim_parity import TIE_SCENE as t
    from ray_amd import api, scenes
    s = api.CreateSceneHIP(use_tex_compression=False)
    scenes.random_instances(s, seed=t["seed"])
    blob = api.export_scene_blob(s)

    def frame(lib, batch):
        ctx = hip.Context(0, lib)
        ctx.upload_static(util.pmj())
        ctx.resize(t["w"], t["h"])
        ctx.upload_scene_blob(blob)
        if batch:
            ctx.render_batch(1, t["spp"])
        else:
            util.render_frames(ctx, t["spp"])
        return ctx.readback(hip.BUF_RAW).copy()

    monkeypatch.setenv("HOSTSIM_BVH4", "1")
    monkeypatch.setenv("HOSTSIM_REFINE", "0")
    ref = frame(hostsim_lib, False)
    monkeypatch.setenv("RAYHIP_REFINE_LEAVES", "0")
    plain = frame(gpu_lib, True)
    monkeypatch.delenv("RAYHIP_REFINE_LEAVES")
    refined = frame(gpu_lib, True)

    def outside(img):
        bad = np.abs(img[..., :3] - ref[..., :3]).max(axis=-1) > util.TOL_REL * np.maximum(1.0, np.abs(ref[..., :3]).max(axis=-1))
        ys, xs = np.nonzero(bad)
        return set(zip(xs.tolist(), ys.tolist()))
    assert outside(plain) == set()
    assert outside(refined) <= t["pixels"]


def test_physical_sky_on_the_device(gpu_lib, hostsim_lib):
    """the physical sky on the device (k_surface<.., SKY> queues the narrow rays that leave the scene, k_shade_sky runs the
    analytic integrator of rt_sky.h over them): frames against the committed RendererRef frames within the stated tolerance at 1 and
    8 spp, aux images, a layered pass bit-identical to single iterations; and a larger frame against the host build.  (The device's
    powf / expf / sinf are not glibc's: a handful of star pixels -- sin(1e5) amplified 4e4 times by the hash -- may leave the
    tolerance, which is what the 99.5 % bar allows.)"""
    g = util.golden_ref("cornell_sky")
    ctx = util.make_context(gpu_lib, "cornell_sky")
    ctx.render(1)
    m1 = util.frame_metrics(ctx.readback(hip.BUF_RAW), g["raw_spp1"])
    for it in range(2, 9):
        ctx.render(it)
    raw8 = ctx.readback(hip.BUF_RAW).copy()
    m8 = util.frame_metrics(raw8, g["raw_spp8"])
    print("cornell_sky 1 spp", m1, "8 spp", m8)
    assert m1["frac_within"] >= util.MIN_FRACTION and m1["alpha_equal"]
    assert m8["frac_within"] >= util.MIN_FRACTION and m8["alpha_equal"]
    np.testing.assert_allclose(ctx.readback(hip.BUF_BASE_COLOR), g["base_color_spp8"], atol=2e-3)
    batched = util.make_context(gpu_lib, "cornell_sky")
    batched.render_batch(1, 8)
    assert np.array_equal(batched.readback(hip.BUF_RAW), raw8)
    # 160 x 120, 4 spp: device against the host build of the same sources (itself the reference bit for bit)
    w, h = 160, 120
    imgs = []
    for lib in (hostsim_lib, gpu_lib):
        c = util.make_context(lib, "cornell_sky", w, h)
        if lib is gpu_lib:
            c.render_batch(1, 4)
        else:
            util.render_frames(c, 4)
        imgs.append(c.readback(hip.BUF_RAW).copy())
    m = util.frame_metrics(imgs[1], imgs[0])
    print("cornell_sky 160x120 4 spp, device vs host build", m)
    assert m["frac_within"] >= util.MIN_FRACTION


def test_a_physical_sky_without_its_tables_is_refused(gpu_lib):
    """environment_t::sky_map_spread_angle > 0 promises rayhip_scene_desc::sky*: a blob that lacks them is turned away at upload"""
    import test_hostile_scenes as H
    blob = util.golden_scene("cornell_sky")
    i, off, size = H.sections(blob)["sky"]
    bad = bytearray(blob)
    bad[H.HEADER + i * H.SECTION: H.HEADER + i * H.SECTION + 3] = b"xky"  # the section is no longer found under its name
    ctx = hip.Context(0, gpu_lib)
    ctx.upload_static(util.pmj())
    ctx.resize(32, 32)
    with pytest.raises(Exception, match="physical sky"):
        ctx.upload_scene_blob(bytes(bad))
