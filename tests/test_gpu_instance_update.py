"""Dynamic scenes: instance transforms, instance add / remove, lights and the environment change
between frames; meshes, materials and textures stay.  rayhip_scene_update_instances re-sends the small arrays and rebuilds
the top-level tree ON THE DEVICE (ray_amd/csrc/lbvh.hip.h) instead of a new rayhip_scene_upload.

A BVH only culls: the frames after an update must be the frames of a context that got the same scene through a full upload
(another top-level tree: the host's SAH tree there, the linear builder's here) -- bit for bit, apart from exact-distance ties
between two instances (none in these scenes).  Reference: SceneCPU.cpp:1004-1094 (mutators), 1103-1162 (RebuildTLAS),
1411-1521 (RebuildLightTree)."""
import os
import time

import numpy as np
import pytest

import util
from ray_amd import api, hip, scenes

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def gpu_lib():
    lib = hip.Library()
    assert lib.device_count() > 0, "no HIP device: the product has no CPU path, -m gpu tests cannot run here"
    return lib


def _need_host_lib():
    if not os.path.exists(api.HIP_HOST_LIB):
        pytest.skip("libray_hip.so not built (needs the reference tree at build time)")


def _context(lib, w, h, blob):
    ctx = hip.Context(0, lib)
    ctx.upload_static(util.pmj())
    ctx.resize(w, h)
    ctx.upload_scene_blob(blob)
    return ctx


def test_update_matches_a_full_upload(gpu_lib):
    """drop an instance and a light, move an instance, add a light, change the environment"""
    _need_host_lib()
    w, h, spp = 96, 64, 4
    s = api.CreateSceneHIP()
    scenes.cornell_instances_mutable(s)
    before = api.export_scene_blob(s)
    scenes.mutate_instances_scene(s)
    after = api.export_scene_blob(s)

    ctx = _context(gpu_lib, w, h, before)
    first = util.render_frames(ctx, spp)
    assert ctx.update_instances(after) == 0
    ctx.clear()
    updated = util.render_frames(ctx, spp)
    fresh = util.render_frames(_context(gpu_lib, w, h, after), spp)
    assert not np.array_equal(first, updated)
    assert np.array_equal(updated, fresh)
    # and back again: the update is not a one-way street
    assert ctx.update_instances(before) == 0
    ctx.clear()
    assert np.array_equal(util.render_frames(ctx, spp), first)


def test_a_thousand_instances(gpu_lib, capfd, monkeypatch):
    """every one of 1000 instances moves (16 of them lamps: 192 triangle lights move with them)"""
    _need_host_lib()
    w, h, spp = 128, 96, 2
    s = api.CreateSceneHIP()
    scenes.instance_field(s, 1000)
    before = api.export_scene_blob(s)
    scenes.move_instance_field(s)
    after = api.export_scene_blob(s)
    ctx = _context(gpu_lib, w, h, before)
    util.render_frames(ctx, 1)
    monkeypatch.setenv("RAYHIP_TRACE_UPLOAD", "1")
    ctx.update_instances(after)  # (first call: the builder's scratch buffers are allocated)
    ctx.update_instances(before)
    ctx.sync()
    t0 = time.perf_counter()
    assert ctx.update_instances(after) == 0
    ctx.sync()
    dt = time.perf_counter() - t0
    monkeypatch.delenv("RAYHIP_TRACE_UPLOAD")
    print(capfd.readouterr().err)
    print(f"update of 1000 instances (host gather + device top-level build + light arrays): {dt * 1e3:.2f} ms")
    ctx.clear()
    updated = util.render_frames(ctx, spp)
    fresh = util.render_frames(_context(gpu_lib, w, h, after), spp)
    differing = int((np.abs(updated - fresh).max(axis=-1) > 0).sum())
    print("pixels differing from a full upload:", differing)
    assert differing <= 4  # interpenetrating blocks: an exact-distance tie may pick the other instance
    assert dt < 0.25


def test_update_refuses_what_it_cannot_do(gpu_lib):
    """an instance of a mesh that was not in use at the upload, geometry of another size: 2 = upload the scene; the device keeps
    rendering the scene it has"""
    _need_host_lib()
    w, h = 64, 48
    s = api.CreateSceneHIP()
    scenes.cornell_instances(s)
    a = api.export_scene_blob(s)
    s2 = api.CreateSceneHIP()
    scenes.cornell_lights(s2)
    b = api.export_scene_blob(s2)
    ctx = _context(gpu_lib, w, h, a)
    first = util.render_frames(ctx, 2)
    assert ctx.update_instances(b) == 2
    ctx.clear()
    assert np.array_equal(util.render_frames(ctx, 2), first)
    empty = hip.Context(0, gpu_lib)
    empty.upload_static(util.pmj())
    empty.resize(w, h)
    assert empty.update_instances(a) == 2  # nothing uploaded yet


def test_renderer_hip_takes_the_update_path(gpu_lib, capfd, monkeypatch):
    """RendererHIP: SetMeshInstanceTransform + Finalize between frames goes through the update, AddMesh through an upload"""
    _need_host_lib()
    w, h = 64, 48
    r = api.CreateRenderer(api.Settings(w, h), "HIP")
    s = r.CreateScene()
    scenes.cornell_instances_mutable(s)
    region = api.RegionContext((0, 0, w, h))
    r.RenderScene(s, region)
    r.get_raw_pixels_ref()
    monkeypatch.setenv("RAYHIP_TRACE_UPLOAD", "1")
    capfd.readouterr()
    scenes.mutate_instances_scene(s)
    r.Clear()
    region = api.RegionContext((0, 0, w, h))
    for _ in range(3):
        r.RenderScene(s, region)
    got = r.get_raw_pixels_ref()
    log = capfd.readouterr().err
    assert "instances updated" in log and "bvh uploaded" not in log, log
    fresh = util.render_frames(_context(gpu_lib, w, h, api.export_scene_blob(s)), 3)
    assert np.array_equal(got, fresh)
    # a new mesh: a full upload again
    attrs, idx = scenes.cornell_mesh_arrays(scenes._block_quads("tall"))
    grey = s.AddMaterial(api.ShadingNode(type=api.eShadingNode.Diffuse, base_color=(0.4, 0.4, 0.4)))
    s.AddMeshInstance(s.AddMesh(attrs, idx, [(grey, None, 0, 30)]))
    s.Finalize()
    r.Clear()
    region = api.RegionContext((0, 0, w, h))
    r.RenderScene(s, region)
    r.get_raw_pixels_ref()
    assert "bvh uploaded" in capfd.readouterr().err


def test_single_instance_moved_removed_and_added_again(gpu_lib):
    """the smallest top level (one instance: the builder's root gets a far-away point box as second child), an empty one,
    and one instance again -- each through the update path, each equal to a full upload of the same scene"""
    _need_host_lib()
    w, h, spp = 64, 48, 3
    s = api.CreateSceneHIP()
    s.SetEnvironment(env_col=(0.2, 0.25, 0.3), back_col=(0.2, 0.25, 0.3))
    grey = s.AddMaterial(api.ShadingNode(type=api.eShadingNode.Diffuse, base_color=(0.5, 0.5, 0.5)))
    attrs, idx = scenes.cornell_mesh_arrays(scenes._block_quads("tall"))
    mesh = s.AddMesh(attrs, idx, [(grey, None, 0, 30)])
    mi = s.AddMeshInstance(mesh)
    s.AddLight("sphere", color=(4.0, 4.0, 4.0), position=(-0.1, 0.5, 0.1), radius=0.03)
    scenes._cornell_camera(s)
    s.Finalize()
    ctx = _context(gpu_lib, w, h, api.export_scene_blob(s))
    frames = [util.render_frames(ctx, spp)]
    for step in ("move", "remove", "add"):
        if step == "move":
            s.SetMeshInstanceTransform(mi, scenes._xform(translate=(0.1, 0.05, -0.05), rot_y_deg=30.0, scale=(1.2, 0.8, 1.0)))
        elif step == "remove":
            s.RemoveMeshInstance(mi)
        else:
            mi = s.AddMeshInstance(mesh, scenes._xform(translate=(-0.1, 0.0, 0.1)))
        s.Finalize()
        blob = api.export_scene_blob(s)
        assert ctx.update_instances(blob) == 0, step
        ctx.clear()
        frames.append(util.render_frames(ctx, spp))
        assert np.array_equal(frames[-1], util.render_frames(_context(gpu_lib, w, h, blob), spp)), step
        assert not np.array_equal(frames[-1], frames[-2]), step
