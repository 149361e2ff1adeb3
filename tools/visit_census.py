"""Node visits and triangle tests per ray of the 4-wide walk for one scene under the reference builder's options (tuning tool, runs
on the CPU: tests/hostsim walks the very trees librayhip would upload).

    python tools/visit_census.py [detail=4.3] [width=480] [height=270] [refine=2]

Builds the atrium three ways -- the reference's plain SAH build, mesh_desc_t::allow_spatial_splits, mesh_desc_t::use_fast_bvh_build
(/root/reference/internal/BVHSplit.cpp:148, 323-470; SceneBase.h:130-131) -- and renders one iteration of each through the host build of
the kernel sources with the counters of RAYHIP_FLAG_COUNT_WIDE on.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    detail = float(sys.argv[1]) if len(sys.argv) > 1 else 4.3
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 270
    refine = sys.argv[4] if len(sys.argv) > 4 else "2"
    os.environ["HOSTSIM_BVH4"] = "1"
    os.environ["HOSTSIM_REFINE"] = refine
    import numpy as np

    import oracle_lib as O
    from ray_amd import api, hip, scenes
    lib = hip.Library(O.HOSTSIM_LIB, prefix="hostsim_")
    frames = {}
    for label, kw in (("plain SAH", {}), ("spatial splits", dict(spatial_splits=True)), ("fast build", dict(fast_bvh_build=True))):
        t0 = time.time()
        s = api.CreateSceneHIP()
        ntris = scenes.atrium(s, detail, **kw)
        blob = api.export_scene_blob(s)
        t_build = time.time() - t0
        ctx = hip.Context(0, lib)
        ctx.upload_static(api.pmj_table())
        ctx.resize(w, h)
        ctx.upload_scene_blob(blob)
        ctx.trav_counters(reset=True)
        ctx.render(1, flags=hip.FLAG_COUNT_WIDE)
        c2, c3 = ctx.trav_counters(reset=True)
        frames[label] = ctx.readback(hip.BUF_RAW).copy()
        r2, r3 = max(c2["rays"], 1), max(c3["rays"], 1)
        print(f"{label:15s} detail {detail}: {ntris} triangles, {s.triangle_count()} references, {s.node_count()} BVH2 nodes, build {t_build:.1f} s | "
              f"closest: {c2['rays']} rays, {c2['nodes4'] / r2:.2f} wide nodes / ray, {c2['tris'] / r2:.2f} triangle tests / ray | "
              f"shadow: {c3['rays']} rays, {c3['nodes4'] / r3:.2f} nodes, {c3['tris'] / r3:.2f} triangles", flush=True)
        ctx.close()
    base = frames["plain SAH"]
    for label, f in frames.items():
        print(f"{label:15s} frame == plain SAH frame: {bool(np.array_equal(f, base))} (max |diff| {float(np.abs(f - base).max()):.3g})")


if __name__ == "__main__":
    main()
