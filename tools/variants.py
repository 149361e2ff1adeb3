#!/usr/bin/env python3
"""Build-time knob sweep for the kernels (tuning tool, not part of the product).

    python tools/variants.py build               # here (no GPU): hipcc every variant into _build/variants/<name>/
    python tools/variants.py run [workload] [K]  # on the GPU box: time each variant, check images agree

Variants are -D flag sets for ray_amd/csrc/rayhip.hip (see the knob list at the top of kernels.hip.h).
"""
import concurrent.futures as cf
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "ray_amd", "csrc")
VDIR = os.path.join(CSRC, "_build", "variants")

VARIANTS = json.loads(os.environ.get("RT_VARIANTS", "null")) or {
    "base": [],
    "stack32": ["-DRT_LDS_STACK_DEPTH=32"],
    "stack32_w5": ["-DRT_LDS_STACK_DEPTH=32", "-DRT_TRACE_MIN_WAVES=5"],
    "stack24_w6": ["-DRT_LDS_STACK_DEPTH=24", "-DRT_TRACE_MIN_WAVES=6"],
    "contract": ["-ffp-contract=fast"],
}


def build_one(name, flags):
    import __graft_entry__ as g
    out = os.path.join(VDIR, name)
    os.makedirs(out, exist_ok=True)
    shade_only = [f[7:] for f in flags if f.startswith("+shade:")]  # "+shade:<flag>": for shade_kernels.hip only
    trace_only = [f[7:] for f in flags if f.startswith("+trace:")]  # "+trace:<flag>": for rayhip.hip only
    flags = [f for f in flags if not f.startswith("+")]
    base = [f for f in g.HIPCC_FLAGS if not (f.startswith("-ffp-contract") and any(x.startswith("-ffp-contract") for x in flags))]
    objs = {"rayhip": ("rayhip.hip", trace_only),
            "shade": ("shade_kernels.hip", shade_only)}
    class R:
        stderr = ""
        returncode = 0
    r = R()
    for o, (src, extra) in objs.items():
        cmd = [g._hipcc(), *base, *flags, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(out, o + ".o")]
        rr = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        r.stderr += rr.stderr
        r.returncode |= rr.returncode
    # (sort.o and unet_kernels.o carry no tuning knobs: the objects of the regular build are linked as they are)
    subprocess.run([g._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[os.path.join(out, o + ".o") for o in objs],
                    os.path.join(CSRC, "_build", "sort.o"), os.path.join(CSRC, "_build", "unet_kernels.o"), "-o", os.path.join(out, "librayhip.so")],
                   cwd=CSRC, check=True)
    res = {}
    cur = None
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            cur = line.split("Function Name:")[1].split("[")[0].strip()
            cur = "pool" if "k_trace_closest_pool" in cur else "refill" if "k_trace_closest_refill" in cur else "closest" if "k_trace_closestILb0" in cur else "shadow" if "k_trace_shadowILb0" in cur else \
                "surface" if "k_surfaceILb0ELb0" in cur else "scatter" if "k_scatterILb1ELb1" in cur else "scatter_nee" if "k_scatterILb1ELb0" in cur else \
                "scatter_cont" if "k_scatterILb0ELb1" in cur else "pick" if "k_light_pick" in cur else None
        elif cur and any(k in line for k in ("VGPRs:", "ScratchSize", "Occupancy", "LDS Size")):
            body = line.split("remark:")[1].rsplit("[-Rpass", 1)[0]
            k, v = body.rsplit(":", 1)
            res.setdefault(cur, {})[k.strip().split(" ")[0]] = v.strip()
    if r.returncode != 0:
        print(name, "FAILED\n", r.stderr[-2000:])
    return name, res


def build():
    todo = {k: v for k, v in VARIANTS.items() if not any(f.startswith("+build:") for f in v)}  # "+build:<name>": reuse that build
    with cf.ThreadPoolExecutor(4) as ex:
        for name, res in ex.map(lambda kv: build_one(*kv), todo.items()):
            print(name, json.dumps(res))


def run(workload="sponza", K=16):
    # one process per variant: the HIP runtime resolves kernels by name across loaded code objects, so two
    # variants of the same kernels must never live in one process
    import bench
    bench.get_scene_blob(workload, bench.WORKLOADS[workload], 0, 1, lambda: None)  # build + cache once
    for name in VARIANTS:
        build_name = next((f[7:] for f in VARIANTS[name] if f.startswith("+build:")), name)
        if os.path.exists(os.path.join(VDIR, build_name, "librayhip.so")):
            subprocess.run([sys.executable, __file__, "run1", name, workload, str(K)])


def run1(name, workload="sponza", K=16):
    # NOTE: must not load libray_hip.so here (ray_amd.api): it pulls the default librayhip.so into the global
    # symbol scope and the variant's kernel stubs (weak template symbols) would bind to it
    import numpy as np
    import bench
    from ray_amd import hip
    wl = bench.WORKLOADS[workload]
    cache_dir = os.environ.get("RAY_AMD_CACHE", "/tmp/ray_amd_cache")
    with open(os.path.join(cache_dir, f"{workload}_{wl.get('detail', 0)}.rayscene"), "rb") as f:
        blob = f.read()
    pmj = np.load(os.path.join(ROOT, "tests", "golden", "pmj02_samples.npy"))
    ref_path = f"/tmp/variants_ref_{workload}_{K}_{os.environ.get('RT_RES_SCALE', '1')}.npy"
    ref_img = np.load(ref_path) if os.path.exists(ref_path) else None
    for f in VARIANTS[name]:  # "+env:NAME=VALUE" pseudo-flags: run-time switches of librayhip for this variant
        if f.startswith("+env:"):
            k, v = f[5:].split("=", 1)
            os.environ[k] = v
    if True:
        build_name = next((f[7:] for f in VARIANTS[name] if f.startswith("+build:")), name)
        path = os.path.join(VDIR, build_name, "librayhip.so")
        L = hip.Library(path)
        ctx = hip.Context(0, L)
        ctx.upload_static(pmj)
        scale = float(os.environ.get("RT_RES_SCALE", "1"))  # tuning: does kernel time scale with the ray count?
        wl = dict(wl, w=int(wl["w"] * scale), h=int(wl["h"] * scale))
        ctx.resize(wl["w"], wl["h"])
        ctx.upload_scene_blob(blob)
        for it in range(1, 3):
            ctx.render(it, flags=(hip.FLAG_SORT_RAYS if "+sort" in VARIANTS[name] else 0))
        ctx.sync()
        ctx.trav_timing()
        ctx.stage_times()
        t0 = time.perf_counter()
        rflags = (hip.FLAG_SORT_RAYS if "+sort" in VARIANTS[name] else 0)
        batch = int(os.environ.get("RT_BATCH", "1"))
        if batch > 1:
            ctx.render_batch(3, K, flags=hip.FLAG_TIME_STAGES | rflags)  # warm-up pass of the same shape (allocations)
            ctx.sync()
            ctx.trav_timing()
            ctx.stage_times()
            t0 = time.perf_counter()
            ctx.render_batch(3 + K, K, flags=hip.FLAG_TIME_STAGES | rflags)
            it_next = 3 + 2 * K
        else:
            for it in range(3, 3 + K):
                ctx.render(it, flags=hip.FLAG_TIME_STAGES | rflags)
            it_next = 3 + K
        ctx.sync()
        dt = time.perf_counter() - t0
        (k2, n2), (k3, n3) = ctx.trav_timing()
        st = ctx.stage_times()
        img = ctx.readback(hip.BUF_RAW)
        if ref_img is None:
            ref_img = img
            np.save(ref_path, img)
        d = np.abs(img - ref_img)
        # (iterations must stay consecutive: a pixel whose required_samples is below the iteration counts as converged)
        ctx.render(it_next, flags=hip.FLAG_COUNT_TRAVERSAL)
        it_next += 1
        c2, c3 = ctx.trav_counters()
        print(f"{name:14s} {wl['w']}x{wl['h']} {wl['w'] * wl['h'] * K / dt / 1e6:7.1f} Msamples/s  step {dt / K * 1e3:6.2f} ms | K2 {k2 / K:6.2f} ms K3 {k3 / K:5.2f} ms "
              f"shade {(st['primary_shade'] + st['secondary_shade']) / K / 1e3:5.2f} ms gen {st['primary_ray_gen'] / K / 1e3:4.2f} sort {st['secondary_sort'] / K / 1e3:4.2f} | "
              f"vs first: max|d| {d.max():.2e} frac>1e-3 {(d.max(axis=-1) > 1e-3).mean():.2e} | max_stack {c2['max_stack']}/{c3['max_stack']}",
              flush=True)
        if hasattr(L.lib, "rayhip_tuning_read_shade_profile"):
            import ctypes
            f = L.lib.rayhip_tuning_read_shade_profile
            f.argtypes, f.restype = [ctypes.POINTER(ctypes.c_ulonglong * 64), ctypes.c_int], ctypes.c_int
            buf = (ctypes.c_ulonglong * 64)()
            f(ctypes.byref(buf), 1)
            names = {0: "scatter stage", 2: "sample_light", 4: "  triangle light", 6: "  environment light", 8: "principled: lobes", 10: "principled: NEE",
                     12: "principled: continuation", 14: "  diffuse draw", 16: "  gloss draw", 18: "  coat draw", 20: "  transmission draw", 22: "diffuse material",
                     24: "glossy material", 26: "refractive material", 28: "roulette survivors", 30: "shadow ray set-up"}
            print("  shade-stage lane census (all passes so far): section, wave-level executions, share of the stage's executions, active lanes")
            for k in sorted(names):
                if buf[k + 1]:
                    print(f"    {names[k]:28s} {buf[k + 1]:12d} {buf[k + 1] / max(buf[1], 1):6.3f} {buf[k] / buf[k + 1]:6.1f}")
        if hasattr(L.lib, "rayhip_tuning_read_profile"):
            import ctypes
            f = L.lib.rayhip_tuning_read_profile
            f.argtypes, f.restype = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong * 32), ctypes.c_int], ctypes.c_int
            buf = (ctypes.c_ulonglong * 32)()
            f(ctx._ctx, ctypes.byref(buf), 1)
            if os.environ.get("RT_PROF_RAW"):
                print("  raw (timed passes):", list(buf))
            if batch > 1:
                ctx.render_batch(it_next, K)
            else:
                for it in range(it_next, it_next + K):
                    ctx.render(it)
            f(ctx._ctx, ctypes.byref(buf), 1)
            if os.environ.get("RT_PROF_RAW"):
                print("  raw:", list(buf))
            names = {0: "load ray+hit", 1: "miss/env", 2: "light hit", 3: "surface setup (verts, TBN, lod)", 4: "mix + normal map + tangent",
                     5: "NEE: per-type light sample", 15: "NEE: light-tree descent", 6: "textures + ray init", 7: "diffuse eval+sample", 8: "glossy eval+sample",
                     9: "refractive eval+sample", 10: "emissive", 11: "principled setup", 12: "principled eval",
                     13: "principled sample", 14: "tail (RR, shadow ray)", 28: "pixel write", 29: "compaction + stores",
                     30: "chunk fetch / loop", 31: "exit",
                     16: "K2 node: loop/stack -> fetch issue", 17: "K2 node: wait for node data", 18: "K2 node: 4 box tests + sort",
                     19: "K2 node: push/pop (LDS)", 20: "K2 leaf: before tri fetch", 21: "K2 leaf: wait for first tri",
                     22: "K2 leaf: tri tests (+prefetch waits)", 23: "K2 TLAS: instance transform", 24: "K2 chunk fetch",
                     25: "K2 ray load", 26: "K2 scene walk residue", 27: "K2 store + exit"}
            if buf[9]:
                names.update({24: "refill: phase selection", 19: "refill: phase A (node visits)", 26: "refill: phase B (leaves)", 23: "refill: phase C (TLAS)", 25: "refill: phase D (finish + refill)"})
            if buf[1] and "PROFILE_TRACE" in " ".join(VARIANTS[name]):
                print(f"  lane utilisation: node visits {buf[0] / buf[1] / 64:.3f} ({buf[1]} wave-level visits), "
                      f"triangle tests {buf[2] / max(buf[3], 1) / 64:.3f} ({buf[3]}), instance entries {buf[4] / max(buf[5], 1) / 64:.3f} ({buf[5]})")
                if buf[8]:
                    print(f"  majority loop: {buf[8]} iterations; lanes at a node {buf[6] / buf[8] / 64:.3f}, at a leaf {buf[7] / buf[8] / 64:.3f}, "
                          f"done or idle {1 - (buf[6] + buf[7]) / buf[8] / 64:.3f}")
                if buf[9]:
                    print(f"  refill kernel: {buf[9]} service rounds, {buf[10] / buf[9]:.1f} lanes served per round, {buf[11] / buf[9]:.2f} top-level steps per round")
                if buf[12]:
                    print(f"  pooled kernel: {buf[12]} batch prepares, {buf[13] / buf[12]:.1f} rays per batch; {buf[14]} swaps, {buf[15] / max(buf[14], 1):.1f} lanes per swap")
                for k in range(16):
                    buf[k] = 0
            tot = float(sum(buf)) or 1.0
            print("  wave time by section:")
            for k in range(32):
                if buf[k]:
                    print(f"    {k:2d} {names.get(k, '?'):34s} {100.0 * buf[k] / tot:5.1f} %")
        ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run1":
        run1(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        run(*(sys.argv[2:3] or ["sponza"]), *(int(x) for x in sys.argv[3:4]))
