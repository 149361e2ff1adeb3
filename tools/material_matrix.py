"""The reference's own material test matrix (90 entries of tests/test_shading.cpp, tests/golden/material_matrix.json) on the reference's own test
meshes: every entry rendered by the live oracle (RendererRef from oracle/_ref) and by this backend, frame against frame.

    python tools/material_matrix.py host [size] [spp_cap]     # the host build of the kernel sources (tests/hostsim): expected bit-equal
    python tools/material_matrix.py gpu  [size] [spp_cap]     # librayhip on the GPU: the stated tolerance (tests/util.py)

tests/test_material_matrix.py runs the same function per entry under pytest; this prints the whole table (profiles/r05/material_matrix_*.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    side = sys.argv[1] if len(sys.argv) > 1 else "host"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if side == "gpu" else 64)
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else (None if side == "gpu" else 4)
    only = sys.argv[4:]
    if side == "gpu":
        import torch  # noqa: F401  (first: its HIP runtime opens the device)
    import oracle_lib as O
    import ref_material_scene as M
    import util
    from ray_amd import hip
    if side == "gpu":
        lib = hip.Library()

        def make(w, h, blob):
            ctx = hip.Context(0, lib)
            ctx.upload_static(util.pmj())
            ctx.resize(w, h)
            ctx.upload_scene_blob(blob)
            return ctx
    else:
        make = O.hostsim_context
    threads = max(1, min(16, os.cpu_count() or 1)) if side == "gpu" else 1
    print(f"# {side}: {size} x {size}, samples = min(the test's own count, {cap}); oracle threads {threads}")
    print(f"# {'test':28s} {'scene variant':22s} spp  raw: within-tolerance  PSNR dB   bit-equal px   final / base colour / depth-normals equal   substitutions")
    worst = (2.0, 1e9, None)
    t0 = time.time()
    failures = 0
    for e in M.matrix():
        if only and e["name"] not in only:
            continue
        m, notes = M.run_entry(e, make, size, size, spp_cap=cap, batched=(side == "gpu"), threads=threads)
        raw = m["raw"]
        spp = e["max_samples"] if cap is None else min(e["max_samples"], cap)
        ok = raw["equal"] if side == "host" else (raw["frac_within"] >= util.MIN_FRACTION and raw["psnr"] >= util.MIN_PSNR_8SPP)
        failures += 0 if ok else 1
        if (raw["frac_within"], raw["psnr"]) < worst[:2]:
            worst = (raw["frac_within"], raw["psnr"], e["name"])
        aux = " ".join("=" if m[k]["equal"] else f"{m[k]['frac_within']:.4f}" for k in ("final", "base_color", "depth_normals"))
        print(f"{'  ' if ok else 'X '}{e['name']:28s} {e['scene']:22s} {spp:3d}  {raw['frac_within']:.6f}  {raw['psnr']:7.1f}  {raw['exact']:.4f}   {aux:28s} "
              f"{'; '.join(n.split(':')[0] for n in notes)}", flush=True)
    print(f"# worst: {worst[0]:.6f} within tolerance, {worst[1]:.1f} dB ({worst[2]}); {failures} below the bar; {time.time() - t0:.0f} s")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
