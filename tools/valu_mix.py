#!/usr/bin/env python3
"""Static vector-instruction mix of the closest-hit kernel's walk loop -> profiles/<round>/k2_valu_mix.json.

    python tools/valu_mix.py [out.json]        (here, no GPU: hipcc -S of ray_amd/csrc/rayhip.hip, ~30 s)

Classes and their issue costs are those tools/valu_bench.hip measured on the MI355X (profiles/r03/valu_bench.txt, 8 waves per SIMD):
  full     2.3 SIMD cycles per wave-instruction   v_mul / add / sub / fmac_f32, logic, shifts, v_mov, v_add_u32 ...
  fma      3.0                                     v_fma_f32
  half     4.2                                     compares, v_cndmask, min / max (3), conversions, v_bfe / perm / and_or / lshl_or / add3,
                                                   v_mad_*, carry adds, 64-bit shifts-and-adds, DPP moves, packed fp32
  quarter  8.2                                     v_rcp / rsq / sqrt / exp / log
Two peaks are derived for the loop's mix:
  serial   every instruction costs its class's cycles:  cycles = sum(n_c * cost_c)  -- what rounds 2-3 priced the kernel with
  paired   a full-rate instruction issues in the shadow of a half-rate one (valu_bench: "cvt_ubyte + fma" pairs run at 2.16 cycles per
           instruction, i.e. the pair costs what the conversion alone costs):  cycles = max(sum over half / quarter of n_c * cost_c,
           2.15 * n_all)  -- a LOWER bound of the cycles, so achieved / paired-peak is a fraction < 1 by construction
The mix is the static one of the walk loop's basic blocks (node step + leaf step, rare stack-spill blocks excluded), unweighted by
execution counts: an approximation, good to a few per cent (the two steps have nearly the same mix).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COST = {"full": 2.3, "fma": 3.0, "half": 4.2, "quarter": 8.2}
QUARTER = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
FULL = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
        "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b64", "v_accvgpr",
        "v_readlane", "v_writelane", "v_readfirstlane", "v_nop")


def classify(op: str, line: str) -> str:
    if "dpp" in line or "sdwa" in line:
        return "half"
    if op.startswith(QUARTER):
        return "quarter"
    if op.startswith("v_fma_f32"):
        return "fma"
    if op.startswith(FULL):
        return "full"
    return "half"


def kernel_loop_mix(asm: str, symbol_prefix: str):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol_prefix) and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # the first Depth=2 loop of the kernel is the walk loop; its blocks carry "in Loop: Header=<that label> Depth=2"
    hdr = next(l for l in body if "Loop Header: Depth=2" in l or ("Parent Loop" in l and "Depth=1" in l))
    k = body.index(hdr)
    label = None
    for j in range(k, max(k - 4, 0), -1):
        m = re.match(r"^(\.LBB\d+_\d+):", body[j])
        if m:
            label = m.group(1)[2:]  # ".LBB34_11" -> "BB34_11", as the block annotations spell it
            break
    blocks, cur, in_loop = [], [], False
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
            if cur:
                blocks.append((in_loop, cur))
            cur = []
            in_loop = (f"Header={label} " in l and "Depth=2" in l) or l.startswith(".L" + label + ":")
        cur.append(l)
    if cur:
        blocks.append((in_loop, cur))
    mix = {c: 0 for c in COST}
    ops = {}
    n_blocks = 0
    for inside, blk in blocks:
        if not inside:
            continue
        text = "\n".join(blk)
        if "scratch_" in text or "global_store" in text:  # the HBM half of the stack: rare
            continue
        n_blocks += 1
        for l in blk:
            m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
            if m:
                c = classify(m.group(1), l)
                mix[c] += 1
                ops[m.group(1)] = ops.get(m.group(1), 0) + 1
    return mix, ops, n_blocks


def main():
    import bench
    import __graft_entry__ as g
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04", "k2_valu_mix.json")
    csrc = os.path.join(ROOT, "ray_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, "rayhip.s")
        subprocess.run([g._hipcc(), *[f for f in g.HIPCC_FLAGS if f not in ("-fPIC",)], "--cuda-device-only", "-S", "rayhip.hip", "-o", s_path], cwd=csrc, check=True,
                       stderr=subprocess.DEVNULL)
        asm = open(s_path).read()
    mix, ops, n_blocks = kernel_loop_mix(asm, "_ZN2rt22k_trace_closest_refillILi4ELi40EEE")
    n = sum(mix.values())
    serial = sum(mix[c] * COST[c] for c in mix)
    paired = max(mix["half"] * COST["half"] + mix["quarter"] * COST["quarter"], 2.15 * n)
    simd_hz = 1024 * 2.4e9
    res = {"kernel": "k_trace_closest_refill<4, 40>: the walk loop (node step + leaf step)", "instructions": n, "blocks": n_blocks, "mix": mix,
           "share": {c: round(mix[c] / n, 4) for c in mix}, "cycles_per_instruction": {"serial": serial / n, "paired": paired / n},
           "peak_wave_instructions_per_s": {"serial": simd_hz * n / serial, "paired": simd_hz * n / paired},
           "costs": COST, "top_ops": dict(sorted(ops.items(), key=lambda kv: -kv[1])[:16]), "csrc_hash": bench.csrc_hash(),
           "source": "tools/valu_mix.py: static mix of the loop's blocks; class costs from profiles/r03/valu_bench.txt"}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
