#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in librayhip (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py [extra hipcc flags...]   (CPU only: the compiler reports the numbers)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function",
        "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kr.o"] + sys.argv[1:]
out = ""
for src, extra in (("rayhip.hip", []), ("shade_kernels.hip", [])):
    out += subprocess.run(base + extra + ["-c", src], cwd=os.path.join(ROOT, "ray_amd", "csrc"), capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(rt::SceneView.*|\(.*", "", cur).replace("void rt::", "").replace("rt::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        key = {"TotalSGPRs": "SGPRs", "VGPRs Spill": "Spill"}.get(m.group(1), m.group(1).split(" ")[0])
        rows[cur][key] = int(m.group(2))
print(f"{'kernel':64s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'spill':>6s} {'LDS':>6s} {'waves':>5s}")
for k, v in sorted(rows.items()):
    print(f"{k[:64]:64s} {v.get('VGPRs', 0):5d} {v.get('AGPRs', 0):5d} {v.get('SGPRs', 0):5d} {v.get('ScratchSize', 0):8d} {v.get('Spill', 0):6d} {v.get('LDS', 0):6d} {v.get('Occupancy', 0):5d}")
