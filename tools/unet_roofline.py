#!/usr/bin/env python3
"""Roofline statement of the UNet denoiser's sixteen passes (f16 form) at 1920 x 1080: per pass the tensor bytes it has to read and write
(f16 tensors, one-pixel borders not counted), its FLOPs, its time (rocprofv3 --kernel-trace), the HBM-side bytes the counters saw
(--pmc FETCH_SIZE / WRITE_SIZE, separate runs), and the fractions of the two peaks (8 TB/s; 2.5 PFLOP/s dense f16).

    python tools/unet_roofline.py <trace dir> <FETCH_SIZE dir> <WRITE_SIZE dir>      (directories of rocprofv3 runs of tools/unet_bench.py N f16)
"""
import csv
import glob
import os
import sys

W, H = 1920, 1080
# (name, input channels as stored [first @ its resolution divisor, second], output channels as stored, resolution divisor, pooled output)
PASSES = [
    ("images -> tensor", None, 16, 1, False),
    ("enc_conv0", [(16, 1)], 32, 1, False), ("enc_conv1", [(32, 1)], 32, 1, True), ("enc_conv2", [(32, 2)], 48, 2, True),
    ("enc_conv3", [(48, 4)], 64, 4, True), ("enc_conv4", [(64, 8)], 80, 8, True), ("enc_conv5a", [(80, 16)], 96, 16, False),
    ("enc_conv5b", [(96, 16)], 96, 16, False), ("dec_conv4a", [(96, 16), (64, 8)], 112, 8, False), ("dec_conv4b", [(112, 8)], 112, 8, False),
    ("dec_conv3a", [(112, 8), (48, 4)], 96, 4, False), ("dec_conv3b", [(96, 4)], 96, 4, False), ("dec_conv2a", [(96, 4), (32, 2)], 64, 2, False),
    ("dec_conv2b", [(64, 2)], 64, 2, False), ("dec_conv1a", [(64, 2), (16, 1)], 64, 1, False), ("dec_conv1b", [(64, 1)], 32, 1, False),
    ("dec_conv0", [(32, 1)], None, 1, False),
]


def px(div):
    return (W // div) * (H // div)


def rows(d, want):
    out = []
    for f in glob.glob(os.path.join(d, "**", want), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def main():
    trace, fetch, write = sys.argv[1:4]
    k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows(trace, "*kernel_trace.csv")
         if "conv3x3" in r["Kernel_Name"] or "image_inputs" in r["Kernel_Name"]]
    k.sort()
    k = k[-17:]

    def counter(d, name):
        r = [(int(x.get("Dispatch_Id", 0)), float(x["Counter_Value"]), x["Kernel_Name"]) for x in rows(d, "*counter_collection.csv")
             if x["Counter_Name"] == name and ("conv3x3" in x["Kernel_Name"] or "image_inputs" in x["Kernel_Name"])]
        r.sort()
        return r[-17:]
    fs, ws = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    print(f"# UNet, f16 form, {W} x {H}: tensor bytes a pass must move (read + written, f16), its FLOPs, kernel time, counter bytes (FETCH_SIZE / WRITE_SIZE in KiB -> bytes),")
    print("# fraction of 8 TB/s by the tensor bytes and by the counters, fraction of the 2.5 PFLOP/s dense f16 matrix peak")
    print(f"# {'pass':18s} {'tensor MB':>10s} {'GFLOP':>8s} {'us':>8s} {'HBM MB (pmc)':>13s} {'TB/s tensor':>12s} {'frac':>6s} {'TB/s pmc':>9s} {'frac':>6s} {'TFLOP/s':>8s} {'frac':>6s}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for i, (name, ins, out_ch, div, pooled) in enumerate(PASSES):
        if ins is None:
            rd, flop = px(1) * 3 * 16, 0.0  # three float4 images
            wr = px(1) * out_ch * 2
        else:
            rd = sum(px(d) * ch * 2 for ch, d in ins)
            k_in = sum(ch for ch, _ in ins)
            oc = out_ch if out_ch is not None else 16  # (the last pass computes one 16-channel tile, 3 used)
            flop = 2.0 * 9 * k_in * oc * px(div)
            wr = (px(div * 2 if pooled else div) * out_ch * 2) if out_ch is not None else px(1) * 16
        us = (k[i][1] - k[i][0]) / 1e3
        pmc = (fs[i][1] + ws[i][1]) * 1024.0 if len(fs) == 17 and len(ws) == 17 else float("nan")
        t_bw, p_bw, fl = (rd + wr) / us / 1e6, pmc / us / 1e6, flop / us / 1e6
        print(f"  {name:18s} {(rd + wr) / 1e6:10.1f} {flop / 1e9:8.1f} {us:8.1f} {pmc / 1e6:13.1f} {t_bw:12.2f} {t_bw / 8.0:6.2f} {p_bw:9.2f} {p_bw / 8.0:6.2f} {fl:8.0f} {fl / 2500.0:6.2f}")
        tot[0] += rd + wr; tot[1] += flop; tot[2] += us; tot[3] += pmc
    print(f"  {'all':18s} {tot[0] / 1e6:10.1f} {tot[1] / 1e9:8.1f} {tot[2]:8.1f} {tot[3] / 1e6:13.1f} {tot[0] / tot[2] / 1e6:12.2f} {tot[0] / tot[2] / 8e6:6.2f} {tot[3] / tot[2] / 1e6:9.2f} "
          f"{tot[3] / tot[2] / 8e6:6.2f} {tot[1] / tot[2] / 1e6:8.0f} {tot[1] / tot[2] / 2.5e9:6.2f}")


if __name__ == "__main__":
    main()
