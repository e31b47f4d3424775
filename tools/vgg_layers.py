#!/usr/bin/env python3
"""Per-layer time of the VGG19 conv stack from a rocprofv3 kernel trace of `vgg_bench.py 32 1`.
usage: vgg_layers.py <kernel_trace.csv> [frames]"""
import csv, sys
CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3x3_nhwc_mfma" in r["Kernel_Name"] or "conv3x3_c4_direct" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
layers, cin, hw = [], 3, 224
for c in CFG:
    if c == "M":
        hw //= 2
        continue
    layers.append((cin, c, hw)); cin = c
n = len(layers)
rows = rows[-n:]                                   # the last pass (after the warm-up call)
tot = 0.0
for (ci, co, hw), r in zip(layers, rows):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fl = 2.0 * T * hw * hw * co * ci * 9
    tot += us
    kind = "direct" if "direct" in r["Kernel_Name"] else "mfma"
    print(f"conv {ci:3d}->{co:3d} @{hw:3d}^2  {us:8.1f} us  {fl / us / 1e6:6.1f} TF   grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}  wg {r.get('Workgroup_Size_X', '?')}  {kind}")
print(f"total {tot / 1e3:.2f} ms for {T} frames")
