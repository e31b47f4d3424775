#!/usr/bin/env python3
"""Per-convolution time of the ResNet-152 stack from a rocprofv3 kernel trace of `resnet_bench.py <frames> 1`.
usage: resnet_layers.py <kernel_trace.csv> [frames]   (prints one line per convolution KIND and stage, and the totals)"""
import csv, re, sys
from collections import OrderedDict

T = int(sys.argv[2]) if len(sys.argv) > 2 else 150
blocks, width = (3, 8, 36, 3), 64
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
gemm = [r for r in rows if "gemm_nt" in name(r) or "conv3x3_nhwc_mfma" in name(r) or "mfma" in name(r).lower() and "gemm" in name(r).lower()]
# the plan in EXECUTION order (csrc/resnet.hip): stem; per block [downsample], conv1, conv2, conv3
plan = [("stem 7x7/2", 3, width, 7, 2, 112)]
h, cin = 56, width
for li, nb in enumerate(blocks):
    mid = width << li
    for b in range(nb):
        s = 2 if (b == 0 and li > 0) else 1
        ho = h // s
        if b == 0:
            plan.append((f"layer{li + 1} downsample 1x1/{s}", cin, 4 * mid, 1, s, ho))
        plan.append((f"layer{li + 1} conv1 1x1", cin, mid, 1, 1, h))
        plan.append((f"layer{li + 1} conv2 3x3/{s}", mid, mid, 3, s, ho))
        plan.append((f"layer{li + 1} conv3 1x1 +res", mid, 4 * mid, 1, 1, ho))
        h, cin = ho, 4 * mid
n = len(plan)
assert len(gemm) >= n, (len(gemm), n)
gemm = gemm[-n:]                                    # the last pass
agg = OrderedDict()
tot_us = tot_fl = 0.0
for (label, ci, co, k, s, ho), r in zip(plan, gemm):
    fl = 2.0 * T * ho * ho * co * ci * k * k
    us = dur(r)
    a = agg.setdefault(label + f" {ci}->{co} @{ho}^2", [0, 0.0, 0.0, name(r)[:40]])
    a[0] += 1; a[1] += us; a[2] += fl
    tot_us += us; tot_fl += fl
for label, (c, us, fl, kn) in agg.items():
    print(f"{label:44s} x{c:2d}  {us / c:8.1f} us each  {us / 1e3:7.2f} ms  {fl / us / 1e6:6.1f} TF   {kn}")
other = [r for r in rows if r not in gemm]
print(f"convolution GEMMs: {tot_us / 1e3:.2f} ms for {T} frames = {tot_fl / tot_us / 1e6:.1f} TF")
last_start = int(gemm[0]["Start_Timestamp"])
extra = {}
for r in rows:
    if int(r["Start_Timestamp"]) >= last_start and r not in gemm:
        e = extra.setdefault(name(r)[:50], [0, 0.0]); e[0] += 1; e[1] += dur(r)
for k_, (c, us) in sorted(extra.items(), key=lambda kv: -kv[1][1]):
    print(f"  other: {k_:50s} x{c:3d} {us / 1e3:7.3f} ms")
