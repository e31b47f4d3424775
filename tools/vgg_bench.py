#!/usr/bin/env python3
"""BASELINE config 2 shape: raw 224x224x3 frames -> VGG19-fc7 clip features through the HIP stack.
usage: vgg_bench.py [frames] [reps]   (150 frames = one full DiDeMo video, get_rgb_features.py:47-61)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, synth
for _item in filter(None, __import__("os").environ.get("VFR_OPTS", "").split(",")):     # e.g. VFR_OPTS=vgg_halo=0
    _vfr.set_option(_item.split("=")[0], int(_item.split("=")[1]))

T = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda:0"
CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
g = torch.Generator(device=dev); g.manual_seed(0)
cw, cb, cin = [], [], 3
for c in CFG:
    if c == "M":
        continue
    cw.append(torch.randn((c, cin, 3, 3), generator=g, device=dev) * (2.0 / (cin * 9)) ** 0.5)
    cb.append(torch.randn((c,), generator=g, device=dev) * 0.05)
    cin = c
fc6 = (torch.randn((4096, 25088), generator=g, device=dev) * (2.0 / 25088) ** 0.5, torch.zeros(4096, device=dev))
fc7 = (torch.randn((4096, 4096), generator=g, device=dev) * (2.0 / 4096) ** 0.5, torch.zeros(4096, device=dev))
frames = torch.randint(0, 256, (T, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)

out = _vfr.vgg_fc7(frames, CFG, cw, cb, fc6, fc7); torch.cuda.synchronize()
_vfr.set_option("profile", 1); _vfr.profile_read(True)
t = time.perf_counter()
for _ in range(reps):
    out = _vfr.vgg_fc7(frames, CFG, cw, cb, fc6, fc7)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
sites = _vfr.profile_read(True)
flop = 39.26e9 * T
print(f"{T} frames: {dt * 1e3:.1f} ms  -> {T / dt:.1f} frames/s, {flop / dt / 1e12:.1f} TFLOP/s algorithmic "
      f"({100 * flop / dt / 157.3e12:.1f}% of fp32 MFMA peak); output finite: {bool(torch.isfinite(out).all())}, "
      f"mean {float(out.mean()):.4f}")
for k, (ms, n) in sites.items():
    print(f"   {k:14s} {ms / reps:9.2f} ms/call-set  ({n // reps} launches)")
