#!/usr/bin/env python3
"""ResNet-152 variant of the extractor (get_rgb_features.py:127-131) through the HIP stack: frames/s and per-site times.
usage: resnet_bench.py [frames] [reps]   (150 frames = one full DiDeMo video)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda:0"
for _item in filter(None, __import__("os").environ.get("VFR_OPTS", "").split(",")):     # e.g. VFR_OPTS=gemm_small=1
    _vfr.set_option(_item.split("=")[0], int(_item.split("=")[1]))
packed = _vfr.resnet_pack(synth.resnet_weights(seed=1), device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
frames = torch.randint(0, 256, (T, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
out = _vfr.resnet_pool(frames, packed); torch.cuda.synchronize()
_vfr.set_option("profile", 1); _vfr.profile_read(True)
t = time.perf_counter()
for _ in range(reps):
    out = _vfr.resnet_pool(frames, packed)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
sites = _vfr.profile_read(True)
flop = 2 * 11.51e9 * T            # 11.51 GMAC per 224x224 frame up to the average pool (convolutions; BatchNorm folded)
print(f"{T} frames: {dt * 1e3:.1f} ms  -> {T / dt:.1f} frames/s, {flop / dt / 1e12:.1f} TFLOP/s algorithmic "
      f"({100 * flop / dt / 157.3e12:.1f}% of fp32 MFMA peak); output finite: {bool(torch.isfinite(out).all())}, mean {float(out.mean()):.4f}")
for k, (ms, n) in sites.items():
    print(f"   {k:14s} {ms / reps:9.2f} ms/call-set  ({n // reps} launches)")
