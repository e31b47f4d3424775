#!/usr/bin/env python3
"""a3 (model/data.py:163-181): 25-frame pooling + L2 norm of fc7 frame features, HBM-bound.  usage: pool_bench.py [videos]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
T, F = 150, 4096
frames = torch.rand((nv * T, F), device=dev)
counts = [T] * nv
for mode in ("avg", "max"):
    _vfr.segment_pool_norm_batch(frames, counts, 25, mode); torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        seg, ctx, n = _vfr.segment_pool_norm_batch(frames, counts, 25, mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    gb = frames.numel() * 4 / 1e9
    print(f"{mode}: {nv} videos x {T} frames x {F}: {dt * 1e3:.3f} ms -> {gb / dt:.0f} GB/s algorithmic "
          f"({100 * gb / dt / 8000:.1f}% of 8 TB/s; 6.3 TB/s is the measured copy ceiling)")
