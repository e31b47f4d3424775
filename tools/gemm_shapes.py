#!/usr/bin/env python3
"""Dense chain GEMM over tile counts around whole / partial rounds of the 512 workgroup slots (GPU box)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr
dev = "cuda:0"
torch.manual_seed(0)
K, N = 2048, 4096
W = torch.randn(N, K, device=dev)
for rt in (8, 16, 24, 32, 40, 48, 56, 64):
    M = rt * 128
    A = torch.randn(M, K, device=dev)
    _vfr.linear(A, W); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): _vfr.linear(A, W)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"{rt * 32:5d} tiles = {rt * 32 / 512:4.2f} rounds: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s  ({ms / (rt * 32 / 512):.3f} ms per round)", flush=True)
