#!/usr/bin/env python3
"""Dense GEMM with the ping-pong option on/off (GPU box)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr
dev = "cuda:0"
torch.manual_seed(0)
for name, M, K, N in (("long K [5120x4096]x[4096x4096]^T", 5120, 4096, 4096), ("square 4096^3", 4096, 4096, 4096)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    for pp in (0, 1):
        _vfr.set_option("gemm_pp", pp)
        _vfr.linear(A, W); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): _vfr.linear(A, W)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 10 * 1e3
        print(f"{name:36s} pp={pp} {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
_vfr.set_option("gemm_pp", 0)
