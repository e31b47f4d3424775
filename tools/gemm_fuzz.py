#!/usr/bin/env python3
"""Randomised check of the MFMA chain GEMM against the VALU chain kernel (same chains -> identical bits), GPU box."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr
dev = "cuda:0"
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    M = int(rs.choice([1, 7, 63, 64, 65, 127, 128, 129, 500, 1000, 4097, 20000, 70000]))
    N = int(rs.choice([1, 5, 100, 127, 128, 129, 500, 1000, 2050]))
    K = int(rs.choice([1, 3, 4, 31, 32, 33, 64, 100, 500, 768, 1000, 4096]))
    if M * N * K > 3e11: K = 128
    relu, bias = bool(rs.randint(2)), bool(rs.randint(2))
    g = torch.Generator(device=dev).manual_seed(int(rs.randint(1 << 30)))
    A = torch.randn((M, K), device=dev, generator=g); W = torch.randn((N, K), device=dev, generator=g)
    b = torch.randn((N,), device=dev, generator=g) if bias else None
    out = {}
    for mode in (1, 0):
        _vfr.set_option("gemm", mode)
        out[mode] = _vfr.linear(A, W, b, relu=relu)
    _vfr.set_option("gemm", 1)
    ok = torch.equal(out[0], out[1])
    print(f"{it:3d} M={M:6d} N={N:5d} K={K:5d} bias={int(bias)} relu={int(relu)} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
