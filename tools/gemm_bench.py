#!/usr/bin/env python3
"""Time the chain GEMM (vfr_linear_f32) at the hot shapes.  usage: gemm_bench.py [reps] [other libvfr build, for A/B timing]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr

dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
MODE = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None   # gemm_small: 1 = 64-row tiles for the large GEMMs (experiment); also
if MODE is not None:                                                                 # checks the bits against the default path
    pass
elif len(sys.argv) > 2:
    _vfr.LIB_PATH = Path(sys.argv[2]).resolve()
torch.manual_seed(0)
for name, M, K, N in (("lstm_rec  [5000x1000]x[4000x1000]^T", 5000, 1000, 4000), ("lstm step shape [5120x1152]x[4096x1152]^T", 5120, 1152, 4096),
                      ("2x rows  [10240x1152]x[4096x1152]^T", 10240, 1152, 4096), ("long K [5120x4096]x[4096x4096]^T", 5120, 4096, 4096), ("vis_seg  [210000x4096]x[500x4096]^T", 210000, 4096, 500),
                      ("square 4096^3", 4096, 4096, 4096)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    ref = _vfr.linear(A, W)
    if MODE is not None:
        _vfr.set_option("gemm_small", MODE)
    same = torch.equal(_vfr.linear(A, W), ref); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        _vfr.linear(A, W)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    _vfr.set_option("gemm_small", 0)
    print(f"{name:40s} {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s" + (f"   same bits as default: {same}" if MODE is not None else ""), flush=True)
