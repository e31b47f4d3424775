#!/usr/bin/env python3
"""Small dense GEMMs of the pass (context rows, output layer, query projection) with 64-row vs 32-row tiles.  GPU box."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr
dev = "cuda:0"
torch.manual_seed(0)
SHAPES = (("vis_seg/8", 26250, 4096, 500), ("vis_seg/4", 52500, 4096, 500), ("vis_seg/2", 105000, 4096, 500), ("vis_out/8", 26250, 500, 100),
          ("vis_out", 210000, 500, 100)) if len(sys.argv) > 1 else None
for name, M, K, N in SHAPES or (("vis_ctx", 10000, 4096, 500), ("vis_ctx/8", 1250, 4096, 500), ("lang_fc", 5000, 2000, 100), ("vgg fc6", 150, 25088, 4096),
                      ("vgg fc7", 150, 4096, 4096), ("vocab table", 400, 100, 4096), ("2500x500x4096", 2500, 4096, 500)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    res = {}
    for mode in ((64, 100000) if SHAPES else (64, 0)):
        _vfr.set_option("gemm_small", mode)
        out = _vfr.linear(A, W); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): out = _vfr.linear(A, W)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t) / 10 * 1e3, out)
    _vfr.set_option("gemm_small", 0)
    print(f"{name:16s} [{M}x{K}]x[{N}x{K}]^T  64-row {res[64][0]:7.3f} ms   32-row {res[100000 if SHAPES else 0][0]:7.3f} ms   same bits {torch.equal(res[100000 if SHAPES else 0][1], res[64][1])}", flush=True)
