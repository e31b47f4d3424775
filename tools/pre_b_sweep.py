#!/usr/bin/env python3
"""Fused scoring pass (top-100 + 2 rank keys, 5000 x 10000 x 21) vs the size of ladder stage B (GPU box)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr
import os
nq, nv, n = 5000, int(os.environ.get("NV", 10000)), 21
dev = "cuda:0"
torch.manual_seed(0)
V = torch.randn(nv * n, 100, device=dev) * 0.1
Q = torch.randn(nq, 100, device=dev) * 0.1
off = torch.arange(0, nv * n + 1, n, dtype=torch.int32, device=dev)
bank = _vfr.VideoBank(V, off)
ws = _vfr.topk_workspace(nq, nv, 100, dev)
M = n * (n + 1) // 2
sub = _vfr.VideoBank(V[(nv // 2) * n:(nv // 2 + 1) * n].contiguous(), off[:2].contiguous())
mid = _vfr.score_moments(Q, sub)[:, 0].contiguous()
rd = torch.stack([mid, mid * 1.001]).contiguous()
ri = torch.full((2, nq), (nv // 2) * M, dtype=torch.int64, device=dev)
for b in [int(x) for x in sys.argv[1:]] or [0, 256, 384, 512, 768, 1024, 1536, 2048]:
    _vfr.set_option("score_pre_b", b)
    _vfr.score_topk(Q, bank, 100, rd, ri, workspace=ws); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): _vfr.score_topk(Q, bank, 100, rd, ri, workspace=ws)
    torch.cuda.synchronize()
    print(f"stage B = {b:5d} videos: {(time.perf_counter() - t) / 5 * 1e3:8.3f} ms", flush=True)
_vfr.set_option("score_pre_b", 0)
