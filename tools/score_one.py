#!/usr/bin/env python3
"""Run the fused scoring kernel a few times in one configuration (for rocprofv3 --pmc passes).
usage: score_one.py MODE [nq nv n]   MODE: topk | rank | both"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr

mode = sys.argv[1]
nq, nv, n = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (5000, 10000, 21)
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
for _ in range(2):
    if mode == "topk":
        _vfr.score_topk(Q, bank, 100, workspace=ws)
    elif mode == "rank":
        _vfr.score_topk(Q, bank, 0, rd, ri, workspace=ws)
    else:
        _vfr.score_topk(Q, bank, 100, rd, ri, workspace=ws)
torch.cuda.synchronize()
