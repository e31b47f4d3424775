#!/usr/bin/env python3
"""Micro-benchmark of the fused scoring kernel alone (GPU box).  Usage: python tools/score_bench.py [nq nv n]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr

nq, nv, n = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (5000, 10000, 21)
dev = "cuda:0"
torch.manual_seed(0)
V = torch.randn(nv * n, 100, device=dev) * 0.1
Q = torch.randn(nq, 100, device=dev) * 0.1
off = torch.arange(0, nv * n + 1, n, dtype=torch.int32, device=dev)
bank = _vfr.VideoBank(V, off)
ws = _vfr.topk_workspace(nq, nv, 100, dev)


def timed(label, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"{label:42s} {ms:9.3f} ms   {nq * nv / ms * 1e3:.3e} scorings/s", flush=True)
    return ms


d, i, _ = _vfr.score_topk(Q, bank, 100, workspace=ws)
rd = torch.stack([d[:, 50], d[:, 99]]).contiguous()     # two rank keys in the top-100 region (selective)
ri = torch.stack([i[:, 50], i[:, 99]]).contiguous()
M = n * (n + 1) // 2
mid_id = torch.full((2, nq), (nv // 2) * M, dtype=torch.int64, device=dev)
sub = _vfr.VideoBank(V[(nv // 2) * n:(nv // 2 + 1) * n].contiguous(), off[:2].contiguous())
mid_d = _vfr.score_moments(Q, sub)[:, 0].contiguous()   # a typical (median-ish) distance as rank key
mid_d = torch.stack([mid_d, mid_d * 1.001]).contiguous()

for fast in (1, 2, 0):
    _vfr.set_option("score_fast", 1 if fast else 0); _vfr.set_option("score_split", 1 if fast == 2 else 0)
    tag = {1: "fused", 2: "split", 0: "v1   "}[fast]
    timed(f"[{tag}] top-100 only", lambda: _vfr.score_topk(Q, bank, 100, workspace=ws))
    timed(f"[{tag}] rank x2 only (selective keys)", lambda: _vfr.score_topk(Q, bank, 0, rd, ri, workspace=ws))
    timed(f"[{tag}] rank x2 only (median keys)", lambda: _vfr.score_topk(Q, bank, 0, mid_d, mid_id, workspace=ws))
    timed(f"[{tag}] top-100 + rank x2 (median keys)", lambda: _vfr.score_topk(Q, bank, 100, mid_d, mid_id, workspace=ws))
    timed(f"[{tag}] top-100 + rank x2 (selective keys)", lambda: _vfr.score_topk(Q, bank, 100, rd, ri, workspace=ws))
    timed(f"[{tag}] top-1 only", lambda: _vfr.score_topk(Q, bank, 1, workspace=ws))
