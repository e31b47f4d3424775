#!/usr/bin/env python3
"""Randomised differential check of the fused scorer (threshold ladder, skip, merges, rank counts) against the dense kernel +
a stable sort, on the GPU box.  usage: score_fuzz.py [configs] [seed] [mode: exact|mfma] -- mfma also draws offset corpora
and passes a threshold seed now and then"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr

dev = "cuda:0"
ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
MODE = sys.argv[3] if len(sys.argv) > 3 else "exact"
_vfr.set_option("score_mfma_min", 0)
fell = 0
bad = 0
for it in range(ncfg):
    if MODE == "mfma":       # the round-4 switches too: sorted pass whatever the size (2) or by size (1), early-out deferral, histogram threshold
        _vfr.set_option("score_sort", int(rs.choice([1, 2, 2])))
        _vfr.set_option("score_defer", int(rs.choice([-1, 0, 8, 8, 20])))
        _vfr.set_option("score_hist", int(rs.choice([0, 1, 1])))
    shape = rs.choice(["n21", "n6", "ragged56", "ragged21"])
    nv = int(rs.choice([1, 3, 31, 33, 100, 257, 600, 1100, 3000]))
    nq = int(rs.choice([1, 5, 63, 64, 65, 200, 1000]))
    k = int(rs.choice([1, 10, 100, 128, 300]))
    if shape == "n21": counts = np.full(nv, 21)
    elif shape == "n6": counts = np.full(nv, 6)
    elif shape == "ragged56": counts = rs.choice([5, 6], nv)
    else: counts = rs.randint(1, 22, nv); counts[rs.randint(nv)] = 21
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    g = torch.Generator(device=dev).manual_seed(int(rs.randint(1 << 30)))
    scale = float(rs.choice([0.1, 1.0]))
    shift = float(rs.choice([0.0, 0.0, 0.5, 3.0])) if MODE == "mfma" else 0.0
    V = torch.randn((int(off[-1]), 100), device=dev, generator=g) * scale + shift
    Q = torch.randn((nq, 100), device=dev, generator=g) * scale + shift
    if rs.rand() < 0.3 and nv > 2:                       # duplicated videos: exact ties across videos
        n0 = int(counts[0])
        for v in range(1, nv):
            if counts[v] == n0: V[off[v]:off[v + 1]] = V[off[0]:off[1]]; break
    bank = _vfr.VideoBank(V, torch.from_numpy(off).to(dev))
    dense = _vfr.score_moments(Q, bank)
    total = dense.shape[1]
    order = torch.argsort(dense, dim=1, stable=True)
    kk = min(k, total)
    # two rank keys at random positions of the true order
    p0, p1 = int(rs.randint(0, total)), int(rs.randint(0, total))
    rd = torch.stack([dense.gather(1, order[:, p0:p0 + 1]).squeeze(1), dense.gather(1, order[:, p1:p1 + 1]).squeeze(1)]).contiguous()
    ri = torch.stack([order[:, p0], order[:, p1]]).contiguous()
    seed = None
    if MODE == "mfma" and rs.rand() < 0.3 and total > k:   # a valid threshold seed: the key a little beyond the k-th best
        pos = min(total - 1, kk + int(rs.randint(0, 50)))
        bits = dense.gather(1, order[:, pos:pos + 1]).squeeze(1).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        seed = ((bits << 32) | order[:, pos]).contiguous()
    ws = _vfr.topk_workspace(nq, nv, k, dev, total_clips=int(off[-1]))
    d, i, c = _vfr.score_topk(Q, bank, k, rd, ri, workspace=ws, thr_seed=seed, mode=MODE)
    if MODE == "mfma" and k <= 253:
        fell += _vfr.score_mfma_stats(ws, nq, bank, k)["fallback_groups"]
    ok = torch.equal(i[:, :kk], order[:, :kk]) and torch.equal(d[:, :kk], dense.gather(1, order[:, :kk]))
    if kk < k: ok = ok and bool((i[:, kk:] == -1).all())
    ok = ok and c[0].tolist() == [p0] * nq and c[1].tolist() == [p1] * nq
    _, _, c2 = _vfr.score_topk(Q, bank, 0, rd, ri, mode=MODE)
    ok = ok and torch.equal(c2, c)
    print(f"{it:3d} {shape:9s} nv={nv:5d} nq={nq:5d} k={k:4d} moments={total:7d} ranks=({p0},{p1}) {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad, " fallback groups over all configs:", fell)
sys.exit(1 if bad else 0)
