#!/usr/bin/env python3
"""exact vs MFMA pre-filter (f32: must be identical) vs bf16 (agreement rates) on random corpora, with timings.

    python tools/mfma_check.py [nv nq clips k]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa: E402,F401
from vfr_amd import _vfr  # noqa: E402

dev = torch.device("cuda", 0)
_vfr.set_option("score_mfma_min", 0)                  # small cases must exercise the pre-filter kernels


def run(nv, nq, clips, k, scale=0.1, seed=0, reps=3, offset=0.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    rs = np.random.RandomState(seed)
    counts = np.full(nv, clips) if isinstance(clips, int) else rs.choice([5, 6], nv)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    V = torch.randn((int(off[-1]), 100), device=dev, generator=g) * scale + offset     # offset: a common shift of every embedding
    Q = torch.randn((nq, 100), device=dev, generator=g) * scale + offset
    bank = _vfr.VideoBank(V, torch.from_numpy(off).to(dev))
    ws = _vfr.topk_workspace(nq, nv, k, dev, total_clips=int(off[-1]))
    # rank keys: two mid-distribution moments per query (scores of random moments of random videos)
    vsel = torch.from_numpy(rs.randint(0, nv, nq).astype(np.int32)).to(dev)
    own = _vfr.score_own(Q, bank, vsel)
    M0 = int(counts.min()) * (int(counts.min()) + 1) // 2
    pick = torch.from_numpy(rs.randint(0, M0, (2, nq))).to(dev)
    rd = torch.stack([own.gather(1, pick[r][:, None]).squeeze(1) for r in range(2)]).contiguous()
    ri = (bank.mom_off[vsel.long()][None, :] + pick).contiguous()
    res = {}
    for mode in ("exact", "mfma", "bf16"):
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = _vfr.score_topk(Q, bank, k, rd, ri, workspace=ws, mode=mode)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res[mode] = (out, dt)
        extra = ""
        if mode == "mfma":
            extra = str(_vfr.score_mfma_stats(ws, nq, bank, k))
        print(f"  {mode:6s} {dt * 1e3:8.3f} ms  {extra}")
    (d0, i0, c0), _ = res["exact"]
    (d1, i1, c1), _ = res["mfma"]
    (d2, i2, c2), _ = res["bf16"]
    ok = torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(c0, c1)
    print(f"  mfma == exact: {ok}   (idx {torch.equal(i0, i1)}, dist {torch.equal(d0, d1)}, counts {torch.equal(c0, c1)}"
          f", count diffs {(c0 != c1).sum().item()})")
    r1 = (i0[:, 0] == i2[:, 0]).float().mean().item()
    kk = min(10, k)
    ov = np.mean([len(set(a) & set(b)) / kk for a, b in zip(i0[:, :kk].tolist(), i2[:, :kk].tolist())])
    ovk = np.mean([len(set(a) & set(b)) / k for a, b in zip(i0.tolist(), i2.tolist())])
    rel = ((c2 - c0).abs().float() / bank.total_moments).max().item()
    print(f"  bf16: rank@1 agreement {r1:.4f}, top-{kk} overlap {ov:.4f}, top-{k} overlap {ovk:.4f}, max |rank count diff| / moments {rel:.2e}")
    return ok


if __name__ == "__main__":
    if len(sys.argv) > 4:
        nv, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4])
        clips = int(sys.argv[3]) if sys.argv[3].isdigit() else sys.argv[3]
        cases = [(nv, nq, clips, k)]
    else:
        cases = [(1, 1, 6, 10), (3, 65, 21, 100), (40, 130, 21, 100), (300, 64, 6, 100), (700, 200, "didemo", 50),
                 (2500, 500, 21, 100), (10000, 5000, 21, 100), (2000, 300, 21, 100, 0.1, 3, 2, 4.0)]
    allok = True
    for c in cases:
        print(c)
        allok &= run(*c)
    print("ALL IDENTICAL" if allok else "MISMATCH")
