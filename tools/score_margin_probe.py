#!/usr/bin/env python3
"""Score distribution of the bench corpus around the rank keys: how many (query, video) pairs hold a moment within +-delta of a
rank key (what an approximate pre-filter with error bound delta has to hand to the exact path).

    python tools/score_margin_probe.py [--videos 10000] [--queries 5000] [--sample 64]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=10000)
    ap.add_argument("--queries", type=int, default=5000)
    ap.add_argument("--clips", default="21")
    ap.add_argument("--sample", type=int, default=64)
    args = ap.parse_args()
    import vfr_amd  # noqa: F401
    from vfr_amd import _vfr, engine, models, synth
    dev = torch.device("cuda", 0)
    n_clips = args.clips if args.clips == "didemo" else int(args.clips)
    Nv, Nq, F = args.videos, args.queries, 4096
    counts = synth.clip_counts(Nv, n_clips, seed=123)
    off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    mom = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * (counts + 1) // 2)])
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    raw = torch.rand((int(off[-1]), F), generator=gen, device=dev)
    seg = raw / (raw.norm(dim=1, keepdim=True) + 1e-5)
    clip_off = torch.from_numpy(off.astype(np.int32)).to(dev)
    nloc = (clip_off[1:] - clip_off[:-1]).long()
    ctx = torch.segment_reduce(raw, "sum", lengths=nloc, axis=0) / nloc[:, None].float()
    ctx = ctx / (ctx.norm(dim=1, keepdim=True) + 1e-5)
    del raw
    tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
    own, times = synth.annotations(Nq, counts, seed=123)
    sd = synth.model_weights(F, seed=123)
    model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    ops = engine.HipOps()
    labels = engine.gt_label_table(times, counts[own], [0.5, 0.7])
    with torch.no_grad():
        emb = model.encode_clips(seg, ctx, clip_off)
        Q = model.encode_queries(tokens)
    bank = _vfr.VideoBank(emb, clip_off, 0)
    shard = engine.CorpusShard(bank, 0, Nv, counts, mom, dev)
    gt = engine.prepare_gt(shard, own, labels)
    keys = engine.best_positive_keys(shard, Q, gt, ops, 1)
    rd, ri = engine._unpack_key(keys)
    vn, qn = emb.norm(dim=1), Q.norm(dim=1)
    print(f"|v| min/mean/max {vn.min():.4f} {vn.mean():.4f} {vn.max():.4f}   |q| {qn.min():.4f} {qn.mean():.4f} {qn.max():.4f}")
    S = args.sample
    sel = torch.arange(0, Nq, max(1, Nq // S), device=dev)[:S]
    dense = _vfr.score_moments(Q[sel].contiguous(), bank)                    # [S, total moments]
    print(f"scores: min {dense.min():.4f} mean {dense.mean():.4f} max {dense.max():.4f} std {dense.std():.5f}")
    # clip distances (1-clip moments are the first n entries of each video's block)
    M = int(n_clips) * (int(n_clips) + 1) // 2 if n_clips != "didemo" else None
    if M:
        dv = dense.view(S, Nv, M)
        d1 = dv[:, :, :int(n_clips)]
        print(f"clip distances: min {d1.min():.4f} mean {d1.mean():.4f} max {d1.max():.4f} std {d1.std():.5f}")
        srt = dense.sort(dim=1).values
        print(f"top-100 spread: best {srt[:, 0].mean():.5f}  100th {srt[:, 99].mean():.5f}  128th {srt[:, 127].mean():.5f}; "
              f"gap 100th->128th {(srt[:, 127] - srt[:, 99]).mean():.3e}  min {(srt[:, 127] - srt[:, 99]).min():.3e}")
        for delta in (3e-6, 5e-6, 7e-6, 1e-5, 1.7e-5, 3e-5, 1e-3, 1e-2):
            amb = torch.zeros((S, Nv), dtype=torch.bool, device=dev)
            for r in range(2):
                x = rd[r][sel][:, None, None]
                amb |= ((dv - x).abs() <= delta * dv.clamp(min=1e-3)).any(dim=2)
            print(f"delta {delta:.1e} (relative): ambiguous (query, video) pairs {amb.float().mean() * 100:.2f} %   "
                  f"rank keys: mean {rd.mean():.4f}, percentile of key in scores ~ {(dense < rd[0][sel][:, None]).float().mean() * 100:.1f} %")


if __name__ == "__main__":
    main()
