#!/usr/bin/env python3
"""Per-rank work of the N-GPU strong-scaling bench, run alone on one GPU (no collectives): the rank's clip shard, its
1/N slice of the queries, all Nq queries against the shard.  usage: shard_sim.py [N]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, engine, models, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Nv_all, Nq, n, F, k = 10000, 5000, 21, 4096, 100
Nv = Nv_all // N
dev = "cuda:0"
sd = synth.model_weights(F, seed=123)
model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
model = model.to(dev).eval()
counts = np.full(Nv, n, np.int32)
clip_off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
seg = torch.rand((Nv * n, F), device=dev, generator=g); ctx = torch.rand((Nv, F), device=dev, generator=g)
tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
mine = tokens[: -(-Nq // N)].contiguous()
ws = _vfr.topk_workspace(Nq, Nv, k, dev)
with torch.no_grad():
    Qall = model.encode_queries(tokens)
    V = model.encode_clips(seg, ctx, clip_off)
bank = _vfr.VideoBank(V, clip_off, 0, max_clips=n, total_moments=Nv * n * (n + 1) // 2, min_clips=n)
sub = _vfr.VideoBank(V[:n].contiguous(), clip_off[:2].contiguous())
mid = _vfr.score_moments(Qall, sub)[:, 0].contiguous()
rd = torch.stack([mid, mid * 1.001]).contiguous(); ri = torch.zeros((2, Nq), dtype=torch.int64, device=dev)

def timed(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"{label:46s} {ms:8.3f} ms", flush=True)
    return ms

with torch.no_grad():
    a = timed(f"clip encoder, {Nv} videos", lambda: model.encode_clips(seg, ctx, clip_off))
    b = timed(f"query encoder, {mine.shape[0]} queries", lambda: model.encode_queries(mine))
    c = timed(f"score {Nq} x {Nv}: top-{k} + 2 rank keys", lambda: _vfr.score_topk(Qall, bank, k, rd, ri, workspace=ws))
print(f"per-rank step without collectives at N={N}: {a + b + c:.3f} ms   (1-GPU step / N = {38.0 / N:.3f} ms)")
