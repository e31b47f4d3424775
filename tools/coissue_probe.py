#!/usr/bin/env python3
"""Do the MFMA-bound encoders and the VALU-bound scorer overlap when they run on two streams?  (GPU box)
Times encoders alone, scoring alone, and both at once (scoring of the previous batch beside the encoders of the next)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, engine, models, synth
dev = torch.device("cuda:0")
Nv, Nq, n, F, k = 10000, 5000, 21, 4096, 100
counts = np.full(Nv, n, np.int64)
off = torch.arange(0, Nv * n + 1, n, dtype=torch.int32, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
seg = torch.rand((Nv * n, F), device=dev, generator=g); seg /= seg.norm(dim=1, keepdim=True)
ctx = torch.rand((Nv, F), device=dev, generator=g); ctx /= ctx.norm(dim=1, keepdim=True)
tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
sd = synth.model_weights(F, seed=123)
model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()}); model = model.to(dev).eval()
with torch.no_grad():
    V = model.encode_clips(seg, ctx, off); Q = model.encode_queries(tokens)
bank = _vfr.VideoBank(V, off, 0, max_clips=n, total_moments=Nv * 231, min_clips=n)
sub = _vfr.VideoBank(V[:n].contiguous(), off[:2].contiguous())
mid = _vfr.score_moments(Q, sub)[:, 0].contiguous()
rd = torch.stack([mid, mid * 1.001]).contiguous(); ri = torch.zeros((2, Nq), dtype=torch.int64, device=dev)
ws = _vfr.topk_workspace(Nq, Nv, k, dev)
s2 = torch.cuda.Stream(dev)

def enc():
    with torch.no_grad():
        model.encode_clips(seg, ctx, off); model.encode_queries(tokens)
def score():
    _vfr.score_topk(Q, bank, k, rd, ri, workspace=ws)
def both():
    s2.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s2):
        score()
    enc()
    torch.cuda.current_stream(dev).wait_stream(s2)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
te, tsc, tb = timed(enc), timed(score), timed(both)
print(f"encoders alone {te:.2f} ms   scoring alone {tsc:.2f} ms   sum {te + tsc:.2f} ms   both on two streams {tb:.2f} ms")
