#!/usr/bin/env python3
"""Centred norms of the bench corpus' clip and query embeddings and what a per-video (instead of global) clip norm in the
pre-filter's error bound E2 = 20 u (R + |q|)^2 would buy: on the bench corpus |q - mu| = 1.24 dominates |v - mu| <= 0.37, the
per-video factor ((R_v + |q|) / (R + |q|))^2 averages 0.915 -- 8 % narrower windows at best; not built.

    python tools/norm_probe.py
"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import vfr_amd
from vfr_amd import _vfr, engine, models, synth
dev = torch.device("cuda", 0)
Nv, Nq, F = 10000, 5000, 4096
counts = synth.clip_counts(Nv, 21, seed=123)
off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
gen = torch.Generator(device=dev); gen.manual_seed(1234)
raw = torch.rand((int(off[-1]), F), generator=gen, device=dev)
seg = raw / (raw.norm(dim=1, keepdim=True) + 1e-5)
clip_off = torch.from_numpy(off.astype(np.int32)).to(dev)
nloc = (clip_off[1:] - clip_off[:-1]).long()
ctx = torch.segment_reduce(raw, "sum", lengths=nloc, axis=0) / nloc[:, None].float()
ctx = ctx / (ctx.norm(dim=1, keepdim=True) + 1e-5)
del raw
tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
sd = synth.model_weights(F, seed=123)
model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev).eval()
with torch.no_grad():
    emb = model.encode_clips(seg, ctx, clip_off)
    Q = model.encode_queries(tokens)
mu = emb.double().mean(0).float()
vc = (emb - mu).norm(dim=1)
qc = (Q - mu).norm(dim=1)
print("centred clip norms: min %.4f mean %.4f max %.4f std %.4f" % (vc.min(), vc.mean(), vc.max(), vc.std()))
print("centred query norms: min %.4f mean %.4f max %.4f" % (qc.min(), qc.mean(), qc.max()))
R = vc.max()
rv = vc.view(Nv, 21).max(dim=1).values
qm = qc.mean()
f = ((rv + qm) / (R + qm)) ** 2
print("per-video window factor ((R_v + q)/(R + q))^2: mean %.3f min %.3f max %.3f" % (f.mean(), f.min(), f.max()))
fc = ((vc + qm) / (R + qm)) ** 2
print("per-clip factor: mean %.3f" % fc.mean())
