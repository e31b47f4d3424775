#!/usr/bin/env python3
"""One training step of CALModel (model/main.py:58-67: three clip batches + one query batch forward, ranking loss, backward) on
the GPU: the HIP autograd functions (train.py) vs the torch.nn sub-modules (models.HIP_TRAINING = False) on the same weights.
usage: train_bench.py [samples per batch] [reps]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa: E402,F401
from vfr_amd import losses, models, synth  # noqa: E402

from vfr_amd import _vfr  # noqa: E402
for _item in filter(None, __import__("os").environ.get("VFR_OPTS", "").split(",")):
    _vfr.set_option(_item.split("=")[0], int(_item.split("=")[1]))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
sd = synth.model_weights(4096, seed=1)
rs = np.random.RandomState(2)
P, Nn = 3 * S, 3 * S                                   # ~3 clip rows per sample in each of posit / intra / inter
posit = torch.from_numpy(rs.rand(P, 8194).astype(np.float32)).to(dev)
intra = torch.from_numpy(rs.rand(Nn, 8194).astype(np.float32)).to(dev)
inter = torch.from_numpy(rs.rand(P, 8194).astype(np.float32)).to(dev)
lang = torch.from_numpy(synth.query_tokens(S, seed=3)).to(dev)
maskp = torch.from_numpy(np.sort(rs.randint(0, S, P))).to(dev); maskp[:S] = torch.arange(S, device=dev); maskp = maskp.sort().values
maskn = torch.from_numpy(np.sort(rs.randint(0, S, Nn))).to(dev); maskn[:S] = torch.arange(S, device=dev); maskn = maskn.sort().values


def step(model, opt):
    opt.zero_grad()
    pe, ne, ie = model(posit), model(intra), model(inter)
    le = model(lang, False, dev)
    loss, n = losses.ranking_loss(pe, ne, ie, le, maskp, maskn)
    loss.backward()
    opt.step()
    return float(loss.detach())


for hip in (True, False):
    models.HIP_TRAINING = hip
    m = models.CALModel(8194, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    torch.manual_seed(0)
    first = step(m, opt)
    step(m, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        last = step(m, opt)
    torch.cuda.synchronize()
    print(f"{'HIP autograd functions' if hip else 'torch.nn sub-modules  '}: {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms / training step "
          f"({S} samples, {P}+{Nn}+{P} clip rows); loss first step {first:.5f}")
models.HIP_TRAINING = True
