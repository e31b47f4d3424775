#!/usr/bin/env python3
"""A serving request (query encoder + labels + fused scoring of Nq queries against a resident 10 000-video bank) launched call
by call from Python vs replayed as a captured HIP graph (engine.GraphedRequest).  usage: graph_request.py [Nq ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, engine, models, synth

dev = torch.device("cuda:0")
Nv, n, F, k = 10000, 21, 4096, 100
counts = synth.clip_counts(Nv, n, seed=123)
off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
mom = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * (counts + 1) // 2)])
sd = synth.model_weights(F, seed=123)
model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
model = model.to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
emb = torch.randn((int(off[-1]), 100), generator=g, device=dev) * 0.1        # (a resident bank: its values do not matter for the timing)
bank = _vfr.VideoBank(emb, torch.from_numpy(off.astype(np.int32)).to(dev), 0, max_clips=n, total_moments=int(mom[-1]), min_clips=n)
shard = engine.CorpusShard(bank, 0, Nv, counts, mom, dev)
ops = engine.HipOps()


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out


for Nq in [int(x) for x in sys.argv[1:]] or [1, 8, 32, 64]:
    tokens = synth.query_tokens(Nq, seed=123)
    own, times = synth.annotations(Nq, counts, seed=123)
    tok_d = torch.from_numpy(tokens).to(dev)
    ws = _vfr.topk_workspace(Nq, Nv, k, dev, total_clips=int(off[-1]))

    def eager():
        Q = engine.encode_queries(model, tok_d, dev, ops)
        labels = engine.gt_labels(times, counts[own], [0.5, 0.7], True, dev, ops)
        return engine.corpus_ranks(shard, Q, own, labels, ops, k=k, workspace=ws)
    with torch.no_grad():
        ms_e, out_e = timed(eager, 20)
        gr = engine.GraphedRequest(model, shard, Nq, k, ops)
        gr.load(tokens, times, own)
        ms_g, out_g = timed(gr.replay, 50)
        gr.check()
        same = all(torch.equal(a, b) for a, b in zip(out_g, out_e))

        def request():                      # what a server does per request: stage the inputs, replay, wait, check
            gr.load(tokens, times, own); gr.replay(); torch.cuda.synchronize(); gr.check()
        t = time.perf_counter()
        for _ in range(50):
            request()
        ms_r = (time.perf_counter() - t) / 50 * 1e3
    print(f"Nq={Nq:3d}: eager {ms_e:.3f} ms   graph replay {ms_g:.3f} ms   load + replay + sync + check {ms_r:.3f} ms   identical: {same}", flush=True)
