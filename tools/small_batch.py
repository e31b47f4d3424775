#!/usr/bin/env python3
"""One serving request: Nq queries (encoder + labels + fused scoring, top-100 + two rank keys) against a resident bank of
10 000 videos x 21 clips.  Prints the time per request and the profiler sites.  usage: small_batch.py [Nq ...]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, engine, models, synth

for _item in filter(None, __import__("os").environ.get("VFR_OPTS", "").split(",")):     # e.g. VFR_OPTS=lstm_persist=0
    _vfr.set_option(_item.split("=")[0], int(_item.split("=")[1]))
Nv, n, F, k = 10000, 21, 4096, 100
dev = torch.device("cuda:0")
counts = synth.clip_counts(Nv, n, seed=123)
off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
mom = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * (counts + 1) // 2)])
g = torch.Generator(device=dev).manual_seed(5)
emb = torch.randn((int(off[-1]), 100), device=dev, generator=g) * 0.3
clip_off = torch.from_numpy(off.astype(np.int32)).to(dev)
bank = _vfr.VideoBank(emb, clip_off, 0, max_clips=n, total_moments=int(mom[-1]), min_clips=n)
shard = engine.CorpusShard(bank, 0, Nv, counts, mom, dev)
sd = synth.model_weights(F, seed=123)
model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
model = model.to(dev).eval()
ops = engine.HipOps()
for Nq in [int(x) for x in sys.argv[1:]] or [1, 64]:
    tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
    own, times = synth.annotations(Nq, counts, seed=123)
    packed = engine.pack_times(times)
    td = tuple(torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev) for x in packed)
    nd = torch.from_numpy(counts[own].astype(np.int32)).to(dev)
    idx = engine.gt_index(shard, own)
    ws = _vfr.topk_workspace(Nq, Nv, k, dev)

    def request():
        with torch.no_grad():
            Q = engine.encode_queries(model, tokens, dev, ops)
            labels = ops.gt_labels(td, nd, [0.5, 0.7], True, dev, Mmax=n * (n + 1) // 2)
            gt = engine.prepare_gt(shard, own, labels, index=idx)
            return engine.corpus_ranks(shard, Q, own, labels, ops, k=k, world=1, workspace=ws, gt=gt)

    for _ in range(3): request()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): request()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    _vfr.set_option("profile", 1); _vfr.profile_read(reset=True)
    for _ in range(reps): request()
    torch.cuda.synchronize()
    sites = _vfr.profile_read(reset=True); _vfr.set_option("profile", 0)
    print(f"Nq={Nq}: {ms:.3f} ms per request")
    for name, (t, c) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
        print(f"   site {name:18s} {t / reps:7.3f} ms  x{c / reps:.0f}")
