#!/usr/bin/env python3
"""Rank 0's step of the N-GPU strong-scaling bench, alone on one GPU, WITH the exchange-side compute: engine's
collectives are replaced by local stand-ins (all_gather = N copies of the own tensor, all_reduce = no-op), so
everything except wire time is timed: sample pass, merges of N lists, seeded main pass, key packing.
The duplicated sample keeps the seed at the same quantile as the real global sample (k/N-th best of 256/N videos).
usage: rank_sim.py [N] [reps] [overlap 0|1] [lstm_tile]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, engine, models, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
OVERLAP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 4:
    _vfr.set_option("lstm_tile", int(sys.argv[4]))
for _item in filter(None, __import__("os").environ.get("VFR_OPTS", "").split(",")):     # e.g. VFR_OPTS=score_tasks=4096
    _vfr.set_option(_item.split("=")[0], int(_item.split("=")[1]))
Nv_all, Nq, n, F, k = 10000, 5000, 21, 4096, 100
dev = torch.device("cuda:0")


class FakeDist:
    """Stand-in collectives without wire time AND without host work the real ones do not have: a gather is one broadcast
    copy of the own row into the N slots; the best-GT keys of queries owned by other ranks (KEY_INF here, finite after the
    real MIN over ranks) are replaced by a typical key computed ONCE, with a single elementwise kernel per call."""
    class ReduceOp:
        SUM, MIN, MAX = "sum", "min", "max"
    calls = 0
    typical = None

    @classmethod
    def patch(cls, gtk):
        if cls.typical is None:
            cls.typical = gtk[gtk != engine.KEY_INF].median().clone()
        torch.where(gtk == engine.KEY_INF, cls.typical, gtk, out=gtk)

    def all_gather_into_tensor(self, out, t):
        FakeDist.calls += 1
        if t.dim() == 1 and t.numel() == Nq * k + 2 * Nq:
            if FakeDist.rows % 2 == 0:             # first exchange of sharded_search_fused: [sample keys | best-GT keys]
                self.patch(t[Nq * k:])
            FakeDist.rows += 1
        out.view(N, -1).copy_(t.reshape(1, -1).expand(N, -1))

    def all_reduce(self, t, op=None):
        FakeDist.calls += 1
        if op == "min" and t.dtype == torch.int64:
            self.patch(t)

    def barrier(self):
        pass


FakeDist.rows = 0
engine._dist = lambda: FakeDist()
counts_all = synth.clip_counts(Nv_all, n, seed=123)
off_all = np.concatenate([[0], np.cumsum(counts_all.astype(np.int64))])
mom_all = np.concatenate([[0], np.cumsum(counts_all.astype(np.int64) * (counts_all + 1) // 2)])
lo, hi = engine.shard_range(Nv_all, 0, N)
C = int(off_all[hi] - off_all[lo])
g = torch.Generator(device=dev).manual_seed(1234)
raw = torch.rand((C, F), generator=g, device=dev)
seg = raw / (raw.norm(dim=1, keepdim=True) + 1e-5)
ctx = raw.view(hi - lo, n, F).mean(1); ctx = ctx / (ctx.norm(dim=1, keepdim=True) + 1e-5)
del raw
clip_off = torch.from_numpy((off_all[lo:hi + 1] - off_all[lo]).astype(np.int32)).to(dev)
tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
own, times = synth.annotations(Nq, counts_all, seed=123)
sd = synth.model_weights(F, seed=123)
model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
model = model.to(dev).eval()
ops = engine.HipOps()
labels = engine.gt_label_table(times, counts_all[own], [0.5, 0.7])


def make_shard(emb):
    bank = _vfr.VideoBank(emb, clip_off, 0, max_clips=n, total_moments=int(mom_all[hi] - mom_all[lo]), min_clips=n)
    return engine.CorpusShard(bank, lo, hi, counts_all, mom_all, dev)


with torch.no_grad():
    gt = engine.prepare_gt(make_shard(model.encode_clips(seg, ctx, clip_off)), own, labels)
ws = _vfr.topk_workspace(Nq, hi - lo, k, dev)
marks = {}


def step(split=False):
    with torch.no_grad():
        t = [time.perf_counter()]
        def mark():
            if split:
                torch.cuda.synchronize(); t.append(time.perf_counter())
        if OVERLAP and not split:
            emb, Q = engine.overlapped(dev, lambda: model.encode_clips(seg, ctx, clip_off),
                                       lambda: engine.encode_queries(model, tokens, dev, ops, 0, N))
            shard = make_shard(emb)
        else:
            shard = make_shard(model.encode_clips(seg, ctx, clip_off)); mark()
            Q = engine.encode_queries(model, tokens, dev, ops, 0, N); mark()
        out = engine.corpus_ranks(shard, Q, own, labels, ops, k=k, world=N, workspace=ws, gt=gt); mark()
        if split:
            for name, a, b in zip(("clip", "query", "score+exchange"), t, t[1:]):
                marks.setdefault(name, []).append((b - a) * 1e3)
        return out


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
FakeDist.calls = 0
step(); print("collective calls per step:", FakeDist.calls)
for _ in range(reps):
    step(split=True)
_vfr.set_option("profile", 1); _vfr.profile_read(reset=True)
for _ in range(reps):
    step()
torch.cuda.synchronize()
sites = _vfr.profile_read(reset=True); _vfr.set_option("profile", 0)
ONE = float(__import__("os").environ.get("VFR_ONE_GPU_MS", "23.62"))      # measured 1-GPU step (profiles/r3f_bench_unprofiled.json)
print(f"N={N}: rank-0 step without wire time {ms:.3f} ms   (1-GPU step {ONE} ms / N = {ONE / N:.3f} ms)  -> speed-up bound {ONE / ms:.2f}x")
for name, v in marks.items():
    print(f"  {name:16s} {min(v):7.3f} ms (synchronised)")
tot = 0.0
for name, (t, c) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
    print(f"  site {name:18s} {t / reps:7.3f} ms/step  x{c / reps:.0f}"); tot += t / reps
print(f"  kernel sites total {tot:.3f} ms/step")
