"""Shared builders for the test-suite: synthetic problems and an in-memory dataset with the reference's
iteration contract (``model/data.py:216-231,359-418``)."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from vfr_amd import data as vdata
from vfr_amd import models as vmodels
from vfr_amd import synth


class MemoryDataset(torch.utils.data.Dataset):
    """What ``data.CustomDataset`` holds after loading, filled from arrays (no files)."""

    def __init__(self, seg, ctx, counts, tokens, own, times):
        self.validate = True
        off = synth.clip_offsets(counts)
        self.videos = [f"v{v:05d}" for v in range(len(counts))]
        self.video_features, self.num_segments_info, self.lang_features = {}, {}, {}
        for v, name in enumerate(self.videos):
            self.video_features[name] = dict(segment_features=seg[off[v]:off[v + 1]].astype(np.float64),
                                             context_features=ctx[v], num_segments=int(counts[v]))
            self.num_segments_info[name] = int(counts[v])
        self.annotations = {}
        for q in range(tokens.shape[0]):
            self.lang_features[q] = torch.from_numpy(tokens[q:q + 1])
            self.annotations[q] = dict(video=self.videos[int(own[q])], description="", times=times[q])

    make_visual_features = vdata.CustomDataset.make_visual_features
    feature_bank = vdata.CustomDataset.feature_bank
    __getitem__ = vdata.CustomDataset.__getitem__

    def iterators(self):
        vi = DataLoader(self, shuffle=False, collate_fn=vdata.validate_collate,
                        batch_sampler=vdata.VideoBatchSampler(self.videos, self.num_segments_info))
        li = DataLoader(self, shuffle=False, collate_fn=vdata.validate_collate,
                        batch_sampler=vdata.LanguageBatchSampler(self.annotations, self.num_segments_info))
        return vi, li


def problem(nv, nq, clips, feat_dim=4096, seed=123, hidden=1000, vocab=400):
    counts = synth.clip_counts(nv, clips, seed=seed)
    seg, ctx = synth.video_features(counts, feat_dim, seed=seed)
    tokens = synth.query_tokens(nq, vocab=vocab, seed=seed)
    own, times = synth.annotations(nq, counts, seed=seed)
    sd = synth.model_weights(feat_dim, vocab=vocab, hidden=hidden, seed=seed)
    return dict(counts=counts, off=synth.clip_offsets(counts), seg=seg, ctx=ctx, tokens=tokens, own=own, times=times, sd=sd)


def make_model(sd, feat_dim=4096, hidden=1000, normalize_lang=False):
    m = vmodels.CALModel(2 * feat_dim + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]),
                         hidden_size=hidden, normalize_lang=normalize_lang)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


def lstm_of(sd):
    return {k[len("lstm."):]: v for k, v in sd.items() if k.startswith("lstm.")}


class ThreadRanks:
    """N ranks as N threads of this process sharing one device: the collectives engine.py uses (all_gather, all_reduce,
    barrier), implemented with a thread barrier.  Lets the world > 1 code path run through the real kernels on a one-GPU
    box (all threads launch on the same stream, so host-side launch order is device order)."""

    class ReduceOp:
        SUM, MIN, MAX = "sum", "min", "max"

    def __init__(self, world):
        import threading
        self.world, self.slots = world, [None] * world
        self.bar = threading.Barrier(world)
        self.tls = threading.local()

    def _rank(self):
        return self.tls.rank

    def barrier(self):
        self.bar.wait()

    def all_gather(self, parts, t):
        self.slots[self._rank()] = t
        self.bar.wait()
        for p, s in zip(parts, list(self.slots)):
            p.copy_(s)
        self.bar.wait()

    def all_reduce(self, t, op=None):
        self.slots[self._rank()] = t.clone()
        self.bar.wait()
        allv = torch.stack(list(self.slots))
        red = allv.sum(0) if op == "sum" else (allv.min(0).values if op == "min" else allv.max(0).values)
        self.bar.wait()
        t.copy_(red)

    def run(self, fn):
        """fn(rank, world) on every rank; returns the list of results, re-raising the first failure."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            self.tls.rank = r
            try:
                out[r] = fn(r, self.world)
            except BaseException as e:                      # noqa: BLE001 -- reported to the caller below
                err[r] = e
                self.bar.abort()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out
