"""Corpus-wide moment retrieval with the reference's ``evaluate`` surface (``model/evaluate.py:19-90``).

``evaluate(model, video_iterator, lang_iterator, annotations, device, ...)`` consumes the same iterators and
returns the same dict (``{"model, IoU=0.5": {"R@1", "R@10", "R@100", "MR"}, ...}``), but batches internally:
all clips are embedded in one clip-encoder launch sequence, all queries in one BiLSTM pass, and every
query is ranked against every moment of every video by the fused scoring kernel -- the Nq x Nv x M Python
iterations and per-moment ``.item()`` syncs of ``evaluate.py:49-65`` do not exist here.  What the metrics
need is the 0-based rank of the best ground-truth-positive moment (``:77``), which the kernel counts
directly; ``R@k`` is ``rank < k`` (``:80``).

Differences kept on purpose: ties are broken by global moment id (the reference's
unstable argsort leaves them unspecified); ``np.random.choice`` is drawn only when 'chance' is requested
(Q7).  A query without any positive moment raises IndexError as in the reference (Q3).
"""
from __future__ import annotations

import itertools

import numpy as np
import torch

from . import engine


def get_metrics(recalls):
    """lists -> ``{"R@k": mean*100, "MR": median}``."""
    return {(name if name == "MR" else f"R@{name}"): (np.median(v) if name == "MR" else np.mean(v) * 100)
            for name, v in recalls.items()}


def _drain_videos(video_iterator):
    names, feats = [], []
    for batch in video_iterator:
        names.append(batch["video"])
        feats.append(batch["feature"])
    return names, feats


def _drain_queries(lang_iterator):
    tokens, videos, annot_ids = [], [], []
    for batch in lang_iterator:
        tokens.append(batch["feature"])
        videos.append(batch["video"])
        annot_ids.append(batch["annot_id"])
    return tokens, videos, annot_ids


def embed_corpus(model, video_iterator, device, ops, rank=0, world=1, drained=None):
    """-> (CorpusShard, video names).  Uses the dataset's packed FeatureBank (factored clip encoder, one
    launch sequence for the whole shard) when the iterator exposes one; otherwise embeds the per-video
    ``[n, 2F+2]`` tensors the iterator yields, concatenated into one batch."""
    names, feats = drained if drained is not None else _drain_videos(video_iterator)
    dataset = getattr(video_iterator, "dataset", None)
    if hasattr(dataset, "feature_bank"):
        return engine.build_corpus(model, dataset.feature_bank(names), device, ops, rank, world), names
    counts = [int(f.shape[0]) for f in feats]
    with torch.no_grad():
        emb = model(torch.cat(feats).to(device)) if feats else torch.zeros((0, 1))
    return engine.corpus_from_embeddings(emb, counts, device, ops, rank, world), names


def _encode_both(model, video_iterator, drained, tokens, device, ops, rank, world, query_fn=None):
    """Clip shard and query batch; from 4 shards on the two independent encoders run side by side (engine.overlapped)."""
    def queries():
        with torch.no_grad():
            return query_fn() if query_fn else engine.encode_queries(model, torch.cat(tokens), device, ops, rank, world)
    (shard, names), Q = engine.overlapped(device, lambda: embed_corpus(model, video_iterator, device, ops, rank, world, drained),
                                          queries, enable=world >= 4 and len(tokens) > 0)
    return shard, names, Q


def evaluate(model, video_iterator, lang_iterator, annotations, device, preliminary=100, model_types=['model'],
             iou_thresholds=[0.5, 0.7], rank=0, world=1, return_topk=0):
    was_training = model.training
    model.eval()
    ops = engine.ops_for(device)
    drained = _drain_videos(video_iterator)
    tokens, q_videos, annot_ids = _drain_queries(lang_iterator)
    if not tokens:
        embed_corpus(model, video_iterator, device, ops, rank, world, drained)
        model.train(was_training)
        return {f"{mt}, IoU={thr}": get_metrics({1: [], 10: [], 100: [], "MR": []})
                for mt, thr in itertools.product(model_types, iou_thresholds)}
    shard, names, Q = _encode_both(model, video_iterator, drained, tokens, device, ops, rank, world)
    video_index = {name: i for i, name in enumerate(names)}
    own = np.asarray([video_index[v] for v in q_videos], np.int64)
    times = [annotations[a]["times"] for a in annot_ids]
    labels = engine.gt_labels(times, shard.counts_all[own], list(iou_thresholds), True, shard.device, ops)   # a11, on the device
    ranks, top_dist, top_idx = engine.corpus_ranks(shard, Q, own, labels, ops, k=return_topk, world=world)
    ranks = ranks.cpu().numpy()

    recalls = {}
    total = int(shard.mom_off_all[-1])
    first = None
    if "chance" in model_types:
        # evaluate.py:68-72: ONE permutation per query (drawn in query order), shared by every IoU threshold; the chance
        # rank is the first position of the permutation that holds a positive moment
        lab_h = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else labels
        first = np.empty((len(iou_thresholds), len(own)), np.int64)
        for q in range(len(own)):
            perm = np.random.choice(np.arange(total), size=total, replace=False)
            base = int(shard.mom_off_all[own[q]])
            for r in range(len(iou_thresholds)):
                pos = np.nonzero(lab_h[r, q])[0] + base
                hit = np.nonzero(np.isin(perm, pos))[0]
                if not len(hit):
                    raise IndexError("index 0 is out of bounds for axis 0 with size 0 (no ground-truth-positive moment)")
                first[r, q] = hit[0]
    for r, thr in enumerate(iou_thresholds):
        if "model" in model_types:
            recalls[("model", thr)] = {1: (ranks[r] < 1).astype(int), 10: (ranks[r] < 10).astype(int),
                                       100: (ranks[r] < 100).astype(int), "MR": ranks[r]}
        if "chance" in model_types:
            recalls[("chance", thr)] = {1: (first[r] < 1).astype(int), 10: (first[r] < 10).astype(int),
                                        100: (first[r] < 100).astype(int), "MR": first[r]}
    if preliminary and len(own) > preliminary and rank == 0:
        for upto in range(preliminary + 1, len(own) + 1, preliminary):   # the reference prints at li = 100, 200, ... (li + 1 queries)
            print()
            for (mt, thr), rec in recalls.items():
                part = get_metrics({k: v[:upto] for k, v in rec.items()})
                print(f"{mt}, IoU={thr}:\t", "".join(f"{n}: {v:.4f}\t" for n, v in part.items()))
    model.train(was_training)
    out = {f"{mt}, IoU={thr}": get_metrics(recalls[(mt, thr)])
           for mt, thr in itertools.product(model_types, iou_thresholds) if (mt, thr) in recalls}
    if return_topk:
        return out, (top_dist, top_idx)
    return out


def validate_epoch(model, video_iterator, lang_iterator, annotations, device, size=250, iou_thresholds=[0.5, 0.7],
                   atk=[1, 10, 100], bert=False, rank=0, world=1):
    """The retrieval metrics of ``Trainer.validate_epoch`` (``model/main.py:121-212``) through the same fused pass.

    Same arguments as the method (minus ``self``; ``device``/``bert`` are the Trainer attributes it reads) and the
    same numbers, returned instead of written to TensorBoard::

        {"CustomRecall": {"1_IoU05": ..}, "MedianRank": {"IoU05": ..}, "MeanReciprocalRank": {"IoU05": ..},
         "pr_curve": {"precision": {k: [..11..]}, "recall": {k: [..11..]}}}      # pr_curve == {} unless size == -1

    Semantics that differ from ``evaluate`` and are kept: IoU ``>=`` threshold (``:161``), 1-based rank
    (``:169``), only the first ``size`` queries are scored (``:186``; ``-1`` = all, and then the 11-point threshold
    sweep with precision/recall@k, ``:137,178-179,200-206``).  A query without a positive moment raises IndexError.
    """
    was_training = model.training
    model.eval()
    ops = engine.ops_for(device)
    drained = _drain_videos(video_iterator)
    thr_range = [i / 10 for i in range(11)] if size == -1 else list(iou_thresholds)

    tokens, q_videos, annot_ids = _drain_queries(lang_iterator if size <= 0 else itertools.islice(lang_iterator, size))   # :186 never fires for size <= 0
    nq = len(tokens)
    bert_fn = (lambda: model(torch.cat(tokens).to(device), False, device, True).contiguous()) if bert else None
    shard, names, Q = _encode_both(model, video_iterator, drained, tokens, device, ops, rank, world, bert_fn)
    video_index = {name: i for i, name in enumerate(names)}
    own = np.asarray([video_index[v] for v in q_videos], np.int64)
    times = [annotations[a]["times"] for a in annot_ids]
    labels = engine.gt_labels(times, shard.counts_all[own], thr_range, False, shard.device, ops)   # main.py:161: >=
    kmax = max(atk) if size == -1 else 0
    ranks0, _, top_idx = engine.corpus_ranks(shard, Q, own, labels, ops, k=kmax, world=world)
    ranks = ranks0.cpu().numpy() + 1                                              # :169  "+ 1"
    model.train(was_training)

    keep = [r for r, thr in enumerate(thr_range) if thr in iou_thresholds]
    tag = lambda thr: f"IoU0{round(thr * 10)}"
    out = {"CustomRecall": {f"{k}_{tag(thr_range[r])}": float(np.mean(ranks[r] <= k)) for r in keep for k in atk},
           "MedianRank": {tag(thr_range[r]): float(np.median(ranks[r])) for r in keep},
           "MeanReciprocalRank": {tag(thr_range[r]): float(np.mean(1.0 / ranks[r])) for r in keep},
           "pr_curve": {}}
    if size == -1:
        # positives among the top-k: only moments of the query's own video can be positive (:162-166)
        labels = labels.cpu().numpy()
        top = top_idx.cpu().numpy()                                               # [Nq, kmax] global moment ids, -1 padded
        base = shard.mom_off_all[own][:, None]
        local = top - base
        Mown = (shard.counts_all[own] * (shard.counts_all[own] + 1) // 2)[:, None]
        inside = (top >= 0) & (local >= 0) & (local < Mown)
        qsel = np.arange(nq)[:, None]
        hits = labels[:, qsel, np.where(inside, local, 0)] & inside[None]         # [R, Nq, kmax]
        relevant = labels.sum(axis=(1, 2))
        pr = {"precision": {}, "recall": {}}
        for k in atk:
            tp = hits[:, :, :k].sum(axis=(1, 2))
            pr["precision"][k] = [float(t) / (k * nq) for t in tp]
            pr["recall"][k] = [float(t) / float(rel) for t, rel in zip(tp, relevant)]
        out["pr_curve"] = pr
    return out
