"""Single-video moment localisation with the reference's ``evaluate_single`` surface
(``model/evaluate_single.py:19-87``): each query is ranked against the moments of its OWN video.

The embedding phases and the per-query scoring (``:31-34,44-53``) run batched on the device
(``vfr_score_own_f32``); what follows is the reference's integer bookkeeping on the host.  Quirks mirrored:
the model ordering is the argsort REVERSED, i.e. descending distance (Q1, ``:54``); 'chance' draws
``random.sample`` for every query, requested or not (``:55``); ``prior[num_segments]`` is indexed
unconditionally, so a dict must be passed even for ``['model']`` when it is consulted (Q9, ``:56``).
"""
from __future__ import annotations

import itertools
import random

import numpy as np
import torch

from . import engine
from .evaluate import _drain_queries, embed_corpus
from .utils import get_iou


def get_metrics(recalls):
    return {(name if name == "MR" else f"R@{name}"): (np.median(v) if name == "MR" else np.mean(v) * 100)
            for name, v in recalls.items()}


def evaluate(model, video_iterator, lang_iterator, annotations, device, model_types=['model'], prior=[],
             iou_thresholds=[0.5, 0.7]):
    was_training = model.training
    model.eval()
    ops = engine.ops_for(device)
    shard, names = embed_corpus(model, video_iterator, device, ops)
    video_index = {name: i for i, name in enumerate(names)}
    moments = lang_iterator.batch_sampler.moments

    tokens, q_videos, annot_ids = _drain_queries(lang_iterator)
    rank_hits = {mt: {1: [], 5: [], 10: [], "mIoU": []} for mt in model_types}
    recall_hits = {(mt, thr): {1: [], 5: [], 10: []} for mt, thr in itertools.product(model_types, iou_thresholds)}
    if tokens:
        with torch.no_grad():
            Q = engine.encode_queries(model, torch.cat(tokens), device, ops)
        own = np.asarray([video_index[v] for v in q_videos], np.int64)
        own_t = torch.from_numpy(own.astype(np.int32)).to(device)
        scores = ops.score_own(Q, shard.bank, own_t).cpu().numpy()          # [Nq, Mmax], +inf beyond M
    for q, annot_id in enumerate(annot_ids):
        n = int(shard.counts_all[own[q]])
        spans = moments[n]
        order = np.argsort(scores[q, :len(spans)], kind="stable")
        predicts = {"model": [spans[i] for i in order][::-1],
                    "chance": random.sample(spans, k=len(spans)),
                    "prior": prior[n]}
        times = annotations[annot_id]["times"]
        for mt, hits in rank_hits.items():
            ranking = predicts[mt]
            ranks = sorted(ranking.index(tuple(t)) + 1 for t in times)
            ious = np.sort([get_iou([ranking[0]], t[0], t[1])[0] for t in times])
            for k in hits:
                hits[k].append(np.mean(ious[-3:]) if k == "mIoU" else int(np.mean(ranks[:3]) <= k))
        for (mt, thr), hits in recall_hits.items():
            good = np.array([(get_iou(times, s, e) > thr).sum() >= 2 for s, e in predicts[mt]]).astype(int)
            for k in hits:
                hits[k].append(int(good[:k].sum() > 0))
    model.train(was_training)

    metrics = {mt: {(k if k == "mIoU" else f"Rank@{k}"): np.mean(v) * 100 for k, v in hits.items()}
               for mt, hits in rank_hits.items()}
    for (mt, thr), hits in recall_hits.items():
        metrics[f"{mt}, IoU={thr}"] = {f"Recall@{k}": np.mean(v) * 100 for k, v in hits.items()}
    return metrics
