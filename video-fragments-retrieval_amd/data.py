"""Data side of the hot path with the reference's ``data`` surface (``model/data.py``).

In scope: clip pooling + L2 normalisation (a3, ``:163-181``) -- on the GPU through
``vfr_segment_pool_norm_*`` --, ``[seg | ctx | tef]`` assembly (a4, ``:204-213``), tokenisation and
``WordIndexer`` (a5, ``:33-118,190-197``) and the evaluation iteration contract (a16, ``:359-418``).
Training-time triplet sampling, the h5 "prep" layout and the BERT tokeniser path are out of scope.

Besides the reference's per-item API, ``CustomDataset.feature_bank()`` exposes the whole corpus as one
packed ``FeatureBank`` (``seg [sum n, F]``, ``ctx [Nv, F]``, CSR offsets) so the evaluators can feed the
kernels at HBM rate instead of one video per call.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch
from torch.utils.data.dataset import Dataset
from torch.utils.data.sampler import BatchSampler

from .utils import generate_moments

SELECT_FPS = 25
FRAMES_PER_SEC = 5
SEC_PER_SEGMENT = 5
FEATURE_DIM = dict(vgg19=4096, resnet152=2048)
POOLING = dict(avg=np.mean, max=np.max)
EMBEDDING_DIM = 100
PAD_TOKEN = "<pad>"
UNK_TOKEN = "<unk>"

_TOKEN_RE = re.compile(r"('\w )|([\w\d]+)")


def tokenize(description: str):
    """Query string -> word list: lower-case, strip trailing newline/space, keep the ``[\\w\\d]+`` runs
    (the second group of the reference's pattern; its first group only swallows ``'s ``-style clitics)."""
    text = description.rstrip("\n ").lower()
    return [m[1] for m in _TOKEN_RE.findall(text) if m[1] != ""]


class WordIndexer:
    """GloVe vocabulary: row 0 is the all-zero pad vector, then the file's rows in order."""

    def __init__(self, emb_path, emb_dim=EMBEDDING_DIM, vocab=None):
        self.pad, self.unk, self.emb_dim = PAD_TOKEN, UNK_TOKEN, emb_dim
        self.item2idx_dict, self.idx2item_dict, self.embedding_dict = {}, {}, {}
        self.add_item(self.pad, [0] * emb_dim)
        with open(Path(emb_path) / f"glove.6B.{emb_dim}d.txt", "r", encoding="UTF-8") as fh:
            for line in fh:
                word, *values = line.strip().split(" ")
                vector = [float(x) for x in values]
                assert len(vector) == emb_dim
                if vocab is None or word in vocab:
                    self.add_item(word, vector)

    def get_items_list(self):
        return list(self.item2idx_dict)

    def get_items_count(self):
        return len(self.item2idx_dict)

    def add_item(self, item, item_vector):
        idx = len(self.item2idx_dict)
        self.item2idx_dict[item] = idx
        self.idx2item_dict[idx] = item
        self.embedding_dict[item] = item_vector
        return idx

    def items2idx(self, item_sequences):
        table = self.item2idx_dict
        # out-of-vocabulary -> the <unk> row; KeyError if the GloVe file has none (as in the reference, Q6)
        return [[table[w] if w in table else table[self.unk] for w in seq] for seq in item_sequences]

    def idx2items(self, idx_sequences):
        return [[self.idx2item_dict[i] for i in seq] for seq in idx_sequences]

    def idx2tensor(self, idx_sequences, align="left", word_len=-1):
        if word_len == -1:
            word_len = max(len(seq) for seq in idx_sequences)
        out = torch.zeros(len(idx_sequences), word_len, dtype=torch.long)
        for row, seq in enumerate(idx_sequences):
            seq = list(seq[:word_len])                      # longer queries are truncated
            if align == "left":
                start = 0
            elif align == "center":
                start = (word_len - len(seq)) // 2
            else:
                raise ValueError("Unknown align string.")
            out[row, start:start + len(seq)] = torch.as_tensor(seq, dtype=torch.long)
        return out

    def items2tensor(self, item_sequences, tensor_size, align="left"):
        return self.idx2tensor(self.items2idx(item_sequences), align, word_len=tensor_size)

    def get_embeddings(self):
        matrix = torch.zeros(self.get_items_count(), self.emb_dim)
        for word, idx in self.item2idx_dict.items():
            matrix[idx] = torch.tensor(self.embedding_dict[word])
        return matrix


@dataclass
class FeatureBank:
    """Packed clip features of a corpus: the HBM layout the clip-encoder kernel streams."""
    videos: list            # names, iteration order == global moment id order
    seg: torch.Tensor       # [sum n, F] fp32, rows L2-normalised
    ctx: torch.Tensor       # [Nv, F] fp32
    clip_off: torch.Tensor  # [Nv+1] int32

    def to(self, device):
        return FeatureBank(self.videos, self.seg.to(device), self.ctx.to(device), self.clip_off.to(device))

    @property
    def counts(self):
        return (self.clip_off[1:] - self.clip_off[:-1]).tolist()


def pool_frames_host(frames: np.ndarray, pooling: str = "avg"):
    """Host (numpy) form of the pooling, used only when no ROCm device is present at dataset build time."""
    step = FRAMES_PER_SEC * SEC_PER_SEGMENT
    nseg = -(-frames.shape[0] // step)
    op = POOLING[pooling]
    rows = []
    for i in range(nseg):
        r = op(frames[i * step:(i + 1) * step], axis=0)
        rows.append(r / (np.linalg.norm(r) + np.float32(1e-5)))
    c = op(frames, axis=0)
    return np.stack(rows).astype(np.float32), (c / (np.linalg.norm(c) + np.float32(1e-5))).astype(np.float32)


class CustomDataset(Dataset):
    """Per-video pooled clip features + per-annotation token tensors (reference constructor signature)."""

    def __init__(self, videos, annotations, ft_directory, ft_type, word_indexer=None, bert_tokenizer=None,
                 bert_model=None, validate=False, max_query_len=20, pooling="avg", prep=False, pool_device=None):
        if prep:
            raise NotImplementedError("the h5 'prep' feature layout (model/data.py:145-161) is out of scope")
        if bert_tokenizer is not None:
            raise NotImplementedError("the BERT tokeniser path (model/data.py:198-202) needs fetched weights")
        self.word_indexer, self.max_query_len = word_indexer, max_query_len
        self.ft_directory, self.ft_type = ft_directory, ft_type
        self.validate, self.pooling = validate, pooling
        self.pool_device = pool_device if pool_device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.num_segments_info, self.video_features, self.lang_features = {}, {}, {}
        self.store = None
        self.load_video_features(videos)
        self.load_lang_features(annotations)

    def load_video_features(self, videos):
        F = FEATURE_DIM[self.ft_type]
        packed = Path(self.ft_directory) / f"features_{self.ft_type}.vfs"
        if packed.exists():                         # packed store written by store.FeatureStore: no pooling, no copies
            from .store import FeatureStore
            self.store = FeatureStore.open(packed)
            if self.store.F != F or self.store.meta["pooling"] != self.pooling:
                raise ValueError(f"{packed}: built for F={self.store.F}, pooling={self.store.meta['pooling']}")
            for video in videos:
                seg_v, ctx_v = self.store.video_rows(video)
                self.video_features[video] = dict(segment_features=seg_v.numpy(), context_features=ctx_v.numpy(),
                                                  num_segments=int(seg_v.shape[0]))
                self.num_segments_info[video] = int(seg_v.shape[0])
            return
        arrays = []
        for video in videos:
            path = Path(self.ft_directory) / f"features_{self.ft_type}" / f"{self.ft_type}_ft_{video}.npy"
            a = np.load(path)
            arrays.append(np.ascontiguousarray(a.reshape(a.shape[0], F), dtype=np.float32))
        if not arrays:
            return
        if str(self.pool_device).startswith("cuda"):
            from . import _vfr
            frames = torch.from_numpy(np.concatenate(arrays)).to(self.pool_device)
            seg, ctx, nseg = _vfr.segment_pool_norm_batch(frames, [a.shape[0] for a in arrays], SELECT_FPS, self.pooling)
            seg, ctx, nseg = seg.cpu().numpy(), ctx.cpu().numpy(), nseg.tolist()
            off = np.concatenate([[0], np.cumsum(nseg)])
            pooled = [(seg[off[i]:off[i + 1]], ctx[i]) for i in range(len(arrays))]
        else:
            pooled = [pool_frames_host(a, self.pooling) for a in arrays]
        for video, (seg_v, ctx_v) in zip(videos, pooled):
            self.video_features[video] = dict(segment_features=seg_v.astype(np.float64), context_features=ctx_v,
                                              num_segments=int(seg_v.shape[0]))
            self.num_segments_info[video] = int(seg_v.shape[0])

    def load_lang_features(self, annotations):
        if self.word_indexer is None:
            return
        for annot_id, info in annotations.items():
            self.lang_features[annot_id] = self.word_indexer.items2tensor([tokenize(info["description"])],
                                                                          self.max_query_len)

    def make_visual_features(self, video, start_t, end_t):
        entry = self.video_features[video]
        n = entry["num_segments"]
        seg = torch.from_numpy(np.asarray(entry["segment_features"][start_t:end_t + 1])).float()
        ctx = torch.from_numpy(np.asarray(entry["context_features"]).reshape(1, -1)).float().expand(seg.size(0), -1)
        t = torch.arange(start_t, end_t + 1, dtype=torch.float32).view(-1, 1)
        return torch.cat([seg, ctx, t / n, (t + 1) / n], dim=1)

    def feature_bank(self, videos=None) -> FeatureBank:
        videos = list(self.video_features) if videos is None else list(videos)
        if getattr(self, "store", None) is not None:
            return self.store.feature_bank(videos)
        seg = np.concatenate([np.asarray(self.video_features[v]["segment_features"], np.float32) for v in videos])
        ctx = np.stack([np.asarray(self.video_features[v]["context_features"], np.float32) for v in videos])
        counts = [self.num_segments_info[v] for v in videos]
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return FeatureBank(videos, torch.from_numpy(seg), torch.from_numpy(ctx), torch.from_numpy(off))

    def __getitem__(self, sample):
        if not self.validate:
            raise NotImplementedError("training triplets (model/data.py:232-246) are out of scope")
        if "annotation_id" in sample:
            return dict(features=self.lang_features[sample["annotation_id"]], video=sample["video_pos"],
                        annot_id=sample["annotation_id"])
        return dict(features=self.make_visual_features(sample["video_pos"], sample["start_t"], sample["end_t"]),
                    video=sample["video_pos"])


class VideoBatchSampler(BatchSampler):
    """One whole video per batch, in the given order."""

    def __init__(self, videos, num_segments_info):
        self.videos, self.num_segments_info = videos, num_segments_info

    def __iter__(self):
        for video in self.videos:
            yield [dict(video_pos=video, start_t=0, end_t=self.num_segments_info[video] - 1)]

    def __len__(self):
        return len(self.videos)


class LanguageBatchSampler(BatchSampler):
    """One annotation per batch; carries the moment tables the evaluators index by clip count.

    The reference builds the table for 0..6 clips only (Q5); ``max_segments`` widens it (21-clip runs)."""

    def __init__(self, annotations, num_segments_info, max_segments=None):
        self.annotations, self.num_segments_info = annotations, num_segments_info
        top = max([6] + list(num_segments_info.values())) if max_segments is None else max_segments
        self.moments = {n: generate_moments(n) for n in range(top + 1)}

    def get_annotations(self, annot_id):
        info = self.annotations[annot_id]
        spans = np.array(info["times"])
        return spans[spans[:, 1] <= self.num_segments_info[info["video"]]]

    def __iter__(self):
        for annot_id, info in list(self.annotations.items()):
            yield [dict(annotation_id=annot_id, video_pos=info["video"])]

    def __len__(self):
        return len(self.annotations)


def validate_collate(batch):
    item = batch[0]
    return dict(feature=item["features"], video=item["video"], annot_id=item.get("annot_id", []))
