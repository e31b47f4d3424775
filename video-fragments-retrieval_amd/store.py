"""Packed clip-feature store: the step between the extractor's ``.npy`` files and the clip encoder.

The reference writes one ``[T, F]`` ``.npy`` per video (``get_rgb_features.py:150-151``) and re-pools every file
at dataset construction (``model/data.py:163-181``); its h5 "prep" layout (``model/data.py:145-161``) stores the
pooled rows per video.  This store keeps the same pooled rows -- clip rows ``seg [sum n, F]`` and one context row
per video ``ctx [Nv, F]`` (the ``n + 1`` rows per video of SURVEY.md 8f row 3), CSR offsets -- in ONE file whose
sections are page-aligned, so that

* opening it is an ``mmap`` (no parsing, no per-video allocation),
* ``feature_bank()`` hands the evaluators zero-copy views in exactly the layout the clip-encoder kernel streams,
* ``pin()`` moves the sections into page-locked host memory once, after which ``engine.build_corpus`` feeds the
  GPU with chunked asynchronous H2D copies that overlap the clip encoder (``engine.encode_clips_streamed``).

File layout (little endian)::

    [0:8)    b"VFRSTORE"      [8:12) u32 version (1)      [12:16) u32 length of the JSON header
    [16:..)  JSON  {"F", "Nv", "C", "ft_type", "pooling", "videos": [...], "sections": {name: [offset, bytes]}}
    4096-aligned sections:  clip_off  int32 [Nv+1]  |  ctx  f32 [Nv, F]  |  seg  f32 [C, F]

Pooling is done by ``vfr_segment_pool_norm_*`` on the GPU when one is present (the same kernels the dataset
uses), by ``data.pool_frames_host`` otherwise.
"""
from __future__ import annotations

import json
import struct
from pathlib import Path

import numpy as np
import torch

from . import data as vdata

MAGIC = b"VFRSTORE"
VERSION = 1
ALIGN = 4096


def _align(x: int) -> int:
    return -(-x // ALIGN) * ALIGN


def _layout(videos, F, Nv, C, ft_type, pooling):
    """-> (header bytes, sections dict); iterates because the header's own length moves the section offsets."""
    sizes = (("clip_off", 4 * (Nv + 1)), ("ctx", 4 * Nv * F), ("seg", 4 * C * F))
    start = ALIGN
    while True:
        sections, pos = {}, start
        for name, nbytes in sizes:
            sections[name] = [pos, nbytes]
            pos = _align(pos + nbytes)
        head = json.dumps(dict(F=F, Nv=Nv, C=C, ft_type=ft_type, pooling=pooling, videos=list(videos),
                               sections=sections)).encode("utf-8")
        if 16 + len(head) <= start:
            return head, sections, pos
        start = _align(16 + len(head))


class FeatureStore:
    """Read side: mmap-backed (optionally pinned) pooled features of a corpus."""

    def __init__(self, path, meta, clip_off, ctx, seg):
        self.path, self.meta = Path(path), meta
        self.videos = list(meta["videos"])
        self.F, self.Nv, self.C = int(meta["F"]), int(meta["Nv"]), int(meta["C"])
        self.clip_off, self.ctx, self.seg = clip_off, ctx, seg          # torch views, host
        self._index = None

    # ------------------------------------------------------------------ writing
    @staticmethod
    def write(path, videos, clip_off, ctx, seg, ft_type="vgg19", pooling="avg"):
        """Write pooled rows (numpy or torch, host) as a store file; returns the path."""
        clip_off = np.ascontiguousarray(np.asarray(clip_off), dtype=np.int32)
        ctx = np.ascontiguousarray(np.asarray(ctx), dtype=np.float32)
        seg = np.ascontiguousarray(np.asarray(seg), dtype=np.float32)
        Nv, F = ctx.shape if ctx.ndim == 2 else (0, int(seg.shape[1]) if seg.ndim == 2 else 0)
        C = int(seg.shape[0])
        if len(videos) != Nv or clip_off.shape != (Nv + 1,) or (Nv and int(clip_off[-1]) != C) or \
                (C and seg.shape[1] != F) or len(set(videos)) != Nv:
            raise ValueError("FeatureStore.write: inconsistent shapes or duplicate video names")
        head, sections, total = _layout(videos, F, Nv, C, ft_type, pooling)
        with open(path, "wb") as fh:
            fh.write(MAGIC + struct.pack("<II", VERSION, len(head)) + head)
            for name, arr in (("clip_off", clip_off), ("ctx", ctx), ("seg", seg)):
                fh.seek(sections[name][0])
                fh.write(arr.tobytes())
            fh.truncate(total)
        return Path(path)

    @classmethod
    def from_npy(cls, path, videos, ft_directory, ft_type="vgg19", pooling="avg", pool_device=None,
                 chunk_videos=512):
        """Pool the extractor's per-video ``.npy`` files (``get_rgb_features.py:150-151`` naming, the same files
        ``model/data.py:164`` loads) into a store, ``chunk_videos`` files at a time."""
        F = vdata.FEATURE_DIM[ft_type]
        dev = pool_device if pool_device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        videos = list(videos)
        segs, ctxs, counts = [], [], []
        for lo in range(0, len(videos), chunk_videos):
            arrays = []
            for video in videos[lo:lo + chunk_videos]:
                a = np.load(Path(ft_directory) / f"features_{ft_type}" / f"{ft_type}_ft_{video}.npy")
                arrays.append(np.ascontiguousarray(a.reshape(a.shape[0], F), dtype=np.float32))
            if str(dev).startswith("cuda"):
                from . import _vfr
                frames = torch.from_numpy(np.concatenate(arrays)).to(dev)
                seg, ctx, nseg = _vfr.segment_pool_norm_batch(frames, [a.shape[0] for a in arrays],
                                                              vdata.SELECT_FPS, pooling)
                segs.append(seg.cpu().numpy()); ctxs.append(ctx.cpu().numpy()); counts += nseg.tolist()
            else:
                for a in arrays:
                    s, c = vdata.pool_frames_host(a, pooling)
                    segs.append(s); ctxs.append(c[None]); counts.append(s.shape[0])
        seg = np.concatenate(segs) if segs else np.zeros((0, F), np.float32)
        ctx = np.concatenate(ctxs) if ctxs else np.zeros((0, F), np.float32)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        cls.write(path, videos, off, ctx, seg, ft_type, pooling)
        return cls.open(path)

    # ------------------------------------------------------------------ reading
    @classmethod
    def open(cls, path, pin=False):
        path = Path(path)
        with open(path, "rb") as fh:
            fixed = fh.read(16)
            if len(fixed) != 16 or fixed[:8] != MAGIC:
                raise ValueError(f"{path}: not a feature store")
            version, hlen = struct.unpack("<II", fixed[8:])
            if version != VERSION:
                raise ValueError(f"{path}: store version {version}, this build reads {VERSION}")
            meta = json.loads(fh.read(hlen).decode("utf-8"))
        size = path.stat().st_size
        Nv, F, C = int(meta["Nv"]), int(meta["F"]), int(meta["C"])
        want = dict(clip_off=4 * (Nv + 1), ctx=4 * Nv * F, seg=4 * C * F)
        for name, (off, nbytes) in meta["sections"].items():
            if nbytes != want.get(name) or off % ALIGN or off + nbytes > size:
                raise ValueError(f"{path}: section {name} is truncated or inconsistent with the header")
        raw = np.memmap(path, dtype=np.uint8, mode="r")

        def view(name, dtype, shape):
            off, nbytes = meta["sections"][name]
            arr = raw[off:off + nbytes].view(dtype).reshape(shape)
            return torch.from_numpy(np.asarray(arr)) if nbytes else torch.zeros(shape, dtype=getattr(torch, np.dtype(dtype).name))

        import warnings
        with warnings.catch_warnings():                     # read-only mmap: torch warns about writability; never written
            warnings.simplefilter("ignore")
            store = cls(path, meta, view("clip_off", np.int32, (Nv + 1,)), view("ctx", np.float32, (Nv, F)),
                        view("seg", np.float32, (C, F)))
        off = store.clip_off.numpy()
        if off[0] != 0 or (np.diff(off) < 0).any() or int(off[-1]) != C:
            raise ValueError(f"{path}: clip offsets are not a CSR of {C} rows")
        return store.pin() if pin else store

    def pin(self):
        """Copy the sections into page-locked host memory (needs a ROCm device); idempotent."""
        if not self.seg.is_pinned():
            self.seg, self.ctx = self.seg.clone().pin_memory(), self.ctx.clone().pin_memory()
            self.clip_off = self.clip_off.clone()
        return self

    @property
    def counts(self):
        return np.diff(self.clip_off.numpy()).astype(np.int64)

    @property
    def num_segments_info(self):
        return dict(zip(self.videos, self.counts.tolist()))

    def video_rows(self, video):
        """-> (seg [n, F], ctx [F]) views of one video (the per-item API of ``CustomDataset``)."""
        if self._index is None:
            self._index = {v: i for i, v in enumerate(self.videos)}
        i = self._index[video]
        off = self.clip_off
        return self.seg[int(off[i]):int(off[i + 1])], self.ctx[i]

    def feature_bank(self, videos=None) -> "vdata.FeatureBank":
        """Packed bank in the given video order; zero-copy when that is the store's own order."""
        if videos is None or list(videos) == self.videos:
            return vdata.FeatureBank(self.videos, self.seg, self.ctx, self.clip_off)
        if self._index is None:
            self._index = {v: i for i, v in enumerate(self.videos)}
        ids = np.asarray([self._index[v] for v in videos], np.int64)
        off = self.clip_off.numpy().astype(np.int64)
        counts = off[ids + 1] - off[ids]
        rows = np.concatenate([np.arange(off[i], off[i + 1]) for i in ids]) if len(ids) else np.zeros(0, np.int64)
        new_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return vdata.FeatureBank(list(videos), self.seg[torch.from_numpy(rows)], self.ctx[torch.from_numpy(ids)],
                                 torch.from_numpy(new_off))
