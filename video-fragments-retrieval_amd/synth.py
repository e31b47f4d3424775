"""Deterministic synthetic DiDeMo-shaped inputs and CALModel weights.

Everything is drawn from ``np.random.RandomState(seed)`` -- the legacy MT19937 stream is frozen across
numpy versions, so the CPU container, the GPU box and the committed golden fixtures all see the same
numbers without shipping any arrays.  Shapes follow the reference:

* clip features: ``seg [sum n, F]`` rows L2-normalised with +1e-5, ``ctx [Nv, F]`` = normalised mean of
  the raw rows (what ``model/data.py:173-181`` produces from a ``[T, 4096]`` fc7 array);
* queries: ``int64 [Nq, 20]`` left-aligned, 0-padded token ids (``model/data.py:87-104``);
* weights: ``state_dict`` of ``models.CALModel`` (key names/shapes of ``model/models.py:21-48``);
* annotations: 4 ``(s, e)`` spans per query, two of them identical so a GT-positive moment exists
  (avoids the IndexError of ``model/evaluate.py:77`` when nothing is positive).
"""
from __future__ import annotations

import numpy as np

MAX_QUERY_LEN = 20
EMB_DIM = 100


def clip_counts(num_videos: int, clips, seed: int = 123) -> np.ndarray:
    """``clips`` = int (uniform) or 'didemo' (86 % six-clip / 14 % five-clip, didemo_video_info.json)."""
    if isinstance(clips, str):
        if clips != "didemo":
            raise ValueError(f"unknown clip layout {clips!r}")
        rs = np.random.RandomState(seed + 7)
        return np.where(rs.rand(num_videos) < 0.86, 6, 5).astype(np.int32)
    return np.full(num_videos, int(clips), dtype=np.int32)


def clip_offsets(counts) -> np.ndarray:
    return np.concatenate([[0], np.cumsum(np.asarray(counts, np.int64))]).astype(np.int32)


def video_features(counts, feat_dim: int = 4096, seed: int = 123):
    """-> (seg [sum n, F] f32, ctx [Nv, F] f32).  Non-negative (post-ReLU-like) rows."""
    rs = np.random.RandomState(seed)
    counts = np.asarray(counts)
    off = clip_offsets(counts)
    raw = rs.rand(int(off[-1]), feat_dim).astype(np.float32)
    ctx = np.empty((len(counts), feat_dim), np.float32)
    for v in range(len(counts)):
        m = raw[off[v]:off[v + 1]].mean(axis=0)
        ctx[v] = m / (np.linalg.norm(m) + np.float32(1e-5))
    seg = raw / (np.linalg.norm(raw, axis=1, keepdims=True) + np.float32(1e-5))
    return seg.astype(np.float32), ctx


def query_tokens(num_queries: int, vocab: int = 400, seed: int = 123, min_len: int = 2, max_len: int = 14):
    rs = np.random.RandomState(seed + 1)
    tok = np.zeros((num_queries, MAX_QUERY_LEN), np.int64)
    lens = rs.randint(min_len, max_len + 1, size=num_queries)
    for i, n in enumerate(lens):
        tok[i, :n] = rs.randint(1, vocab, size=n)
    return tok


def annotations(num_queries: int, counts, seed: int = 123):
    """-> (own_video [Nq] int32, times [Nq][4][2] python ints)."""
    rs = np.random.RandomState(seed + 2)
    counts = np.asarray(counts)
    own = rs.randint(0, len(counts), size=num_queries).astype(np.int32)
    times = []
    for q in range(num_queries):
        n = int(counts[own[q]])
        spans = []
        for _ in range(3):
            s = int(rs.randint(0, n))
            e = int(rs.randint(s, n))
            spans.append([s, e])
        spans.append(list(spans[0]))
        times.append(spans)
    return own, times


def model_weights(feat_dim: int = 4096, vocab: int = 400, hidden: int = 1000, mlp_hidden: int = 500,
                  emb_dim: int = EMB_DIM, seed: int = 123, random_bias: bool = True, normalize_lang: bool = False):
    """``state_dict``-shaped dict of float32 numpy arrays for ``models.CALModel(2*feat_dim+2, emb)``.

    Linear: U(-0.08, 0.08) like ``models.init_weights`` (biases small random unless ``random_bias`` is
    False, the reference zeroes them; random exercises the bias path in parity tests).  LSTM:
    U(-1/sqrt(H), 1/sqrt(H)) (torch default).  Embedding N(0,1) with the pad row zero.
    """
    rs = np.random.RandomState(seed + 3)
    u = lambda *shape, a=0.08: rs.uniform(-a, a, size=shape).astype(np.float32)
    k = 1.0 / np.sqrt(hidden)
    sd = {
        "visual_fc.0.weight": u(mlp_hidden, 2 * feat_dim + 2),
        "visual_fc.0.bias": u(mlp_hidden) if random_bias else np.zeros(mlp_hidden, np.float32),
        "visual_fc.2.weight": u(emb_dim, mlp_hidden),
        "visual_fc.2.bias": u(emb_dim) if random_bias else np.zeros(emb_dim, np.float32),
    }
    emb = rs.randn(vocab, emb_dim).astype(np.float32)
    emb[0] = 0.0
    sd["word_embedding.weight"] = emb
    if normalize_lang:
        ll = (1.0 + 0.25 * rs.randn(vocab, 1)).astype(np.float32)
        ll[0] = 1.0
        sd["learnable_length.weight"] = ll
    for suffix in ("", "_reverse"):
        sd[f"lstm.weight_ih_l0{suffix}"] = u(4 * hidden, emb_dim, a=k)
        sd[f"lstm.weight_hh_l0{suffix}"] = u(4 * hidden, hidden, a=k)
        sd[f"lstm.bias_ih_l0{suffix}"] = u(4 * hidden, a=k)
        sd[f"lstm.bias_hh_l0{suffix}"] = u(4 * hidden, a=k)
    sd["lang_fc.weight"] = u(emb_dim, 2 * hidden)
    sd["lang_fc.bias"] = u(emb_dim) if random_bias else np.zeros(emb_dim, np.float32)
    return sd


def frames_u8(num_frames: int, height: int, width: int, seed: int = 123) -> np.ndarray:
    rs = np.random.RandomState(seed + 4)
    return rs.randint(0, 256, size=(num_frames, height, width, 3)).astype(np.uint8)


def vgg_weights(cfg, in_hw, fc_dim: int, seed: int = 123):
    """Random weights for a VGG-"E"-shaped stack ``cfg`` (ints = conv widths, 'M' = maxpool)."""
    rs = np.random.RandomState(seed + 5)
    conv_w, conv_b = [], []
    cin = 3
    for item in cfg:
        if item == "M":
            continue
        std = np.sqrt(2.0 / (cin * 9))
        conv_w.append((rs.randn(item, cin, 3, 3) * std).astype(np.float32))
        conv_b.append((rs.randn(item) * 0.05).astype(np.float32))
        cin = item
    k6 = cin * 49
    fc6 = ((rs.randn(fc_dim, k6) * np.sqrt(2.0 / k6)).astype(np.float32), (rs.randn(fc_dim) * 0.05).astype(np.float32))
    fc7 = ((rs.randn(fc_dim, fc_dim) * np.sqrt(2.0 / fc_dim)).astype(np.float32),
           (rs.randn(fc_dim) * 0.05).astype(np.float32))
    return conv_w, conv_b, fc6, fc7


def resnet_weights(blocks=(3, 8, 36, 3), width: int = 64, seed: int = 123):
    """Random torchvision-style ``state_dict`` (numpy float32) for a Bottleneck ResNet (``resnet152`` = blocks (3, 8, 36, 3),
    width 64): He-scaled convolutions, BatchNorm running statistics away from (0, 1), and a small gamma on every block's last
    BatchNorm so 50 residual additions with random weights stay O(1)."""
    rs = np.random.RandomState(seed + 6)
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = (rs.randn(cout, cin, k, k) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)

    def bn(name, c, gamma_scale=1.0):
        sd[name + ".weight"] = (rs.uniform(0.5, 1.0, c) * gamma_scale).astype(np.float32)
        sd[name + ".bias"] = (rs.randn(c) * 0.05).astype(np.float32)
        sd[name + ".running_mean"] = (rs.randn(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)

    conv("conv1", width, 3, 7); bn("bn1", width)
    cin = width
    for li, nb in enumerate(blocks):
        mid = width * 2 ** li
        for b in range(nb):
            pre = f"layer{li + 1}.{b}"
            conv(pre + ".conv1", mid, cin, 1); bn(pre + ".bn1", mid)
            conv(pre + ".conv2", mid, mid, 3); bn(pre + ".bn2", mid)
            conv(pre + ".conv3", 4 * mid, mid, 1); bn(pre + ".bn3", 4 * mid, 0.3)
            if b == 0:
                conv(pre + ".downsample.0", 4 * mid, cin, 1); bn(pre + ".downsample.1", 4 * mid)
            cin = 4 * mid
    return sd


def ranking_batch(seed: int, P: int = 96, Nn: int = 80, S: int = 16, D: int = 100):
    """Seeded stand-in for one triplet batch of ``Trainer.train_epoch`` (``model/main.py:48-61``): embeddings ~ N(0, 0.1),
    masks = sorted sample ids covering every sample.  -> posit [P,D], intra [Nn,D], inter [P,D], lang [S,D], maskp, maskn."""
    rs = np.random.RandomState(seed)
    maskp = np.sort(np.concatenate([np.arange(S), rs.randint(0, S, P - S)])).astype(np.int64)
    maskn = np.sort(np.concatenate([np.arange(S), rs.randint(0, S, Nn - S)])).astype(np.int64)
    f = lambda n: (rs.randn(n, D) * 0.1).astype(np.float32)
    return f(P), f(Nn), f(P), f(S), maskp, maskn
