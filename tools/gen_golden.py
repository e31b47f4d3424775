#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE itself (this container only).

The reference (``/root/reference``, read-only, never copied) is imported with one absent and unused
dependency stubbed (``h5py``, ``model/data.py:8``; only touched under ``prep=True``).  Inputs and
weights come from the seeded recipes in ``vfr_amd.synth`` so the fixtures hold only the reference's
OUTPUTS plus the recipe parameters; tests regenerate the inputs from the same seeds.

    python tools/gen_golden.py            # writes tests/golden/*.npz, tokens.json

Fixtures:  G1 encoders, G2 scoring + both evaluate() dicts (n=6, n=21, ragged 5/6),
G3 generate_moments / get_iou, G4 load_video_features pooling, G5 tokeniser + WordIndexer, G6 validate_epoch,
G7 ranking loss, G8 evaluate() with the 'chance' baseline, G9 one full-size VGG-19 frame through torch.nn modules,
G10 encoder gradients (loss.backward() through CALModel), G11 DiDeMoDataset.__getitem__ of get_rgb_features.py (frame selection +
normalisation) with torchvision.io.read_video stubbed by a seeded frame generator, G12 two full-size frames through a
ResNet-152 built from torch.nn modules.
"""
import json
import random
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, str(REF / "model"))

import data as ref_data  # noqa: E402  (reference)
import evaluate as ref_evaluate  # noqa: E402
import evaluate_single as ref_evaluate_single  # noqa: E402
import models as ref_models  # noqa: E402
import utils as ref_utils  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from vfr_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(8)


def ref_model(sd_np, feat_dim, normalize_lang=False):
    emb = torch.from_numpy(sd_np["word_embedding.weight"])
    m = ref_models.CALModel(2 * feat_dim + 2, pretrained_emb=emb, normalize_lang=normalize_lang)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    return m.eval()


def make_dataset(seg, ctx, off, tokens, own, times):
    """A reference CustomDataset filled in memory (no files): what load_video_features /
    load_lang_features would have produced (model/data.py:183-188,196-197)."""
    ds = ref_data.CustomDataset.__new__(ref_data.CustomDataset)
    ds.validate = True
    ds.video_features, ds.num_segments_info, ds.lang_features = {}, {}, {}
    videos = [f"v{v:05d}" for v in range(len(off) - 1)]
    for v, name in enumerate(videos):
        n = int(off[v + 1] - off[v])
        ds.video_features[name] = dict(segment_features=seg[off[v]:off[v + 1]].astype(np.float64),
                                       context_features=ctx[v], num_segments=n)
        ds.num_segments_info[name] = n
    annots = {}
    for q in range(tokens.shape[0]):
        ds.lang_features[q] = torch.from_numpy(tokens[q:q + 1])
        annots[q] = dict(video=videos[own[q]], description="", times=times[q])
    return ds, videos, annots


def g1_encoders():
    feat_dim, counts = 4096, np.array([5, 6, 21, 6, 5, 21, 6, 6], np.int32)
    seg, ctx = synth.video_features(counts, feat_dim, seed=11)
    off = synth.clip_offsets(counts)
    tokens = synth.query_tokens(16, seed=11)
    tokens[0] = 0                                   # all-pad query
    tokens[1] = 0; tokens[1, 0] = 7                 # length 1
    tokens[2] = np.arange(1, 21)                    # full length 20
    out = {"counts": counts, "tokens": tokens}
    for tag, nl in (("", False), ("_normlang", True)):
        sd = synth.model_weights(feat_dim, seed=11, normalize_lang=nl)
        m = ref_model(sd, feat_dim, nl)
        ds, videos, _ = make_dataset(seg, ctx, off, tokens, np.zeros(16, int), [[[0, 0]] * 4] * 16)
        with torch.no_grad():
            vis = torch.cat([m(ds.make_visual_features(v, 0, ds.num_segments_info[v] - 1)) for v in videos])
            lang = m(torch.from_numpy(tokens), False, "cpu")
        out["visual_emb" + tag] = vis.numpy()
        out["query_emb" + tag] = lang.numpy()
    # BERT branch (models.py:31,58-59): Linear(768 -> 100) on a pooled vector
    rs = np.random.RandomState(5)
    Wb = rs.uniform(-0.08, 0.08, (100, 768)).astype(np.float32)
    bb = rs.uniform(-0.08, 0.08, 100).astype(np.float32)
    xb = rs.randn(6, 768).astype(np.float32)
    mb = ref_models.CALModel(2 * feat_dim + 2, pretrained_emb=None)
    mb.lang_fc.load_state_dict({"weight": torch.from_numpy(Wb), "bias": torch.from_numpy(bb)})
    with torch.no_grad():
        out["bert_out"] = mb.eval()(torch.from_numpy(xb), False, "cpu", True).numpy()
    np.savez_compressed(OUT / "g1_encoders.npz", **out)
    print("G1", {k: v.shape for k, v in out.items()})


def g2_scoring(tag, clips, nv=100, nq=50, feat_dim=4096):
    counts = synth.clip_counts(nv, clips, seed=123)
    off = synth.clip_offsets(counts)
    seg, ctx = synth.video_features(counts, feat_dim, seed=123)
    tokens = synth.query_tokens(nq, seed=123)
    own, times = synth.annotations(nq, counts, seed=123)
    sd = synth.model_weights(feat_dim, seed=123)
    m = ref_model(sd, feat_dim)
    ds, videos, annots = make_dataset(seg, ctx, off, tokens, own, times)
    nmax = int(counts.max())

    def iters():
        vi = DataLoader(ds, shuffle=False, collate_fn=ref_data.validate_collate,
                        batch_sampler=ref_data.VideoBatchSampler(videos, ds.num_segments_info))
        ls = ref_data.LanguageBatchSampler(annots, ds.num_segments_info)
        ls.moments = {n: ref_utils.generate_moments(n) for n in range(nmax + 1)}   # Q5: stock table stops at 6
        li = DataLoader(ds, shuffle=False, collate_fn=ref_data.validate_collate, batch_sampler=ls)
        return vi, li

    np.random.seed(123); random.seed(123); torch.manual_seed(123)
    vi, li = iters()
    corpus = ref_evaluate.evaluate(m, vi, li, annots, "cpu")
    prior = {}
    rs = np.random.RandomState(9)
    for n in sorted(set(counts.tolist())):
        mom = ref_utils.generate_moments(n)
        prior[n] = [mom[i] for i in rs.permutation(len(mom))]
    random.seed(123)
    vi, li = iters()
    single = ref_evaluate_single.evaluate(m, vi, li, annots, "cpu", model_types=["model", "chance", "prior"], prior=prior)

    # the reference's own expressions for embeddings, distances and moment means (evaluate.py:35,44,53-58)
    with torch.no_grad():
        vemb = {v: m(ds.make_visual_features(v, 0, ds.num_segments_info[v] - 1)) for v in videos}
        qemb = torch.cat([m(ds.lang_features[q], False, "cpu") for q in range(nq)])
    moments = {n: ref_utils.generate_moments(n) for n in range(nmax + 1)}
    top_idx, top_dist, gaps, dense = [], [], [], []
    for q in range(nq):
        distances = []
        for v in videos:
            n = vemb[v].size(0)
            dist = F.pairwise_distance(vemb[v], qemb[q:q + 1].repeat(n, 1))
            for s, e in moments[n]:
                distances.append(dist.index_select(0, torch.arange(s, e + 1)).mean().item())
        d = np.asarray(distances)
        order = np.argsort(d)
        top_idx.append(order[:100]); top_dist.append(d[order[:100]].astype(np.float32))
        gaps.append(np.diff(d[order[:101]]).astype(np.float32))
        if q < 4:
            dense.append(d.astype(np.float32))
    out = dict(counts=counts, own=own, times=np.asarray(times, np.int32),
               visual_emb=torch.cat([vemb[v] for v in videos]).numpy(), query_emb=qemb.numpy(),
               top_idx=np.asarray(top_idx, np.int64), top_dist=np.asarray(top_dist), top_gaps=np.asarray(gaps),
               dense_scores=np.asarray(dense), corpus_metrics=json.dumps(corpus),
               single_metrics=json.dumps(single),
               prior=json.dumps({str(k): v for k, v in prior.items()}))
    np.savez_compressed(OUT / f"g2_scoring_{tag}.npz", **out)
    print("G2", tag, corpus, {k: v for k, v in single.items() if k == "model"})


def g3_moments_iou():
    out = {}
    for n in list(range(7)) + [21]:
        out[f"moments_{n}"] = np.asarray(ref_utils.generate_moments(n), np.int32).reshape(-1, 2)
    val = ref_utils.read_json(REF / "didemo_download" / "data" / "val_data.json")[:300]
    times = [a["times"] for a in val]
    ious = []
    for t in times:
        ious.append([ref_utils.get_iou(t, s, e) for s, e in ref_utils.generate_moments(6)])
    lens = np.asarray([len(t) for t in times], np.int32)
    flat = np.concatenate([np.asarray(t, np.int32) for t in times])
    out.update(times_flat=flat, times_len=lens,
               iou_flat=np.concatenate([np.asarray(i, np.float64).T.reshape(-1) for i in ious]))
    np.savez_compressed(OUT / "g3_moments_iou.npz", **out)
    print("G3", len(times), "annotations")


def g4_pooling():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        d = Path(td) / "features_vgg19"
        d.mkdir()
        names = []
        for T in (150, 138, 125, 112):
            x = np.random.RandomState(1000 + T).rand(T, 4096).astype(np.float32)
            x[x < 0.3] = 0.0        # post-ReLU sparsity
            np.save(d / f"vgg19_ft_vid{T}.npy", x)
            names.append(f"vid{T}")
        for mode in ("avg", "max"):
            ds = ref_data.CustomDataset(names, {}, td, "vgg19", word_indexer=None, pooling=mode)
            for T, name in zip((150, 138, 125, 112), names):
                vf = ds.video_features[name]
                out[f"seg_{mode}_{T}"] = vf["segment_features"].astype(np.float32)
                out[f"ctx_{mode}_{T}"] = np.asarray(vf["context_features"], np.float32)
                out[f"nseg_{mode}_{T}"] = np.int32(vf["num_segments"])
    np.savez_compressed(OUT / "g4_pooling.npz", **out)
    print("G4", sorted(out)[:4], "...")


def g5_tokens():
    val = ref_utils.read_json(REF / "didemo_download" / "data" / "val_data.json")
    descs = sorted({a["description"] for a in val}, key=lambda s: (-len(s.split()), s))
    sample = descs[:40] + descs[40::27][:260]          # the longest ones (> 20 words) + a spread
    captured = []

    class Recorder:                                     # stands in for WordIndexer inside load_lang_features
        def items2tensor(self, seqs, size):
            captured.append(list(seqs[0]))
            return None

    ds = ref_data.CustomDataset.__new__(ref_data.CustomDataset)
    ds.word_indexer, ds.bert_tokenizer, ds.max_query_len, ds.lang_features = Recorder(), None, 20, {}
    ds.load_lang_features({i: dict(description=s) for i, s in enumerate(sample)})
    # a real WordIndexer on a tiny GloVe-format file (model/data.py:33-118)
    vocab = sorted({w for words in captured for w in words})[::3][:150]
    rs = np.random.RandomState(3)
    with tempfile.TemporaryDirectory() as td:
        lines = [f"{w} " + " ".join(f"{x:.5f}" for x in rs.randn(100)) for w in ["<unk>"] + vocab]
        (Path(td) / "glove.6B.100d.txt").write_text("\n".join(lines) + "\n", encoding="UTF-8")
        wi = ref_data.WordIndexer(td)
        tensors = [wi.items2tensor([w], 20)[0].tolist() for w in captured]
        emb = wi.get_embeddings().numpy()
        glove_text = "\n".join(lines) + "\n"
    json.dump(dict(descriptions=sample, words=captured, glove_text=glove_text, tensors=tensors,
                   emb_checksum=float(np.abs(emb).sum()), vocab_size=int(emb.shape[0])),
              open(OUT / "g5_tokens.json", "w"))
    print("G5", len(sample), "descriptions, vocab", emb.shape)


def g6_validate_epoch():
    """Trainer.validate_epoch (model/main.py:121-212): the third copy of the scoring loop, with `>=` IoU thresholds,
    1-based ranks and reciprocal rank.  main.py imports tensorboard (absent here) only for SummaryWriter; a recorder
    stands in for it and captures the scalars the reference logs."""
    import matplotlib
    matplotlib.use("Agg")
    captured = {}

    class Recorder:
        def __init__(self, *a, **k): pass
        def add_scalars(self, tag, scalars, global_step=None): captured[tag] = {k: float(v) for k, v in scalars.items()}
        def add_scalar(self, *a, **k): pass
        def add_figure(self, *a, **k): pass

    stub = types.ModuleType("torch.utils.tensorboard")
    stub.SummaryWriter = Recorder
    sys.modules["torch.utils.tensorboard"] = stub
    import main as ref_main  # noqa: E402  (reference)

    out = {}
    for tag, clips in (("n6", 6), ("ragged", "didemo")):
        nv, nq, feat_dim = 60, 40, 4096
        counts = synth.clip_counts(nv, clips, seed=77)
        off = synth.clip_offsets(counts)
        seg, ctx = synth.video_features(counts, feat_dim, seed=77)
        tokens = synth.query_tokens(nq, seed=77)
        own, times = synth.annotations(nq, counts, seed=77)
        sd = synth.model_weights(feat_dim, seed=77)
        m = ref_model(sd, feat_dim)
        ds, videos, annots = make_dataset(seg, ctx, off, tokens, own, times)
        for size in (25, -1):
            vi = DataLoader(ds, shuffle=False, collate_fn=ref_data.validate_collate,
                            batch_sampler=ref_data.VideoBatchSampler(videos, ds.num_segments_info))
            li = DataLoader(ds, shuffle=False, collate_fn=ref_data.validate_collate,
                            batch_sampler=ref_data.LanguageBatchSampler(annots, ds.num_segments_info))
            tr = ref_main.Trainer.__new__(ref_main.Trainer)
            tr.device, tr.bert, tr.val_writer, tr.global_step = "cpu", False, Recorder(), 0
            captured.clear()
            pr = tr.validate_epoch(m, vi, li, annots, size=size)
            out[f"{tag}_size{size}"] = dict(scalars={k: dict(v) for k, v in captured.items()},
                                            pr_curve={a: {str(k): [float(x) for x in v] for k, v in b.items()} for a, b in pr.items()})
    json.dump(out, open(OUT / "g6_validate_epoch.json", "w"))
    print("G6", {k: v["scalars"].get("MedianRank") for k, v in out.items()})


def g7_ranking_loss():
    """Trainer.ranking_loss (model/main.py:214-232): loss value and autograd gradients, with and without normalize_loss."""
    import matplotlib
    matplotlib.use("Agg")
    stub = types.ModuleType("torch.utils.tensorboard")
    stub.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", stub)
    import main as ref_main  # noqa: E402  (reference)
    out = {}
    for tag, nl in (("plain", False), ("normalized", True)):
        posit, intra, inter, lang, maskp, maskn = synth.ranking_batch(41)
        t = [torch.from_numpy(a).clone().requires_grad_(True) for a in (posit, intra, inter, lang)]
        tr = ref_main.Trainer.__new__(ref_main.Trainer)
        tr.normalize_loss, tr.b, tr.lamb = nl, 0.1, 0.4
        loss, n = tr.ranking_loss(t[0], t[1], t[2], t[3], torch.from_numpy(maskp), torch.from_numpy(maskn))
        loss.backward()
        out[f"loss_{tag}"] = np.float32(loss.item())
        out[f"n_{tag}"] = np.int64(n)
        for name, x in zip(("posit", "intra", "inter", "lang"), t):
            out[f"grad_{name}_{tag}"] = x.grad.numpy()
    np.savez_compressed(OUT / "g7_ranking_loss.npz", **out)
    print("G7", {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items() if k.startswith(("loss", "n_"))})


def g8_chance(nv=40, nq=130, feat_dim=256):
    """evaluate.evaluate(model_types=['model', 'chance']) under np.random.seed(123): the chance permutation is drawn ONCE
    per query and shared by the IoU thresholds (evaluate.py:68-72); nq = 130 also crosses the `preliminary` print."""
    out = {}
    for tag, clips in (("n6", 6), ("ragged", "didemo")):
        counts = synth.clip_counts(nv, clips, seed=88)
        off = synth.clip_offsets(counts)
        seg, ctx = synth.video_features(counts, feat_dim, seed=88)
        tokens = synth.query_tokens(nq, seed=88)
        own, times = synth.annotations(nq, counts, seed=88)
        sd = synth.model_weights(feat_dim, seed=88)
        m = ref_model(sd, feat_dim)
        ds, videos, annots = make_dataset(seg, ctx, off, tokens, own, times)
        vi = DataLoader(ds, shuffle=False, collate_fn=ref_data.validate_collate,
                        batch_sampler=ref_data.VideoBatchSampler(videos, ds.num_segments_info))
        li = DataLoader(ds, shuffle=False, collate_fn=ref_data.validate_collate,
                        batch_sampler=ref_data.LanguageBatchSampler(annots, ds.num_segments_info))
        np.random.seed(123)
        out[tag] = ref_evaluate.evaluate(m, vi, li, annots, "cpu", model_types=["model", "chance"])
    json.dump({k: {kk: {n: float(x) for n, x in vv.items()} for kk, vv in v.items()} for k, v in out.items()},
              open(OUT / "g8_chance.json", "w"), indent=1)
    print("G8", out)


def g9_vgg_full():
    """a2 at FULL size: one 224x224 frame through the VGG-19 "E" stack built from torch.nn modules (the layers
    torchvision.models.vgg19 is made of -- get_rgb_features.py:122-126 truncates its classifier to [0..4]; torchvision itself
    is absent here, so this pins the composition, not torchvision's file), seeded full-width weights, fc 4096."""
    import torch.nn as nn
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    cw, cb, fc6, fc7 = synth.vgg_weights(cfg, (224, 224), 4096, seed=5)
    frames = synth.frames_u8(1, 224, 224, seed=5)
    layers, cin, i = [], 3, 0
    for item in cfg:
        if item == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = nn.Conv2d(cin, item, kernel_size=3, padding=1)
            conv.weight.data.copy_(torch.from_numpy(cw[i])); conv.bias.data.copy_(torch.from_numpy(cb[i]))
            layers += [conv, nn.ReLU(inplace=True)]
            cin, i = item, i + 1
    features = nn.Sequential(*layers)
    avgpool = nn.AdaptiveAvgPool2d((7, 7))
    l6, l7 = nn.Linear(512 * 49, 4096), nn.Linear(4096, 4096)
    l6.weight.data.copy_(torch.from_numpy(fc6[0])); l6.bias.data.copy_(torch.from_numpy(fc6[1]))
    l7.weight.data.copy_(torch.from_numpy(fc7[0])); l7.bias.data.copy_(torch.from_numpy(fc7[1]))
    classifier = nn.Sequential(l6, nn.ReLU(True), nn.Dropout(), l7, nn.ReLU(True))          # vgg19.classifier[0..4]
    model = nn.Sequential(features, avgpool, nn.Flatten(1), classifier).eval()
    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])
    x = torch.from_numpy(frames).transpose(3, 1).transpose(2, 3).float().div(255)                 # get_rgb_features.py:64-69
    x = x.sub(mean[None, :, None, None]).div(std[None, :, None, None])
    with torch.no_grad():
        out = model(x).numpy()
    np.savez_compressed(OUT / "g9_vgg_full.npz", fc7=out.astype(np.float32))
    print("G9", out.shape, float(out.max()), float(out.mean()))


def g10_encoder_grads():
    """loss.backward() through the reference's CALModel (model/main.py:58-67 with model/models.py:54-66): gradients of every
    trainable parameter for a seeded reduced model (F = 16, hidden 24, vocab 60; dropout 0 so the fixture does not depend on
    torch's RNG), loss = sum(vis * wv) + sum(lang * wl) with fixed weights wv / wl; plain and normalize_lang."""
    out = {}
    for tag, nl in (("plain", False), ("normlang", True)):
        sd = synth.model_weights(16, vocab=60, hidden=24, seed=31, normalize_lang=nl)
        emb = torch.from_numpy(sd["word_embedding.weight"])
        m = ref_models.CALModel(2 * 16 + 2, pretrained_emb=emb, hidden_size=24, dropout_rate=0.0, normalize_lang=nl)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.train()
        rs = np.random.RandomState(32)
        x = torch.from_numpy(rs.rand(37, 34).astype(np.float32))
        tok = torch.from_numpy(synth.query_tokens(11, vocab=60, seed=33))
        wv = torch.from_numpy(rs.randn(37, 100).astype(np.float32))
        wl = torch.from_numpy(rs.randn(11, 100).astype(np.float32))
        vis = m(x)
        lang = m(tok, False, "cpu")
        loss = (vis * wv).sum() + (lang * wl).sum()
        loss.backward()
        out[f"{tag}_vis"] = vis.detach().numpy()
        out[f"{tag}_lang"] = lang.detach().numpy()
        for name, p_ in m.named_parameters():
            if p_.grad is not None:
                out[f"{tag}_grad_{name}"] = p_.grad.numpy().copy()
    np.savez_compressed(OUT / "g10_encoder_grads.npz", **out)
    print("G10", sorted(k for k in out if "grad" in k))


G11_CASES = [  # (decoded frames, fps, num_segments)
    (900, 30.0, 6), (819, 30.0, 6), (750, 30.0, 5), (150, 30.0, 1), (60, 30.0, 1), (1, 30.0, 1), (0, 30.0, 6),
    (899, 29.97, 6), (720, 24.0, 6), (719, 23.976, 6), (625, 25.0, 5), (610, 25.0, 5), (450, 15.0, 6), (375, 12.5, 6),
    (1798, 59.94, 6), (1500, 59.94, 6), (301, 30.0, 2), (299, 30.0, 2), (164, 30.0, 2), (151, 30.0, 2), (436, 29.97, 3),
    (449, 29.97, 3), (30, 30.0, 6), (866, 30.0, 6), (884, 30.0, 6), (885, 30.0, 6), (120, 24.0, 1), (132, 24.0, 1),
]


def g11_frame_front_end():
    """a1 + f4 front half: the UNMODIFIED ``DiDeMoDataset.__getitem__`` (get_rgb_features.py:37-78).  torchvision is absent
    from this image; the module only needs ``torchvision.io.read_video`` at call time, so a stub returns seeded uint8
    frames and ``{'video_fps': fps}`` (decode is codec I/O, outside the path).  Index cases: 1x1 frames whose red / green
    bytes spell the frame number, so the selected indices are read back from the reference's own normalised output.
    Normalisation case: 16x16 frames carrying every byte value in every channel; the full [T_sel, 3, 16, 16] tensor is kept."""
    table = {}

    def read_video(filename, pts_unit="sec", end_pts=None, **kw):
        frames, fps = table[Path(filename).stem]
        return torch.from_numpy(frames), torch.empty(0), {"video_fps": fps}

    tv = types.ModuleType("torchvision")
    tv.io = types.ModuleType("torchvision.io")
    tv.io.read_video = read_video
    sys.modules["torchvision"], sys.modules["torchvision.io"] = tv, tv.io
    sys.path.insert(0, str(REF))
    import get_rgb_features as ref_rgb  # noqa: E402  (reference; its __main__ block does not run on import)

    mean = np.array([0.485, 0.456, 0.406]); std = np.array([0.229, 0.224, 0.225])
    info, out = [], {"cases": np.asarray(G11_CASES, np.float64)}
    for i, (nf, fps, nseg) in enumerate(G11_CASES):
        fr = np.zeros((nf, 1, 1, 3), np.uint8)
        fr[:, 0, 0, 0] = np.arange(nf) % 256
        fr[:, 0, 0, 1] = np.arange(nf) // 256
        table[f"case{i}"] = (fr, fps)
        info.append(dict(video=f"case{i}", num_segments=nseg))
    ds = ref_rgb.DiDeMoDataset(info, dataset_dir="unused")
    for i, (nf, fps, nseg) in enumerate(G11_CASES):
        item = ds[i]
        x = item["frames"].numpy()
        if nf == 0:
            assert x.shape == (0, 1, 1, 3)                                  # the unreadable-file branch (:75-78)
            out[f"idx_{i}"] = np.zeros(0, np.int64)
            continue
        b = np.rint((x[:, :, 0, 0].astype(np.float64) * std + mean) * 255.0).astype(np.int64)
        out[f"idx_{i}"] = b[:, 0] + 256 * b[:, 1]
    # normalisation: every byte value in every channel
    fr = synth.frames_u8(150, 16, 16, seed=11)
    fr[0, :, :, 0] = np.arange(256, dtype=np.uint8).reshape(16, 16)
    fr[0, :, :, 1] = np.arange(256, dtype=np.uint8).reshape(16, 16)[::-1]
    fr[0, :, :, 2] = np.arange(256, dtype=np.uint8).reshape(16, 16).T
    table["norm"] = (fr, 30.0)
    dsn = ref_rgb.DiDeMoDataset([dict(video="norm", num_segments=1)], dataset_dir="unused")
    out["norm_frames"] = dsn[0]["frames"].numpy()
    np.savez_compressed(OUT / "g11_frame_front_end.npz", **out)
    print("G11", {i: len(out[f"idx_{i}"]) for i in range(len(G11_CASES))}, out["norm_frames"].shape)


def torch_resnet(sd, blocks, width):
    """A Bottleneck ResNet up to its global average pool built from torch.nn modules -- the layers torchvision.models.resnet152
    is made of (torchvision itself is absent here): Conv2d(bias=False) + BatchNorm2d + ReLU, MaxPool2d(3, 2, 1), Bottleneck
    blocks with the stride on the 3x3 convolution and a 1x1 downsample branch on the first block of each layer,
    AdaptiveAvgPool2d((1, 1)); get_rgb_features.py:127-131 keeps exactly children()[:-1] of that model."""
    import torch.nn as nn

    def cb(conv, bn, cin, cout, k, s, p_):
        c = nn.Conv2d(cin, cout, k, s, p_, bias=False)
        c.weight.data.copy_(torch.from_numpy(sd[conv + ".weight"]))
        b = nn.BatchNorm2d(cout)
        b.weight.data.copy_(torch.from_numpy(sd[bn + ".weight"])); b.bias.data.copy_(torch.from_numpy(sd[bn + ".bias"]))
        b.running_mean.copy_(torch.from_numpy(sd[bn + ".running_mean"])); b.running_var.copy_(torch.from_numpy(sd[bn + ".running_var"]))
        return c, b

    class Bottleneck(nn.Module):
        def __init__(self, pre, cin, mid, stride, down):
            super().__init__()
            self.conv1, self.bn1 = cb(pre + ".conv1", pre + ".bn1", cin, mid, 1, 1, 0)
            self.conv2, self.bn2 = cb(pre + ".conv2", pre + ".bn2", mid, mid, 3, stride, 1)
            self.conv3, self.bn3 = cb(pre + ".conv3", pre + ".bn3", mid, 4 * mid, 1, 1, 0)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = nn.Sequential(*cb(pre + ".downsample.0", pre + ".downsample.1", cin, 4 * mid, 1, stride, 0)) if down else None

        def forward(self, x):
            identity = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
            return self.relu(out + identity)

    layers = [*cb("conv1", "bn1", 3, width, 7, 2, 3), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)]
    cin = width
    for li, nb in enumerate(blocks):
        mid = width * 2 ** li
        for b in range(nb):
            layers.append(Bottleneck(f"layer{li + 1}.{b}", cin, mid, 2 if (b == 0 and li > 0) else 1, b == 0))
            cin = 4 * mid
    layers.append(nn.AdaptiveAvgPool2d((1, 1)))
    return nn.Sequential(*layers).eval()


def g12_resnet_full():
    """f4 at FULL size: two 224x224 frames through ResNet-152 (blocks 3/8/36/3, width 64 -> 2048-d) built from torch.nn modules,
    seeded full-width weights (synth.resnet_weights), input normalised as get_rgb_features.py:64-69 does."""
    blocks, width = (3, 8, 36, 3), 64
    sd = synth.resnet_weights(blocks, width, seed=12)
    frames = synth.frames_u8(2, 224, 224, seed=12)
    mean = torch.tensor([0.485, 0.456, 0.406]); std = torch.tensor([0.229, 0.224, 0.225])
    x = torch.from_numpy(frames).transpose(3, 1).transpose(2, 3).float().div(255)
    x = x.sub(mean[None, :, None, None]).div(std[None, :, None, None])
    with torch.no_grad():
        out = torch_resnet(sd, blocks, width)(x).flatten(1).numpy()
    np.savez_compressed(OUT / "g12_resnet_full.npz", pooled=out.astype(np.float32))
    print("G12", out.shape, float(out.max()), float(out.mean()))


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # python tools/gen_golden.py g11_frame_front_end
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g11_frame_front_end()
    g12_resnet_full()
    g3_moments_iou()
    g5_tokens()
    g4_pooling()
    g1_encoders()
    g2_scoring("n6", 6)
    g2_scoring("ragged", "didemo")
    g2_scoring("n21", 21)
    g6_validate_epoch()
    g7_ranking_loss()
    g8_chance()
    g9_vgg_full()
    g10_encoder_grads()
