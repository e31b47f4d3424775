"""Parity tests proper: every HIP kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: bit-exact (the kernels implement the oracle's canonical fp32 evaluation order), and
within 1e-4 / identical top-k against the reference's own golden vectors."""
import json
import os
import random
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import MemoryDataset, lstm_of, make_model, problem
from vfr_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vfr():
    from vfr_amd import _vfr
    _vfr.lib()
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return _vfr


@pytest.fixture(autouse=True)
def prefilter_reachable(vfr):
    """The library scores batches of up to 64 queries with the few-queries path; most scoring tests here use such batches to
    exercise the MFMA pre-filter and the fused kernels, so the limit is 8 for every test (the few-queries tests raise it)."""
    old = vfr.get_option("score_smallq")
    vfr.set_option("score_smallq", 8)
    yield
    vfr.set_option("score_smallq", old)


@pytest.fixture(params=["exact", "mfma"])
def score_mode(request, vfr):
    """Every scoring test runs twice: the exact VALU kernels, and the fp32 MFMA pre-filter + exact re-scoring path
    (vfr_score_topk_mfma, dtype f32), whose outputs must be the same bits."""
    old, old_min = vfr.DEFAULT_SCORE_MODE, vfr.get_option("score_mfma_min")
    vfr.DEFAULT_SCORE_MODE = request.param
    vfr.set_option("score_mfma_min", 0)              # the tests' small banks must reach the pre-filter kernels too
    yield request.param
    vfr.DEFAULT_SCORE_MODE = old
    vfr.set_option("score_mfma_min", old_min)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype else t).to(DEV)


def same(a, b):
    """bit-for-bit as values (-0 == +0), NaN-free"""
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    return a.shape == b.shape and np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------
def test_canonical_math_bit_exact(vfr, oracle):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.randn(200000) * 4, rs.uniform(-100, 100, 50000), [0.0, -0.0, 1e-30, 88.0, -88.0, 20.0]]).astype(np.float32)
    y = np.concatenate([rs.uniform(0.5, 30, 200000), rs.randint(1, 22, 50006)]).astype(np.float32)
    for op in (0, 1, 2, 5):
        assert same(vfr.math_f32(op, dev(x), dev(y)), oracle.math_f32(op, x, y)), f"op {op}"
    xp = np.abs(x)
    assert same(vfr.math_f32(3, dev(xp), dev(y)), oracle.math_f32(3, xp, y))       # IEEE division
    assert same(vfr.math_f32(4, dev(xp)), oracle.math_f32(4, xp))                  # IEEE sqrt


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (7, 100, 13), (130, 768, 100), (257, 1003, 70), (64, 4096, 500)])
@pytest.mark.parametrize("gemm", [0, 1])
def test_linear_bit_exact(vfr, oracle, M, K, N, gemm):
    rs = np.random.RandomState(M + K + N)
    A, W, b = rs.randn(M, K).astype(np.float32), rs.randn(N, K).astype(np.float32), rs.randn(N).astype(np.float32)
    vfr.set_option("gemm", gemm)
    try:
        for relu in (False, True):
            assert same(vfr.linear(dev(A), dev(W), dev(b), relu), oracle.linear(A, W, b, relu))
        assert same(vfr.linear(dev(A), dev(W)), oracle.linear(A, W))
    finally:
        vfr.set_option("gemm", 1)


@pytest.mark.parametrize("mode", ["avg", "max"])
def test_segment_pool_norm(vfr, oracle, golden, mode):
    g = golden("g4_pooling.npz")
    frames, counts = [], []
    for T in (150, 138, 125, 112, 1, 26):
        x = np.random.RandomState(1000 + T).rand(T, 4096).astype(np.float32)
        x[x < 0.3] = 0.0
        seg, ctx = vfr.segment_pool_norm(dev(x), 25, mode)
        oseg, octx = oracle.segment_pool_norm(x, 25, mode)
        assert same(seg, oseg) and same(ctx, octx)
        if T >= 112:
            np.testing.assert_allclose(seg.cpu().numpy(), g[f"seg_{mode}_{T}"], rtol=0, atol=3e-7)
            np.testing.assert_allclose(ctx.cpu().numpy(), g[f"ctx_{mode}_{T}"], rtol=0, atol=3e-7)
        frames.append(x); counts.append(T)
    seg, ctx, nseg = vfr.segment_pool_norm_batch(dev(np.concatenate(frames)), counts, 25, mode)
    off = np.concatenate([[0], np.cumsum(nseg.numpy())])
    for i, x in enumerate(frames):
        oseg, octx = oracle.segment_pool_norm(x, 25, mode)
        assert same(seg[off[i]:off[i + 1]], oseg) and same(ctx[i], octx)


@pytest.mark.parametrize("gemm", [0, 1])
def test_visual_mlp_bit_exact_and_golden(vfr, oracle, golden, gemm):
    g = golden("g1_encoders.npz")
    counts = g["counts"]
    seg, ctx = synth.video_features(counts, 4096, seed=11)
    sd = synth.model_weights(4096, seed=11)
    off = synth.clip_offsets(counts)
    vfr.set_option("gemm", gemm)
    try:
        got = vfr.visual_mlp(dev(seg), dev(ctx), dev(off), dev(sd["visual_fc.0.weight"]), dev(sd["visual_fc.0.bias"]),
                             dev(sd["visual_fc.2.weight"]), dev(sd["visual_fc.2.bias"]))
    finally:
        vfr.set_option("gemm", 1)
    want = oracle.visual_mlp(seg, ctx, off, sd["visual_fc.0.weight"], sd["visual_fc.0.bias"], sd["visual_fc.2.weight"],
                             sd["visual_fc.2.bias"])
    assert same(got, want)
    np.testing.assert_allclose(got.cpu().numpy(), g["visual_emb"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("normlang", [False, True])
def test_bilstm_bit_exact_and_golden(vfr, oracle, golden, normlang):
    g = golden("g1_encoders.npz")
    sd = synth.model_weights(4096, seed=11, normalize_lang=normlang)
    tokens = g["tokens"]
    lt = sd.get("learnable_length.weight")
    got = vfr.bilstm_final(dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
                           dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]), dev(lt) if lt is not None else None)
    want = oracle.bilstm_final(tokens, sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"], lt)
    assert same(got, want)
    np.testing.assert_allclose(got.cpu().numpy(), g["query_emb" + ("_normlang" if normlang else "")], rtol=0, atol=1e-4)


def test_bilstm_odd_shapes(vfr, oracle):
    sd = synth.model_weights(16, vocab=50, hidden=24, seed=5)
    tokens = synth.query_tokens(67, vocab=50, seed=5)[:, :11]
    got = vfr.bilstm_final(dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
                           dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]))
    want = oracle.bilstm_final(tokens, sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
    assert same(got, want)


# ---------------------------------------------------------------------------------------------
def _bank(vfr, V, off, id_base=0):
    return vfr.VideoBank(dev(V), dev(off.astype(np.int32)), id_base)


@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo"), ("n21", 21)])
def test_scoring_on_reference_embeddings(vfr, oracle, golden, tag, clips, score_mode):
    """a10-a12 on the reference's own embeddings: dense scores, own-video scores, fused top-k and rank counts."""
    g = golden(f"g2_scoring_{tag}.npz")
    Q, V, counts = g["query_emb"], g["visual_emb"], g["counts"]
    off = synth.clip_offsets(counts)
    bank = _bank(vfr, V, off)
    dense = vfr.score_moments(dev(Q), bank)
    odense = oracle.score_moments(Q, V, off)
    assert same(dense, odense)
    np.testing.assert_allclose(dense.cpu().numpy()[:g["dense_scores"].shape[0]], g["dense_scores"], rtol=0, atol=1e-4)
    own = g["own"].astype(np.int32)
    nmax = int(counts.max())
    assert same(vfr.score_own(dev(Q), bank, dev(own)), oracle.score_own(Q, V, off, own, nmax * (nmax + 1) // 2))
    for k in (1, 10, 100, 200):
        od, oi, _ = vfr.score_topk(dev(Q), bank, k)
        wd, wi = oracle.score_topk(Q, V, off, k)
        assert same(oi, wi), f"top-{k} moment indices differ"
        assert same(od, wd)
    # vs the reference's own argsort: top-1 and top-10 identical for every query
    od, oi, _ = vfr.score_topk(dev(Q), bank, 100)
    assert np.array_equal(oi.cpu().numpy()[:, :10], g["top_idx"][:, :10])
    np.testing.assert_allclose(od.cpu().numpy(), g["top_dist"], rtol=0, atol=1e-4)
    # rank counting for 3 arbitrary keys per query
    rs = np.random.RandomState(1)
    pick = rs.randint(0, odense.shape[1], size=(3, Q.shape[0]))
    rd = np.take_along_axis(odense, pick.T, axis=1).T.copy()
    _, _, cnt = vfr.score_topk(dev(Q), bank, 0, dev(rd), dev(pick.astype(np.int64)))
    for r in range(3):
        assert cnt[r].cpu().numpy().tolist() == oracle.rank_of(Q, V, off, rd[r], pick[r]).tolist()


@pytest.mark.parametrize("nq,nv,clips,k,D", [(1, 1, 6, 100, 100), (65, 3, 5, 7, 100), (130, 40, 21, 128, 100),
                                             (33, 17, "didemo", 100, 64), (5, 9, 64, 448, 100), (200, 600, 6, 100, 100)])
def test_fused_topk_edge_shapes(vfr, oracle, nq, nv, clips, k, D, score_mode):
    rs = np.random.RandomState(nq * 1000 + nv)
    counts = synth.clip_counts(nv, clips, seed=nq)
    off = synth.clip_offsets(counts)
    V = rs.randn(int(off[-1]), D).astype(np.float32)
    Q = rs.randn(nq, D).astype(np.float32)
    od, oi, _ = vfr.score_topk(dev(Q), _bank(vfr, V, off), k)
    wd, wi = oracle.score_topk(Q, V, off, k)
    assert same(oi, wi) and same(od, wd)


def test_fused_topk_exact_ties_break_by_moment_id(vfr, oracle, score_mode):
    """Duplicate videos -> many exactly equal scores; order must be (score, id)."""
    rs = np.random.RandomState(3)
    one = rs.randn(6, 100).astype(np.float32)
    V = np.tile(one, (50, 1))
    off = synth.clip_offsets(np.full(50, 6))
    Q = rs.randn(9, 100).astype(np.float32)
    od, oi, _ = vfr.score_topk(dev(Q), _bank(vfr, V, off), 100)
    wd, wi = oracle.score_topk(Q, V, off, 100)
    assert same(oi, wi) and same(od, wd)
    assert (np.diff(oi.cpu().numpy()[:, :50], axis=1) == 21).all()      # 50 copies of the best moment, ids 21 apart


def test_topk_merge_and_shard_equivalence(vfr, oracle, score_mode):
    """Per-shard top-k with id_base, merged, equals the unsharded top-k; counts add up (8e semantics)."""
    rs = np.random.RandomState(9)
    counts = synth.clip_counts(90, "didemo", seed=9)
    off = synth.clip_offsets(counts)
    V = rs.randn(int(off[-1]), 100).astype(np.float32)
    Q = rs.randn(70, 100).astype(np.float32)
    mom = np.concatenate([[0], np.cumsum(counts * (counts + 1) // 2)])
    wd, wi = oracle.score_topk(Q, V, off, 100)
    full = oracle.score_moments(Q, V, off)
    pick = rs.randint(0, full.shape[1], size=70)
    rd = full[np.arange(70), pick].copy()
    parts_d, parts_i, cnt = [], [], None
    for lo, hi in ((0, 31), (31, 64), (64, 90)):
        sub = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        bank = _bank(vfr, V[off[lo]:off[hi]], sub, id_base=int(mom[lo]))
        d, i, cnt = vfr.score_topk(dev(Q), bank, 100, dev(rd), dev(pick.astype(np.int64)), count_lt=cnt)
        parts_d.append(d); parts_i.append(i)
    md, mi = vfr.topk_merge(torch.stack(parts_d), torch.stack(parts_i))
    assert same(mi, wi) and same(md, wd)
    assert cnt[0].cpu().numpy().tolist() == oracle.rank_of(Q, V, off, rd, pick).tolist()


def test_seeded_sharded_search_equals_unsharded(vfr, oracle, score_mode):
    """The multi-GPU flow of engine.sharded_search replayed on one device: per-shard sample lists -> merged global
    sample -> its k-th key as thr_seed for the main passes -> merge(main parts + sample) == unsharded top-k."""
    from vfr_amd import engine
    rs = np.random.RandomState(21)
    counts = synth.clip_counts(3000, 21, seed=21)
    off = synth.clip_offsets(counts)
    V = rs.randn(int(off[-1]), 100).astype(np.float32) * 0.1
    Q = rs.randn(96, 100).astype(np.float32) * 0.1
    k = 100
    bank = _bank(vfr, V, off)
    wd, wi, _ = vfr.score_topk(dev(Q), bank, k)                    # unsharded (internal pre-pass path: Nv >= 2048)
    od, oi = oracle.score_topk(Q, V, off, k)
    assert same(wi, oi) and same(wd, od)
    shards = [(0, 1400), (1400, 3000)]
    samples, mains = [], []
    for lo, hi in shards:
        sb = vfr.slice_bank(bank, counts, lo, lo + 128)
        samples.append(vfr.score_topk(dev(Q), sb, k)[:2])
    sd, si = vfr.topk_merge(torch.stack([s[0] for s in samples]), torch.stack([s[1] for s in samples]))
    seed = engine._pack_key(sd[:, k - 1].contiguous(), si[:, k - 1])
    for lo, hi in shards:
        mb = vfr.slice_bank(bank, counts, lo + 128, hi)
        mains.append(vfr.score_topk(dev(Q), mb, k, thr_seed=seed)[:2])
    md, mi = vfr.topk_merge(torch.stack([m[0] for m in mains] + [sd]), torch.stack([m[1] for m in mains] + [si]))
    assert same(mi, oi) and same(md, od)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo"), ("n21", 21)])
def test_evaluators_end_to_end_on_gpu(vfr, oracle, golden, tag, clips, score_mode):
    """Drop-in surface on the device: same iterators, same dicts as the reference; embeddings == oracle bits."""
    from vfr_amd import evaluate as vevaluate
    from vfr_amd import evaluate_single as vsingle
    g = golden(f"g2_scoring_{tag}.npz")
    p = problem(100, 50, clips)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    model = make_model(p["sd"]).to(DEV)
    vi, li = ds.iterators()
    got, (td, ti) = vevaluate.evaluate(model, vi, li, ds.annotations, DEV, return_topk=100)
    ref = json.loads(str(g["corpus_metrics"]))
    for key in ref:
        assert got[key] == pytest.approx(ref[key], abs=1e-9), key
    sd = p["sd"]
    vis = oracle.visual_mlp(p["seg"], p["ctx"], p["off"], sd["visual_fc.0.weight"], sd["visual_fc.0.bias"],
                            sd["visual_fc.2.weight"], sd["visual_fc.2.bias"])
    qemb = oracle.bilstm_final(p["tokens"], sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
    wd, wi = oracle.score_topk(qemb, vis, p["off"], 100)
    assert same(ti, wi) and same(td, wd)                           # bit-exact moment indices, end to end
    assert np.array_equal(ti.cpu().numpy()[:, :10], g["top_idx"][:, :10])
    prior = {int(k): [tuple(m) for m in v] for k, v in json.loads(str(g["prior"])).items()}
    random.seed(123)
    vi, li = ds.iterators()
    got = vsingle.evaluate(model, vi, li, ds.annotations, DEV, model_types=["model", "chance", "prior"], prior=prior)
    ref = json.loads(str(g["single_metrics"]))
    for key in ref:
        assert got[key] == pytest.approx(ref[key], abs=1e-9), key


def test_cal_model_forward_dispatch_on_gpu(vfr, golden):
    g = golden("g1_encoders.npz")
    sd = synth.model_weights(4096, seed=11)
    model = make_model(sd).to(DEV)
    seg, ctx = synth.video_features(g["counts"], 4096, seed=11)
    ds = MemoryDataset(seg, ctx, g["counts"], g["tokens"], np.zeros(16, int), [[[0, 0]] * 4] * 16)
    x = torch.cat([ds.make_visual_features(v, 0, ds.num_segments_info[v] - 1) for v in ds.videos]).to(DEV)
    with torch.no_grad():
        vis = model(x)                                             # generic [rows, 2F+2] path through vfr_linear_f32
        q = model(dev(g["tokens"]), False, DEV)
    np.testing.assert_allclose(vis.cpu().numpy(), g["visual_emb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(q.cpu().numpy(), g["query_emb"], rtol=0, atol=1e-4)
    rs = np.random.RandomState(5)
    from vfr_amd import models as vmodels
    mb = vmodels.CALModel(8194, pretrained_emb=None).to(DEV).eval()
    W = rs.uniform(-0.08, 0.08, (100, 768)).astype(np.float32); b = rs.uniform(-0.08, 0.08, 100).astype(np.float32)
    mb.lang_fc.load_state_dict({"weight": torch.from_numpy(W), "bias": torch.from_numpy(b)})
    with torch.no_grad():
        out = mb(dev(rs.randn(6, 768).astype(np.float32)), False, DEV, True)
    np.testing.assert_allclose(out.cpu().numpy(), g["bert_out"], rtol=0, atol=1e-4)


# ---------------------------------------------------------------------------------------------
SMALL_VGG = [8, 8, "M", 16, 16, "M", 24, 24, 24, 24, "M", 32, 32, 32, 32, "M", 32, 32, 32, 32, "M"]


@pytest.mark.parametrize("hw", [(32, 32), (64, 48)])
def test_vgg_stack_bit_exact_reduced_width(vfr, oracle, hw):
    H, W = hw
    frames = synth.frames_u8(5, H, W, seed=2)
    cw, cb, fc6, fc7 = synth.vgg_weights(SMALL_VGG, hw, 64, seed=2)
    assert same(vfr.frames_normalize(dev(frames)), oracle.frames_normalize(frames))
    x = oracle.frames_normalize(frames)
    assert same(vfr.conv3x3_relu(dev(x), dev(cw[0]), dev(cb[0])), oracle.conv3x3_relu(x, cw[0], cb[0]))
    y = oracle.conv3x3_relu(x, cw[0], cb[0])
    assert same(vfr.maxpool2(dev(y)), oracle.maxpool2(y))
    assert same(vfr.adaptive_avgpool7(dev(y)), oracle.adaptive_avgpool7(y))
    got = vfr.vgg_fc7(dev(frames), SMALL_VGG, [dev(w) for w in cw], [dev(b) for b in cb],
                      (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    want = oracle.vgg_fc7(frames, cw, cb, fc6, fc7, SMALL_VGG)
    assert same(got, want)


# layer shapes that reach every conv launch form of gemm_nt: 128x64 tiles (Cout <= 64), 64-row tiles (< 384 workgroups),
# the XCD-aware tile order (several column tiles, >= 64 row tiles) and the plain 128x128 grid
WIDE_VGG = [8, 520, "M", 136, 72, "M", 8, "M", 8, "M", 8, "M"]
TALL_VGG = [72, "M", 8, "M", 8, "M", 8, "M", 8, "M"]


@pytest.mark.gpu
def test_frames_normalize_equals_reference_getitem(vfr, golden):
    """a1 against the reference itself: vfr_frames_normalize_f32 on the frames the product's selection keeps == the tensor the
    unmodified DiDeMoDataset.__getitem__ returned (fixture G11, get_rgb_features.py:37-78), bit for bit."""
    from test_host_logic import g11_norm_input
    from vfr_amd import features
    g = golden("g11_frame_front_end.npz")
    fr = g11_norm_input()
    idx = features.sample_frames(150, 30.0, 1)
    assert idx.tolist() == g["idx_3"].tolist()
    assert same(vfr.frames_normalize(dev(fr[idx])), g["norm_frames"])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,T", [(WIDE_VGG, 5), (TALL_VGG, 17)])
def test_vgg_conv_launch_forms_bit_exact(vfr, oracle, cfg, T):
    frames = synth.frames_u8(T, 64, 48, seed=4)
    cw, cb, fc6, fc7 = synth.vgg_weights(cfg, (64, 48), 64, seed=4)
    got = vfr.vgg_fc7(dev(frames), cfg, [dev(w) for w in cw], [dev(b) for b in cb],
                      (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    assert same(got, oracle.vgg_fc7(frames, cw, cb, fc6, fc7, cfg))
    # the max-pools above ran inside the preceding convolution's epilogue (EPI_POOL2: GEMM rows in pooling-window order);
    # the same stack with the pools as their own kernels
    try:
        vfr.set_option("vgg_fuse_pool", 0)
        vfr.set_option("vgg_direct1", 0)          # and the first convolution through the implicit-GEMM kernel instead of the direct one
        unfused = vfr.vgg_fc7(dev(frames), cfg, [dev(w) for w in cw], [dev(b) for b in cb],
                              (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    finally:
        vfr.set_option("vgg_fuse_pool", 1)
        vfr.set_option("vgg_direct1", 1)
    assert torch.equal(got.view(torch.int32), unfused.view(torch.int32))


@pytest.mark.parametrize("hw,T", [((32, 32), 5), ((48, 40), 3)])
def test_vgg_halo_padded_stack_bit_exact(vfr, oracle, hw, T):
    """Widths that are multiples of 32 and even sizes at every pool: the stack runs on halo-padded activations (`vgg_halo`:
    zero border, convolution loader without tap masks or selects, pooled outputs written into the next padded tensor) -- ==
    the oracle and == the unpadded stack (option off), bit for bit; 128 x 64 and 64-row launch forms, a tile that spans images."""
    cfg = [32, 32, "M", 64, 64, "M", 160, 32, "M"]
    frames = synth.frames_u8(T, hw[0], hw[1], seed=12)
    cw, cb, fc6, fc7 = synth.vgg_weights(cfg, hw, 64, seed=12)
    args = (dev(frames), cfg, [dev(w) for w in cw], [dev(b) for b in cb], (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    got = vfr.vgg_fc7(*args)
    assert same(got, oracle.vgg_fc7(frames, cw, cb, fc6, fc7, cfg))
    try:
        vfr.set_option("vgg_halo", 0)
        plain = vfr.vgg_fc7(*args)
    finally:
        vfr.set_option("vgg_halo", 1)
    assert torch.equal(got.view(torch.int32), plain.view(torch.int32))


def test_vgg_first_conv_direct_kernel_ragged(vfr, oracle):
    """The direct first-convolution kernel's 16-channel rounds (LDS-transposed stores) on a pixel count that is no multiple of
    a wave, a pool on an odd height behind it (not fusable) and fused ones elsewhere: == oracle."""
    cfg = [48, 8, "M", 8, "M", 8, "M"]
    frames = synth.frames_u8(3, 36, 28, seed=9)
    cw, cb, fc6, fc7 = synth.vgg_weights(cfg, (36, 28), 64, seed=9)
    got = vfr.vgg_fc7(dev(frames), cfg, [dev(w) for w in cw], [dev(b) for b in cb],
                      (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    assert same(got, oracle.vgg_fc7(frames, cw, cb, fc6, fc7, cfg))


# ---------------------------------------------------------------------------------------------
def test_full_size_properties(vfr, oracle, score_mode):
    """BASELINE config 1 shape (10k videos x 21 clips): size-independent properties -- shard-and-merge == unsharded,
    top-k sorted with unique ids, rank count consistent with the list, sampled rows == dense kernel -- and, for four
    queries of a 5000-query batch, the fused pass over the FULL corpus against the CPU oracle (top-100 ids, distances and
    both rank counts, bit for bit)."""
    torch.manual_seed(0)
    nv, n, nq, k = 10000, 21, 256, 100
    V = torch.randn(nv * n, 100, device=DEV)
    Q = torch.randn(nq, 100, device=DEV)
    off = torch.arange(0, nv * n + 1, n, dtype=torch.int32, device=DEV)
    bank = vfr.VideoBank(V, off)
    d, i, _ = vfr.score_topk(Q, bank, k)
    assert bool((d[:, 1:] >= d[:, :-1]).all())
    assert all(len(set(r)) == k for r in i.cpu().numpy().tolist())
    half = nv // 2
    M = n * (n + 1) // 2
    b0 = vfr.VideoBank(V[:half * n].contiguous(), off[:half + 1].contiguous(), 0)
    b1 = vfr.VideoBank(V[half * n:].contiguous(), (off[half:] - off[half]).contiguous(), half * M)
    d0, i0, _ = vfr.score_topk(Q, b0, k)
    d1, i1, _ = vfr.score_topk(Q, b1, k)
    md, mi = vfr.topk_merge(torch.stack([d0, d1]), torch.stack([i0, i1]))
    assert torch.equal(mi, i) and torch.equal(md, d)
    # the 37th best moment has exactly 36 moments before it
    _, _, cnt = vfr.score_topk(Q, bank, 0, d[:, 36].contiguous(), i[:, 36].contiguous())
    assert bool((cnt[0] == 36).all())
    # the whole fused pass (threshold ladder, flagged levels, merges, two rank keys) against the dense kernel + a sort, for a
    # few queries over the FULL corpus, inside a 5000-query batch like the bench's
    Qb = torch.randn(5000, 100, device=DEV)
    Qb[:nq] = Q
    sel = [0, 63, 64, 255]
    dense = vfr.score_moments(Qb[sel].contiguous(), bank)                         # [4, 2.31M]
    order = torch.argsort(dense, dim=1, stable=True)[:, :k]                     # stable: ties by moment id
    rk_d = torch.stack([dense.gather(1, order[:, 36:37]).squeeze(1), dense.gather(1, order[:, 90:91]).squeeze(1)])
    rk_i = torch.stack([order[:, 36], order[:, 90]])
    full_rd = torch.full((2, 5000), 1e30, device=DEV); full_ri = torch.zeros((2, 5000), dtype=torch.int64, device=DEV)
    full_rd[:, sel] = rk_d; full_ri[:, sel] = rk_i
    db, ib, cb = vfr.score_topk(Qb, bank, k, full_rd.contiguous(), full_ri.contiguous())
    assert torch.equal(ib[sel], order) and torch.equal(db[sel], dense.gather(1, order))
    assert cb[0, sel].tolist() == [36] * 4 and cb[1, sel].tolist() == [90] * 4
    # the same four queries through the CPU oracle over all 10 000 videos (model/evaluate.py:49-80 restated)
    Vh, offh, Qh = V.cpu().numpy(), off.cpu().numpy(), Qb[sel].cpu().numpy()
    od, oi = oracle.score_topk(Qh, Vh, offh, k)
    assert np.array_equal(ib[sel].cpu().numpy(), oi) and np.array_equal(db[sel].cpu().numpy(), od)
    for t, pos in enumerate((36, 90)):
        oc = oracle.rank_of(Qh, Vh, offh, od[:, pos].copy(), oi[:, pos].copy())
        assert oc.tolist() == [pos] * 4 and cb[t, sel].tolist() == oc.tolist()
    assert torch.equal(ib[:nq], i) and torch.equal(db[:nq], d)                  # a query's result does not depend on its batch
    # spot-check against the dense kernel on the videos that own the winners
    top_vid = (i[:8, 0] // M).cpu().numpy()
    for q, v in enumerate(top_vid):
        sub = vfr.VideoBank(V[v * n:(v + 1) * n].contiguous(), off[:2].contiguous(), int(v) * M)
        dense = vfr.score_moments(Q[q:q + 1].contiguous(), sub)
        assert float(dense.min()) == float(d[q, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo")])
@pytest.mark.parametrize("size", [25, -1])
def test_validate_epoch_on_gpu_matches_reference_scalars(vfr, tag, clips, size):
    """Trainer.validate_epoch (main.py:121-212) through the fused kernels == the scalars the reference logged."""
    from vfr_amd import evaluate as vevaluate
    from test_host_logic import _check_validate
    ref = json.load(open(Path(__file__).parent / "golden" / "g6_validate_epoch.json"))[f"{tag}_size{size}"]
    p = problem(60, 40, clips, seed=77)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    vi, li = ds.iterators()
    got = vevaluate.validate_epoch(make_model(p["sd"]).to(DEV), vi, li, ds.annotations, DEV, size=size)
    _check_validate(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["avg", "max"])
def test_file_dataset_pools_on_gpu_and_evaluates(vfr, oracle, golden, mode, tmp_path):
    """data.CustomDataset over .npy frame features (model/data.py:163-188) with the pooling done by the per-video HIP
    kernel: pooled features == oracle bits and == the reference's arrays (g4); then the full evaluate() from files."""
    from vfr_amd import data as vdata
    from vfr_amd import evaluate as vevaluate
    g = golden("g4_pooling.npz")
    d = tmp_path / "features_vgg19"
    d.mkdir()
    lengths, names = (150, 138, 125, 112), []
    for T in lengths:
        x = np.random.RandomState(1000 + T).rand(T, 4096).astype(np.float32)
        x[x < 0.3] = 0.0
        np.save(d / f"vgg19_ft_vid{T}.npy", x)
        names.append(f"vid{T}")
    sd = synth.model_weights(4096, seed=5)
    words = [f"w{i}" for i in range(1, 60)]
    with open(tmp_path / "glove.6B.100d.txt", "w") as fh:                    # a tiny GloVe file: 59 words + unk
        for w in words + ["unk"]:
            fh.write(w + " " + " ".join(["0.5"] * 100) + "\n")
    wi = vdata.WordIndexer(str(tmp_path))
    rs = np.random.RandomState(3)
    annots = {q: dict(video=names[q % 4], description=" ".join(rs.choice(words, size=rs.randint(1, 25))),
                      times=[[0, 0], [0, 1], [0, 0], [1, 1]]) for q in range(12)}
    ds = vdata.CustomDataset(names, annots, str(tmp_path), "vgg19", word_indexer=wi, validate=True, pooling=mode,
                             pool_device="cuda")
    for T, name in zip(lengths, names):
        x = np.load(d / f"vgg19_ft_{name}.npy")
        oseg, octx = oracle.segment_pool_norm(x, 25, mode)
        vf = ds.video_features[name]
        assert vf["num_segments"] == int(g[f"nseg_{mode}_{T}"])
        assert np.array_equal(vf["segment_features"].astype(np.float32), oseg) and np.array_equal(vf["context_features"], octx)
        np.testing.assert_allclose(vf["segment_features"], g[f"seg_{mode}_{T}"], rtol=0, atol=3e-7)
        np.testing.assert_allclose(vf["context_features"], g[f"ctx_{mode}_{T}"], rtol=0, atol=3e-7)

    def iters():
        vi = torch.utils.data.DataLoader(ds, shuffle=False, collate_fn=vdata.validate_collate,
                                         batch_sampler=vdata.VideoBatchSampler(names, ds.num_segments_info))
        li = torch.utils.data.DataLoader(ds, shuffle=False, collate_fn=vdata.validate_collate,
                                         batch_sampler=vdata.LanguageBatchSampler(annots, ds.num_segments_info))
        return vi, li
    model = make_model(sd)
    cpu = vevaluate.evaluate(model, *iters(), annots, "cpu")
    gpu = vevaluate.evaluate(model.to(DEV), *iters(), annots, DEV)
    assert gpu == cpu


@pytest.mark.gpu
@pytest.mark.parametrize("nq", [1, 2, 3, 5, 8, 13, 32, 57])
def test_few_queries_scoring_path(vfr, oracle, nq):
    """1-64 queries (a serving request) are scored with lanes = clips / videos; the top-k comes from the videos whose smallest
    clip distance can still reach it (`score_smallq_select`; off: a selection tree over the key array)
    (`score_smallq`): top-k lists and rank counts bit-identical to the fused kernels (option off) and to dense + stable sort;
    ragged clip counts, an empty video, duplicated videos (exact ties), k = 0 / 1 / 100 / more than there are moments, 1-4 rank keys.
    ORACLE leg, at the library's default dispatch (`score_smallq` 64, video selection on): top-k ids / distances ==
    `oracle.score_topk`, rank counts == `oracle.rank_of`, bit for bit (model/evaluate.py:53-58,71,77)."""
    vfr.set_option("score_smallq", 64)             # (the library's default; the module's fixture lowers it for the other tests)
    rs = np.random.RandomState(40 + nq)
    for case in range(3):
        nv = [1, 37, 700][case]
        counts = rs.randint(0 if case else 1, 22, nv)
        counts[rs.randint(nv)] = 21
        off = synth.clip_offsets(counts)
        V = rs.randn(int(off[-1]), 100).astype(np.float32) * 0.2
        if nv > 2 and counts[0] and counts[0] == counts[nv - 1]:
            V[off[nv - 1]:off[nv]] = V[off[0]:off[1]]
        bank = vfr.VideoBank(dev(V), dev(off.astype(np.int32)))
        Q = dev(rs.randn(nq, 100).astype(np.float32) * 0.2)
        dense = vfr.score_moments(Q, bank)
        total = dense.shape[1]
        order = torch.argsort(dense, dim=1, stable=True)
        R = [2, 1, 4][case]
        pos = [int(rs.randint(0, total)) for _ in range(R)]
        rd = torch.stack([dense.gather(1, order[:, p:p + 1]).squeeze(1) for p in pos]).contiguous()
        ri = torch.stack([order[:, p] for p in pos]).contiguous()
        for k in (0, 1, 100, total + 7 if total + 7 <= 448 else 300):
            d, i, c = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
            try:
                vfr.set_option("score_smallq", 0)
                d2, i2, c2 = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
                vfr.set_option("score_smallq", 64)
                vfr.set_option("score_smallq_select", 0)       # the key array + selection tree instead of the video selection
                d3, i3, c3 = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
                vfr.set_option("score_smallq_select", 1)
                # rank counts with lane = video (`score_smallq_rank`: from 8 queries on by default) forced on at any query count, and off
                vfr.set_option("score_smallq_rank", 1)
                d4, i4, c4 = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
                vfr.set_option("score_smallq_rank", 0)
                d5, i5, c5 = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
            finally:
                vfr.set_option("score_smallq", 64)
                vfr.set_option("score_smallq_select", 1)
                vfr.set_option("score_smallq_rank", 8)
            assert c.tolist() == [[p] * nq for p in pos] and torch.equal(c, c2) and torch.equal(c, c3)
            assert torch.equal(c, c4) and torch.equal(c, c5)
            if k:
                assert torch.equal(i, i4) and torch.equal(d, d4) and torch.equal(i, i5) and torch.equal(d, d5)
            if k:
                # the CPU oracle on the same inputs (the production dispatch for 1-64 queries answered `d, i, c`)
                assert vfr.get_option("score_smallq") == 64 and vfr.get_option("score_smallq_select") == 1
                Qn, off32 = Q.cpu().numpy(), off.astype(np.int32)
                od, oi = oracle.score_topk(Qn, V, off32, k)
                assert same(i, oi) and same(d, od), f"few-queries top-{k} differs from the oracle (nq {nq}, case {case})"
                for r in range(R):
                    oc = oracle.rank_of(Qn, V, off32, rd[r].cpu().numpy().copy(), ri[r].cpu().numpy().copy())
                    assert same(c[r], oc), f"few-queries rank counts differ from the oracle (nq {nq}, case {case}, key {r})"
            if k:
                assert torch.equal(i, i3) and torch.equal(d, d3)
            if k:
                kk = min(k, total)
                assert torch.equal(i[:, :kk], order[:, :kk]) and torch.equal(d[:, :kk], dense.gather(1, order[:, :kk]))
                assert bool((i[:, kk:] == -1).all()) and bool(torch.isinf(d[:, kk:]).all())
                assert torch.equal(i, i2) and torch.equal(d, d2)


@pytest.mark.gpu
@pytest.mark.parametrize("max_n", [6, 21])
def test_mfma_whole_video_early_out(vfr, oracle, max_n):
    """The MFMA pre-filter skips the rank half of the moment triangle for a video when the video's smallest / largest clip
    distance already decides every moment against both rank keys for (nearly) all 64 lanes (`score_defer`: at most that many
    undecided lanes, which are re-counted exactly; -1 = early-out off).  Rank keys in the near tail (a trained model's regime),
    mid-distribution, and at the far end (every video BELOW the keys): rank counts and top-k identical for every setting,
    == the exact kernels, and == `oracle.rank_of` (model/evaluate.py:67-77) on the first queries; bf16 mode: identical across
    settings (approximate by design, but the early-out must not change what it counts)."""
    rs = np.random.RandomState(300 + max_n)
    nv, nq = 500, 1100
    counts = rs.randint(max_n - 1 if max_n == 6 else 3, max_n + 1, nv)
    counts[rs.randint(nv)] = max_n
    off = synth.clip_offsets(counts)
    V = (rs.randn(int(off[-1]), 100) * 0.2).astype(np.float32)
    src = rs.randint(0, int(off[-1]), nq)
    Qn = (V[src] + rs.randn(nq, 100).astype(np.float32) * 0.05).astype(np.float32)      # every query next to some clip
    bank = vfr.VideoBank(dev(V), dev(off.astype(np.int32)))
    Q = dev(Qn)
    dense = vfr.score_moments(Q, bank)
    total = dense.shape[1]
    order = torch.argsort(dense, dim=1, stable=True)
    old = vfr.get_option("score_defer")
    vfr.set_option("score_sort", 2)                # (2: sort whatever the bank size -- the default waits for 4096 videos x 1024 queries)
    try:
        for pos in ((3, 40), (total // 2, 11), (total - 1, total - 7)):
            rd = torch.stack([dense.gather(1, order[:, p:p + 1]).squeeze(1) for p in pos]).contiguous()
            ri = torch.stack([order[:, p] for p in pos]).contiguous()
            outs = {}
            for defer in (-1, 0, 8, 64):
                vfr.set_option("score_defer", defer)
                for k in (0, 10):
                    outs[(defer, k)] = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
                outs[(defer, "bf16")] = vfr.score_topk(Q, bank, 10, rd, ri, mode="bf16")
            dx, ix, cx = vfr.score_topk(Q, bank, 10, rd, ri, mode="exact")
            try:                                               # the pass on the caller's query order instead of the difficulty-sorted one
                vfr.set_option("score_sort", 0)
                vfr.set_option("score_defer", 8)
                du, iu, cu = vfr.score_topk(Q, bank, 10, rd, ri, mode="mfma")
            finally:
                vfr.set_option("score_sort", 2)
            assert torch.equal(cu, cx) and torch.equal(iu, ix) and torch.equal(du, dx), (pos, "unsorted")
            try:                                               # stage B's threshold throughout instead of the candidate histogram's
                vfr.set_option("score_hist", 0)
                for kk in (10, 100):
                    dh, ih, ch = vfr.score_topk(Q, bank, kk, rd, ri, mode="mfma")
                    vfr.set_option("score_hist", 1)
                    d1, i1, c1 = vfr.score_topk(Q, bank, kk, rd, ri, mode="mfma")
                    vfr.set_option("score_hist", 0)
                    assert torch.equal(ch, c1) and torch.equal(ih, i1) and torch.equal(dh, d1), (pos, kk, "hist")
                    assert torch.equal(i1[:, :10], ix) and torch.equal(d1[:, :10], dx), (pos, kk, "hist vs exact")
            finally:
                vfr.set_option("score_hist", 1)
            for defer in (-1, 0, 8, 64):
                for k in (0, 10):
                    d, i, c = outs[(defer, k)]
                    assert c.tolist() == [[p] * nq for p in pos], (pos, defer, k)
                    if k:
                        assert torch.equal(i, ix) and torch.equal(d, dx), (pos, defer, k)
                db, ib, cb = outs[(defer, "bf16")]
                d0, i0, c0 = outs[(-1, "bf16")]
                assert torch.equal(cb, c0) and torch.equal(ib, i0) and torch.equal(db, d0), (pos, defer, "bf16")
            m = 12
            for r in range(2):
                oc = oracle.rank_of(Qn[:m], V, off.astype(np.int32), rd[r, :m].cpu().numpy().copy(), ri[r, :m].cpu().numpy().copy())
                assert same(cx[r, :m], oc) and same(outs[(8, 0)][2][r, :m], oc), (pos, r)
    finally:
        vfr.set_option("score_defer", old)
        vfr.set_option("score_sort", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("nq", [1, 5, 40])
def test_graphed_request_replays_the_eager_pass(vfr, nq):
    """engine.GraphedRequest: the serving pass (query encoder + a11 labels + best-GT keys + fused top-k / rank counts) captured
    once into a HIP graph and replayed for new requests loaded into its static buffers: rank counts, top-k ids and distances
    == the eager `corpus_ranks` pass on the same request, bit for bit (the same C-ABI calls, replayed); a request without a
    ground-truth-positive moment raises the reference's IndexError from `check()` (model/evaluate.py:77)."""
    from vfr_amd import engine, models
    F = 64
    counts = synth.clip_counts(300, "didemo", seed=5)
    counts[7] = 21
    off = synth.clip_offsets(counts)
    mom = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * (counts + 1) // 2)])
    seg, ctx = synth.video_features(counts, F, seed=5)
    sd = synth.model_weights(F, vocab=90, seed=5)
    model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(DEV).eval()
    ops = engine.HipOps()
    with torch.no_grad():
        emb = model.encode_clips(dev(seg), dev(ctx), dev(off.astype(np.int32)))
    bank = vfr.VideoBank(emb, dev(off.astype(np.int32)), 0, max_clips=int(counts.max()), total_moments=int(mom[-1]), min_clips=int(counts.min()))
    shard = engine.CorpusShard(bank, 0, len(counts), counts, mom, torch.device(DEV))
    k = 20
    gr = engine.GraphedRequest(model, shard, nq, k, ops)
    for seed in (11, 12, 13):
        tokens = synth.query_tokens(nq, vocab=90, seed=seed)
        own, times = synth.annotations(nq, counts, seed=seed)
        with torch.no_grad():
            gr.load(tokens, times, own)
            counts_g, d_g, i_g = gr.replay()
            torch.cuda.synchronize()
            gr.check()
            got = (counts_g.clone(), d_g.clone(), i_g.clone())
            Q = engine.encode_queries(model, dev(tokens), torch.device(DEV), ops)
            labels = engine.gt_labels(times, counts[own], [0.5, 0.7], True, torch.device(DEV), ops)
            want = engine.corpus_ranks(shard, Q, own, labels, ops, k=k)
        assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]) and torch.equal(got[1], want[1]), (nq, seed)
    # a query whose annotators agree on nothing: no positive moment at IoU 0.7 -> the flag, read after the replay
    tokens = synth.query_tokens(nq, vocab=90, seed=14)
    own, times = synth.annotations(nq, counts, seed=14)
    n0 = int(counts[own[0]])
    times[0] = [[0, 0], [n0 - 1, n0 - 1], [0, n0 - 1], [n0 // 2, n0 // 2]]
    with torch.no_grad():
        gr.load(tokens, times, own)
        gr.replay()
        torch.cuda.synchronize()
    with pytest.raises(IndexError):
        gr.check()


@pytest.mark.gpu
def test_mfma_selfcheck_passes_and_can_fail(vfr):
    """What the pre-filter's margins assume about the matrix pipe -- one v_mfma_f32_16x16x4_f32 = four fp32 fmas, k ascending,
    denormals kept -- checked on THIS device against an explicit fmaf chain on adversarial rows (wide exponents, cancelling
    neighbours, denormal operands): no element differs; against the k-descending chain many do (the check can fail).  The
    scoring wrapper runs it once per process before the first "mfma" call and falls back to the exact kernels otherwise."""
    assert vfr.mfma_selfcheck("cuda:0") == 0
    assert vfr.mfma_selfcheck("cuda:0", reversed_reference=True) > 0
    assert vfr._mfma_mode_ok(torch.device("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mfma", "bf16"])
def test_prefilter_bank_products_reused_only_while_valid(vfr, mode):
    """Serving: query batches of different sizes against one resident bank and one workspace reuse the pre-filter's bank-side
    products (VFR_MFMA_BANK_READY); an in-place edit of the embeddings, another bank object, or an exact-mode call in
    between must recompute.  Every result == the same call on a fresh workspace."""
    vfr.set_option("score_mfma_min", 0)
    try:
        rs = np.random.RandomState(3)
        counts = rs.randint(1, 22, 300); counts[5] = 21
        off = synth.clip_offsets(counts)
        V = dev(rs.randn(int(off[-1]), 100).astype(np.float32) * 0.3 + 0.5)
        bank = vfr.VideoBank(V.clone(), dev(off.astype(np.int32)))
        ws = vfr.topk_workspace(200, 300, 50, DEV, total_clips=int(off[-1]))

        def call(b, nq, workspace, k=50):
            Q = dev(np.random.RandomState(nq).randn(nq, 100).astype(np.float32) * 0.3 + 0.5)
            dense = vfr.score_moments(Q, b)
            order = torch.argsort(dense, dim=1, stable=True)
            rd = torch.stack([dense.gather(1, order[:, 7:8]).squeeze(1), dense.gather(1, order[:, 900:901]).squeeze(1)]).contiguous()
            ri = torch.stack([order[:, 7], order[:, 900]]).contiguous()
            return vfr.score_topk(Q, b, k, rd, ri, workspace=workspace, mode=mode)

        def check(b, nq, expect_reuse):
            before = getattr(ws, "_vfr_bank", None)
            assert (before == b.prep_token(vfr.SCORE_MODES[mode])) == expect_reuse
            got, want = call(b, nq, ws), call(b, nq, None)
            for x, y in zip(got, want):
                assert torch.equal(x, y)

        check(bank, 200, False)
        check(bank, 64, True)                       # same bank, smaller batch: reused
        check(bank, 1, True)
        bank.emb.mul_(1.5)                          # in-place edit: torch's version counter moves, products recomputed
        check(bank, 64, False)
        check(bank, 64, True)
        bank2 = vfr.VideoBank(V * 0.5, dev(off.astype(np.int32)))
        check(bank2, 64, False)                     # another bank through the same workspace
        check(bank, 64, False)                      # ... and back: the workspace holds bank2's products
        vfr.score_topk(dev(rs.randn(8, 100).astype(np.float32)), bank, 10, workspace=ws, mode="exact")
        check(bank, 64, False)                      # the exact kernels carved the workspace their own way
        # writes torch does not version (the host-side token still matches, so the call CLAIMS VFR_MFMA_BANK_READY): the
        # library hashes the bank on the device, finds the signature stale and recomputes -- results stay right
        check(bank, 64, True)
        bank.emb.data.mul_(0.5)                     # whole tensor through .data: _version does not move
        check(bank, 64, True)
        alias = torch.from_dlpack(torch.utils.dlpack.to_dlpack(bank.emb))
        alias[int(off[150]) + 1, 17] += 0.25        # ONE element of one clip row, through a DLPack alias
        check(bank, 64, True)
        bank.emb.data[-1, 99] = -bank.emb.data[-1, 99]   # the last word of the bank
        check(bank, 64, True)
        bank.invalidate()                           # the explicit way: the claim is not even made
        check(bank, 64, False)
    finally:
        vfr.set_option("score_mfma_min", 128)


@pytest.mark.gpu
@pytest.mark.parametrize("normlang", [False, True])
def test_bilstm_few_queries_vector_chain(vfr, oracle, normlang):
    """Batches of 1-4 queries (a serving request) take the vector-chain LSTM step (`lstm_small`; one or two queries: the
    single-launch sequence kernel, `lstm_persist`): same bits as the MFMA tile path forced on the same rows, and as the
    oracle; incl. an all-pad query and the normalised-length embedding."""
    sd = synth.model_weights(4096, seed=13, normalize_lang=normlang)
    tokens = synth.query_tokens(4, seed=13)
    tokens[2, :] = 0                                                  # an all-pad query
    lt = sd.get("learnable_length.weight")
    rest = (dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()}, dev(sd["lang_fc.weight"]),
            dev(sd["lang_fc.bias"]), dev(lt) if lt is not None else None)
    want = oracle.bilstm_final(tokens, sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"], lt)
    for B in (1, 2, 3, 4):
        try:
            vfr.set_option("lstm_small", 4)
            vfr.set_option("lstm_persist_min", 32)      # (3 and 4 queries on the vector-chain step, not the MFMA sequence kernel)
            small = vfr.bilstm_final(dev(tokens[:B]), *rest)
            vfr.set_option("lstm_small", 0)
            vfr.set_option("lstm_persist", 0)
            tiles = vfr.bilstm_final(dev(tokens[:B]), *rest)
        finally:
            vfr.set_option("lstm_small", 2)
            vfr.set_option("lstm_persist", 1)
            vfr.set_option("lstm_persist_min", 2)
        assert torch.equal(small.view(torch.int32), tiles.view(torch.int32)), B
        assert same(small, want[:B]), B
        if B <= 2:
            # one or two queries: `small` above ran the whole sequence in ONE launch (weights resident in LDS, h handed from
            # workgroup to workgroup as tagged granules: lstm_persist); the same batch one launch per step, and the
            # single-launch form again (its granule buffers are re-zeroed every call)
            try:
                vfr.set_option("lstm_persist", 0)
                vfr.set_option("lstm_small", 4)
                steps = vfr.bilstm_final(dev(tokens[:B]), *rest)
                if B == 1:                                                # the four-wave weight stream vs the one-wave step
                    vfr.set_option("lstm_small4", 0)
                    one_wave = vfr.bilstm_final(dev(tokens[:1]), *rest)
                    assert torch.equal(steps.view(torch.int32), one_wave.view(torch.int32))
            finally:
                vfr.set_option("lstm_small4", 1)
                vfr.set_option("lstm_persist", 1)
                vfr.set_option("lstm_small", 2)
            assert torch.equal(small.view(torch.int32), steps.view(torch.int32)), B
            for _ in range(3):
                again = vfr.bilstm_final(dev(tokens[:B]), *rest)
                assert torch.equal(small.view(torch.int32), again.view(torch.int32)), B


@pytest.mark.gpu
@pytest.mark.parametrize("normlang", [False, True])
def test_bilstm_mid_batches_sequence_kernel(vfr, oracle, normlang):
    """3-64 queries at the model's shape run the whole BiLSTM sequence in ONE launch on the matrix pipe (lstm_seq_mfma_kernel:
    weights in registers as MFMA B fragments, h handed between workgroups as tagged granules): same bits as the MFMA tile
    steps (`lstm_persist` 0) and as the oracle; one and two row tiles, batches that do not fill a tile, an all-pad query,
    repeated calls (the granule buffers are re-zeroed every call)."""
    sd = synth.model_weights(4096, seed=17, normalize_lang=normlang)
    tokens = synth.query_tokens(64, seed=17)
    tokens[5, :] = 0
    tokens[40, :] = 0
    lt = sd.get("learnable_length.weight")
    rest = (dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()}, dev(sd["lang_fc.weight"]),
            dev(sd["lang_fc.bias"]), dev(lt) if lt is not None else None)
    want = oracle.bilstm_final(tokens, sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"], lt)
    for B in (3, 8, 16, 17, 32, 33, 49, 64):            # (above 32: two parts taking turns on the same resident weights -- an
        vfr.set_option("lstm_persist_max", 64)           #  option, slower than the tile steps there: default limit 32)
        try:
            seq = vfr.bilstm_final(dev(tokens[:B]), *rest)
        finally:
            vfr.set_option("lstm_persist_max", 32)
        try:
            vfr.set_option("lstm_persist", 0)
            tiles = vfr.bilstm_final(dev(tokens[:B]), *rest)
        finally:
            vfr.set_option("lstm_persist", 1)
        assert torch.equal(seq.view(torch.int32), tiles.view(torch.int32)), B
        assert same(seq, want[:B]), B
        if B <= 32:
            again = vfr.bilstm_final(dev(tokens[:B]), *rest)
            assert torch.equal(seq.view(torch.int32), again.view(torch.int32)), B


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 5, 20])
def test_sequence_kernel_give_up_is_repaired_and_reported(vfr, oracle, B):
    """The single-launch sequence kernels wait on each other's h every step.  Test hook `lstm_persist_fault`: one workgroup
    withholds its h of step 1 (what a workgroup that never became resident looks like to the others).  The call must RETURN
    -- bounded sweeps, no hang -- with the RIGHT embeddings (the rescue kernel enqueued behind the sequence kernel re-encodes
    the batch: == the undisturbed call and == the oracle, bit for bit; never NaN with a success code), and the event must
    reach the host: `poll_faults()` returns VFR_FAULT_SEQ_RESCUED, warns, and leaves the explanation in `vfr_last_error()`."""
    import time
    import warnings
    sd = synth.model_weights(4096, seed=19)
    tokens = synth.query_tokens(24, seed=19)
    rest = (dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()}, dev(sd["lang_fc.weight"]),
            dev(sd["lang_fc.bias"]), None)
    good = vfr.bilstm_final(dev(tokens[:B]), *rest)
    torch.cuda.synchronize()
    assert vfr.poll_faults() == 0
    try:
        vfr.set_option("lstm_persist_fault", 7)
        t0 = time.perf_counter()
        repaired = vfr.bilstm_final(dev(tokens[:B]), *rest)
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
    finally:
        vfr.set_option("lstm_persist_fault", -1)
    assert took < 20.0
    assert not bool(torch.isnan(repaired).any())
    assert torch.equal(repaired.view(torch.int32), good.view(torch.int32))
    want = oracle.bilstm_final(tokens[:B], sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
    assert same(repaired, want)
    n_log = len(vfr.FAULT_LOG)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        bits = vfr.poll_faults()
    assert bits & vfr.FAULT_SEQ_RESCUED
    assert any("re-encoded" in str(w.message) for w in caught) and len(vfr.FAULT_LOG) == n_log + 1
    assert "sequence encoder gave up" in vfr.lib().vfr_last_error().decode()
    assert vfr.poll_faults() == 0                                  # reported once
    again = vfr.bilstm_final(dev(tokens[:B]), *rest)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), good.view(torch.int32)) and vfr.poll_faults() == 0


@pytest.mark.gpu
def test_bilstm_select_free_step_agrees(vfr, oracle):
    """The table-start LSTM step has an instantiation without selects in its K-loop (`lstm_fast`: one segment of whole K-tiles,
    the remainder as direct fragments), picked when the launch qualifies (H = 1000: 31 tiles + 8; H = 96: 3 tiles): same bits as
    the general form and as the oracle, 32- and 64-row tiles, a batch with all-pad and short queries."""
    for H, B in ((1000, 300), (1000, 900), (96, 130)):
        sd = synth.model_weights(4096, seed=21, hidden=H)
        tokens = synth.query_tokens(B, seed=21)
        tokens[3, :] = 0
        args = (dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
                dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]))
        fast = vfr.bilstm_final(*args)
        try:
            vfr.set_option("lstm_fast", 0)
            general = vfr.bilstm_final(*args)
        finally:
            vfr.set_option("lstm_fast", 1)
        assert torch.equal(fast.view(torch.int32), general.view(torch.int32)), (H, B)
        n = 40
        want = oracle.bilstm_final(tokens[:n], sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
        sub = vfr.bilstm_final(dev(tokens[:n]), *args[1:])
        assert same(sub, want), (H, B)


@pytest.mark.gpu
def test_bilstm_multi_step_launch_agrees(vfr, oracle):
    """EXPERIMENT `lstm_multi` (off by default): all T steps of both directions in ONE launch (`lstm_steps_mfma_kernel`: one
    ordered task list per XCD group, completion counters per row tile, three rotating state buffers, the recurrent state moved
    with agent-scope accesses) -- the same bits as one launch per step and as the oracle: 32-, 64- and 128-row tiles, H = 96
    (three column tiles: five of the eight groups have nothing to do), a short sequence, an all-pad query."""
    for H, B, T, tile in ((1000, 300, 20, 0), (1000, 900, 20, 0), (1000, 900, 20, 2), (96, 130, 20, 0), (1000, 200, 5, 0)):
        sd = synth.model_weights(4096, seed=23, hidden=H)
        tokens = synth.query_tokens(B, seed=23)[:, :T].copy()
        tokens[3, :] = 0
        args = (dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
                dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]))
        per_step = vfr.bilstm_final(*args)
        try:
            vfr.set_option("lstm_multi", 1)
            vfr.set_option("lstm_tile", tile)
            one = vfr.bilstm_final(*args)
            torch.cuda.synchronize()
        finally:
            vfr.set_option("lstm_multi", 0)
            vfr.set_option("lstm_tile", 0)
        assert vfr.poll_faults() == 0
        assert torch.equal(one.view(torch.int32), per_step.view(torch.int32)), (H, B, T, tile)
        # rows are encoded independently of each other: the oracle on a slice of the batch (its C loop is slow) must match that slice
        rows = np.r_[0:24, B - 16:B]
        want = oracle.bilstm_final(tokens[rows], sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
        assert same(one[torch.from_numpy(rows).to(one.device)], want), (H, B, T, tile)


@pytest.mark.gpu
def test_multi_step_launch_give_up_is_repaired_and_reported(vfr, oracle):
    """The multi-step launch waits on completion counters.  Test hook `lstm_persist_fault`: one task never signals; its
    dependents must give up (bounded waits, no hang), every workgroup leaves, and the rescue kernel behind the launch
    re-encodes every row: the call returns the RIGHT embeddings and the event reaches the host through the fault word."""
    import time
    import warnings
    sd = synth.model_weights(4096, seed=29)
    tokens = synth.query_tokens(100, seed=29)
    args = (dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
            dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]))
    good = vfr.bilstm_final(*args)
    torch.cuda.synchronize()
    assert vfr.poll_faults() == 0
    try:
        vfr.set_option("lstm_multi", 1)
        vfr.set_option("lstm_persist_fault", 5)
        t0 = time.perf_counter()
        repaired = vfr.bilstm_final(*args)
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
    finally:
        vfr.set_option("lstm_persist_fault", -1)
        vfr.set_option("lstm_multi", 0)
    assert took < 20.0
    assert not bool(torch.isnan(repaired).any())
    assert torch.equal(repaired.view(torch.int32), good.view(torch.int32))
    want = oracle.bilstm_final(tokens, sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
    assert same(repaired, want)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        bits = vfr.poll_faults()
    assert bits & vfr.FAULT_SEQ_RESCUED and any("re-encoded" in str(w.message) for w in caught)
    assert vfr.poll_faults() == 0


@pytest.mark.gpu
def test_bilstm_tile_shapes_agree(vfr, oracle):
    """The fused LSTM step picks 32-, 64- or 128-row tiles by batch size; all are the same chains.  Forced either way on one
    batch: identical bits to each other and (on the rows the oracle is run for) to the oracle."""
    sd = synth.model_weights(4096, seed=7)
    tokens = synth.query_tokens(1500, seed=7)
    args = (dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
            dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]))
    outs = {}
    try:
        for mode in (1, 2, 3):
            vfr.set_option("lstm_tile", mode)
            outs[mode] = vfr.bilstm_final(*args)
    finally:
        vfr.set_option("lstm_tile", 0)
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[3])
    want = oracle.bilstm_final(tokens[:64], sd["word_embedding.weight"], lstm_of(sd), sd["lang_fc.weight"], sd["lang_fc.bias"])
    assert same(outs[2][:64], want)
    # the first step skips its recurrent segment (h_0 = 0 -> every term is fma(0, w, acc) == acc): same bits as running it;
    # the XCD-aware workgroup order is a permutation of the same tiles
    for opt in ("lstm_skip0", "lstm_xcd"):
        try:
            vfr.set_option(opt, 0)
            other = vfr.bilstm_final(*args)
        finally:
            vfr.set_option(opt, 1)
        assert torch.equal(other.view(torch.int32), outs[1].view(torch.int32)), opt


@pytest.mark.gpu
def test_gemm_ping_pong_variant_is_bit_identical(vfr):
    """The experimental ping-pong schedule (two tile groups per 512-thread workgroup) walks k in the same order."""
    rs = np.random.RandomState(3)
    A, W, b = (dev(rs.randn(4100, 512).astype(np.float32)), dev(rs.randn(2050, 512).astype(np.float32)),
               dev(rs.randn(2050).astype(np.float32)))
    sd = synth.model_weights(4096, seed=7)
    tokens = synth.query_tokens(1500, seed=7)
    args = (dev(tokens), dev(sd["word_embedding.weight"]), {k: dev(v) for k, v in lstm_of(sd).items()},
            dev(sd["lang_fc.weight"]), dev(sd["lang_fc.bias"]))
    out = {}
    try:
        for pp in (0, 1):
            vfr.set_option("gemm_pp", pp)
            vfr.set_option("lstm_tile", 2)
            out[pp] = (vfr.linear(A, W, b, relu=True), vfr.bilstm_final(*args))
    finally:
        vfr.set_option("gemm_pp", 0)
        vfr.set_option("lstm_tile", 0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,nl", [("plain", False), ("normalized", True)])
def test_ranking_loss_forward_backward(vfr, oracle, golden, tag, nl):
    """Trainer.ranking_loss on the device: forward == oracle bits (and the reference within 1e-6), autograd gradients ==
    the reference's autograd within 1e-5."""
    from vfr_amd import losses
    g = golden("g7_ranking_loss.npz")
    posit, intra, inter, lang, maskp, maskn = synth.ranking_batch(41)
    t = [dev(a).clone().requires_grad_(True) for a in (posit, intra, inter, lang)]
    loss, n = losses.ranking_loss(*t, dev(maskp), dev(maskn), normalize_loss=nl)
    (loss * 1.0).backward()
    assert n == int(g[f"n_{tag}"])
    assert loss.item() == pytest.approx(float(g[f"loss_{tag}"]), rel=1e-6)
    if not nl:
        want, _ = oracle.ranking_loss(posit, intra, inter, lang, maskp, maskn)
        assert np.float32(loss.item()) == want                                  # bit-exact forward
    for name, x in zip(("posit", "intra", "inter", "lang"), t):
        # (grad_lang sums ~15 signed row terms per element: different summation order, so an absolute floor too)
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"grad_{name}_{tag}"], rtol=2e-5, atol=2e-7)


@pytest.mark.gpu
def test_fused_scorer_randomised_differential(vfr, score_mode):
    """A dozen random corpus shapes (uniform / ragged clip counts, tiny to mid-size, duplicated videos = exact ties, odd query
    counts, k from 1 to 300): the fused pass == dense kernel + stable sort, rank counts exact (tools/score_fuzz.py is the
    longer form of this)."""
    rs = np.random.RandomState(11)
    for it in range(12):
        shape = ["n21", "n6", "ragged56", "ragged21"][it % 4]
        nv = int(rs.choice([1, 3, 31, 33, 100, 257, 600, 1100]))
        nq = int(rs.choice([1, 5, 63, 64, 65, 200]))
        k = int(rs.choice([1, 10, 100, 128, 300]))
        counts = {"n21": np.full(nv, 21), "n6": np.full(nv, 6), "ragged56": rs.choice([5, 6], nv),
                  "ragged21": np.maximum(rs.randint(1, 22, nv), 21 * (np.arange(nv) == 0))}[shape]
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        g = torch.Generator(device=DEV).manual_seed(100 + it)
        V = torch.randn((int(off[-1]), 100), device=DEV, generator=g) * 0.1
        Q = torch.randn((nq, 100), device=DEV, generator=g) * 0.1
        if it % 3 == 0 and nv > 2:
            for v in range(1, nv):
                if counts[v] == counts[0]:
                    V[off[v]:off[v + 1]] = V[off[0]:off[1]]
                    break
        bank = vfr.VideoBank(V, dev(off))
        dense = vfr.score_moments(Q, bank)
        total = dense.shape[1]
        order = torch.argsort(dense, dim=1, stable=True)
        kk = min(k, total)
        p0, p1 = int(rs.randint(0, total)), int(rs.randint(0, total))
        rd = torch.stack([dense.gather(1, order[:, p0:p0 + 1]).squeeze(1), dense.gather(1, order[:, p1:p1 + 1]).squeeze(1)])
        ri = torch.stack([order[:, p0], order[:, p1]])
        d, i, c = vfr.score_topk(Q, bank, k, rd.contiguous(), ri.contiguous())
        assert torch.equal(i[:, :kk], order[:, :kk]) and torch.equal(d[:, :kk], dense.gather(1, order[:, :kk])), (it, shape, nv, nq, k)
        assert kk == k or bool((i[:, kk:] == -1).all())
        assert c[0].tolist() == [p0] * nq and c[1].tolist() == [p1] * nq, (it, shape, nv, nq, k)


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [True, False])
def test_feature_store_streamed_encode_is_bit_identical(vfr, oracle, pinned, tmp_path):
    """the packed store + chunked H2D overlapped with the clip encoder (engine.encode_clips_streamed)
    == the one-launch encoder on the same rows, bit for bit (chunk sizes chosen to give ragged, odd chunk counts), and
    == the oracle on a sample of videos."""
    from vfr_amd import engine
    from vfr_amd.store import FeatureStore
    counts = synth.clip_counts(700, "didemo", seed=21)
    seg, ctx = synth.video_features(counts, 4096, seed=21)
    off = synth.clip_offsets(counts)
    names = [f"v{i}" for i in range(len(counts))]
    FeatureStore.write(tmp_path / "c.vfs", names, off, ctx, seg)
    st = FeatureStore.open(tmp_path / "c.vfs", pin=pinned)
    assert st.seg.is_pinned() == pinned and np.array_equal(st.seg.numpy(), seg)
    model = make_model(synth.model_weights(4096, seed=21)).to(DEV)
    ops = engine.ops_for(DEV)
    whole = model.encode_clips(st.seg.to(DEV), st.ctx.to(DEV), st.clip_off.to(DEV))
    for chunk in (257, 1000, 4096):
        part = engine.encode_clips_streamed(ops, model, st.seg, st.ctx, st.clip_off.numpy(), DEV, chunk_clips=chunk)
        assert torch.equal(part, whole), chunk
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    v0, v1 = 300, 308
    want = oracle.visual_mlp(seg[off[v0]:off[v1]], ctx[v0:v1], (off[v0:v1 + 1] - off[v0]).astype(np.int32),
                             sd["visual_fc.0.weight"], sd["visual_fc.0.bias"], sd["visual_fc.2.weight"], sd["visual_fc.2.bias"])
    assert np.array_equal(whole[off[v0]:off[v1]].cpu().numpy(), want)
    # through build_corpus: a host bank above the streaming threshold takes the streamed path
    old = engine.STREAM_CHUNK_CLIPS
    engine.STREAM_CHUNK_CLIPS = 512
    try:
        shard = engine.build_corpus(model, st.feature_bank(), DEV, ops)
    finally:
        engine.STREAM_CHUNK_CLIPS = old
    assert torch.equal(shard.bank.emb, whole)


@pytest.mark.gpu
def test_feature_store_packed_on_gpu_feeds_evaluate(vfr, oracle, tmp_path):
    """FeatureStore.from_npy pools with the HIP kernel (== oracle bits); CustomDataset then opens the .vfs instead of the
    .npy files and evaluate() gives the same dict on the device as on the CPU path."""
    from vfr_amd import data as vdata
    from vfr_amd import evaluate as vevaluate
    from vfr_amd.store import FeatureStore
    d = tmp_path / "features_vgg19"
    d.mkdir()
    lengths = (150, 138, 125, 112, 60, 25, 7)
    names = [f"vid{T}" for T in lengths]
    for T in lengths:
        np.save(d / f"vgg19_ft_vid{T}.npy", np.random.RandomState(2000 + T).rand(T, 4096).astype(np.float32))
    st = FeatureStore.from_npy(tmp_path / "features_vgg19.vfs", names, str(tmp_path), "vgg19", "avg", pool_device=DEV,
                               chunk_videos=3)
    for T, name in zip(lengths, names):
        oseg, octx = oracle.segment_pool_norm(np.load(d / f"vgg19_ft_{name}.npy"), 25, "avg")
        seg_v, ctx_v = st.video_rows(name)
        assert np.array_equal(seg_v.numpy(), oseg) and np.array_equal(ctx_v.numpy(), octx)
    words = [f"w{i}" for i in range(1, 40)]
    with open(tmp_path / "glove.6B.100d.txt", "w") as fh:
        for i, w in enumerate(words + ["unk"]):
            fh.write(w + " " + " ".join([f"{0.01 * (i + 1):.2f}"] * 100) + "\n")
    wi = vdata.WordIndexer(str(tmp_path))
    rs = np.random.RandomState(4)
    annots = {q: dict(video=names[q % len(names)], description=" ".join(rs.choice(words, size=rs.randint(1, 12))),
                      times=[[0, 0], [0, 0], [0, 1]]) for q in range(21)}
    ds = vdata.CustomDataset(names, annots, str(tmp_path), "vgg19", word_indexer=wi, validate=True)
    assert ds.store is not None

    def iters():
        vi = torch.utils.data.DataLoader(ds, shuffle=False, collate_fn=vdata.validate_collate,
                                         batch_sampler=vdata.VideoBatchSampler(names, ds.num_segments_info))
        li = torch.utils.data.DataLoader(ds, shuffle=False, collate_fn=vdata.validate_collate,
                                         batch_sampler=vdata.LanguageBatchSampler(annots, ds.num_segments_info))
        return vi, li
    model = make_model(synth.model_weights(4096, vocab=wi.get_items_count(), seed=9))
    cpu = vevaluate.evaluate(model, *iters(), annots, "cpu")
    gpu = vevaluate.evaluate(model.to(DEV), *iters(), annots, DEV)
    assert gpu == cpu


@pytest.mark.gpu
def test_exchange_key_helpers_match_cpu_provider(vfr):
    """vfr_topk_pack_keys / vfr_topk_merge_keys / vfr_gt_best_keys_f32 (the exchange side of SURVEY 8e) against the
    plain-torch CPU provider: unsorted lists, empty slots, duplicates across lists, G*k above and below the pool size."""
    from vfr_amd import engine
    cpu = engine.TorchCpuOps()
    rs = np.random.RandomState(5)
    for G, Nq, k in ((9, 130, 100), (3, 50, 300), (20, 33, 7), (1, 5, 1), (2, 70, 448)):
        d = rs.rand(G, Nq, k).astype(np.float32)
        d[rs.rand(G, Nq, k) < 0.1] = 0.25                                   # ties on the distance -> id decides
        i = rs.randint(0, 5000, size=(G, Nq, k)).astype(np.int64)
        i[rs.rand(G, Nq, k) < 0.2] = -1                                     # empty slots anywhere
        if G > 1:
            d[1], i[1] = d[0], i[0]                                         # a whole duplicated list
        td, ti = torch.from_numpy(d), torch.from_numpy(i)
        wk = cpu.pack_keys(td, ti)
        gk = vfr.topk_pack_keys(td.to(DEV), ti.to(DEV))
        assert torch.equal(gk.cpu(), wk)
        wd, wi, wkeys = cpu.merge_keys(wk, True, True)
        gd, gi, gkeys = vfr.topk_merge_keys(gk, True, True)
        assert torch.equal(gkeys.cpu(), wkeys) and torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)
        assert vfr.topk_merge_keys(gk, False, True)[2].equal(gkeys)
        md, mi = vfr.topk_merge(td.to(DEV), ti.to(DEV))                     # (dist, idx) form: same kernel, other loader
        assert torch.equal(mi.cpu(), wi) and torch.equal(md.cpu(), wd)
    for R, n_sel, Ms, Ml, Nq in ((2, 300, 231, 231, 900), (11, 40, 21, 28, 40), (1, 1, 3, 3, 2), (2, 0, 6, 6, 10)):
        sc = rs.rand(n_sel, Ms).astype(np.float32)
        sc[:, Ms - 1] = np.inf
        lab = rs.rand(R, n_sel, Ml) < 0.15
        if n_sel:
            lab[0, 0] = False                                                # a query without any positive
        base = rs.randint(0, 1 << 20, size=n_sel).astype(np.int64)
        sel = rs.permutation(Nq)[:n_sel].astype(np.int64)
        want = cpu.gt_best_keys(torch.from_numpy(sc), torch.from_numpy(lab), torch.from_numpy(base), torch.from_numpy(sel), Nq)
        if n_sel:
            got = vfr.gt_best_keys(torch.from_numpy(sc).to(DEV), torch.from_numpy(lab).to(DEV),
                                   torch.from_numpy(base).to(DEV), torch.from_numpy(sel).to(DEV), Nq)
            assert torch.equal(got.cpu(), want)
            assert int(got[0, sel[0]]) == engine.KEY_INF


@pytest.mark.gpu
def test_rccl_world1_runs_the_sharded_protocol(vfr):
    """The exchange steps of the sharded pass through REAL RCCL calls on this one GPU: a world-size-1 "nccl"
    process group in a child process (tests/rccl_world1_worker.py) with engine.FORCE_COLLECTIVES, so the packed all-gathers
    (``all_gather_into_tensor`` on device int64 / fp32 rows) and the MIN / SUM all-reduces are issued exactly as on N GPUs.
    evaluate() -- the three-collective fused protocol -- and validate_epoch() -- 11 thresholds, five-collective form -- must
    equal the plain single-GPU pass: dicts, top-100 ids and distances."""
    import json
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1_worker.py")
    res = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")][-1]
    out = json.loads(line[len("RCCL_WORLD1 "):])
    assert out["backend"] == "nccl" and out["gather_form"] == "tensor"
    assert out["fused_collectives"] == {"all_gather_into_tensor": 3, "all_reduce": 0}      # evaluate(): three collectives
    assert out["calls"]["all_reduce"] > 0                                                   # validate_epoch(): MIN + SUM folds
    assert out["evaluate_equal"] and out["validate_equal"] and out["topk_ids_equal"] and out["topk_dist_equal"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_evaluate_through_kernels_equals_single(vfr, world, score_mode):
    """engine's world > 1 flow (sample exchange -> seed -> main pass -> exchange -> merge; keys MIN, counts SUM; query
    slices gathered) on the real kernels: N ranks as N threads of this process with thread-barrier collectives
    (helpers.ThreadRanks).  Every rank's dict and top-k == the single-rank pass."""
    from helpers import ThreadRanks
    from vfr_amd import engine
    from vfr_amd import evaluate as vevaluate
    p = problem(700, 150, "didemo", seed=31)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    model = make_model(p["sd"]).to(DEV)
    want, (wd, wi) = vevaluate.evaluate(model, *ds.iterators(), ds.annotations, DEV, return_topk=100)
    wv = vevaluate.validate_epoch(model, *ds.iterators(), ds.annotations, DEV, size=-1)
    group = ThreadRanks(world)
    old = engine._dist
    engine._dist = lambda: group
    try:
        def body(rank, n):
            torch.cuda.set_device(0)
            res, (d, i) = vevaluate.evaluate(model, *ds.iterators(), ds.annotations, DEV, rank=rank, world=n, return_topk=100)
            v = vevaluate.validate_epoch(model, *ds.iterators(), ds.annotations, DEV, size=-1, rank=rank, world=n)
            return res, d.cpu(), i.cpu(), v
        outs = group.run(body)
    finally:
        engine._dist = old
    for res, d, i, v in outs:
        assert res == want and v == wv
        assert torch.equal(i, wi.cpu()) and torch.equal(d, wd.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("clips", [6, "didemo", 21])
def test_gt_labels_kernel_equals_host_table(vfr, clips):
    """a11 on the device (vfr_gt_labels_u8) == the host table == utils.get_iou's float64 expression, ragged annotators,
    > and >= forms, 11 thresholds (validate_epoch's sweep)."""
    from vfr_amd import engine
    counts = synth.clip_counts(500, clips, seed=6)
    own, times = synth.annotations(777, counts, seed=6)
    times = [t if i % 4 else t + [t[2], [0, 1], [1, 1]] for i, t in enumerate(times)]
    ops = engine.HipOps()
    for strict, thrs in ((True, [0.5, 0.7]), (False, [i / 10 for i in range(11)])):
        want = engine.gt_label_table(times, counts[own], thrs, strict=strict)
        got = engine.gt_labels(times, counts[own], thrs, strict, DEV, ops)
        assert got.dtype == torch.bool and got.is_cuda and np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_vgg19_full_size_bit_exact_and_torch_fixture(vfr, oracle, golden):
    """BASELINE config 3 at full size: 224x224 frames, VGG-19 "E" widths 64..512, fc6 25088 -> 4096, fc7 -- conv1_1's Cin 3 -> 4
    padding, the 128x64 tiles at 100k-pixel M, fc6 at K = 25088.  HIP == oracle bit for bit on 2 frames; frame 0 == fixture G9
    (torch.nn modules, generated from the reference's layer list) within 1e-4 of the activation scale."""
    from test_oracle_vgg_torch import VGG19_E
    cw, cb, fc6, fc7 = synth.vgg_weights(VGG19_E, (224, 224), 4096, seed=5)
    frames = np.concatenate([synth.frames_u8(1, 224, 224, seed=5), synth.frames_u8(1, 224, 224, seed=6)])
    got = vfr.vgg_fc7(dev(frames), VGG19_E, [dev(w) for w in cw], [dev(b) for b in cb],
                      (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    want = oracle.vgg_fc7(frames, cw, cb, fc6, fc7, VGG19_E)
    assert same(got, want)
    ref = golden("g9_vgg_full.npz")["fc7"]
    np.testing.assert_allclose(got.cpu().numpy()[:1], ref, rtol=0, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    # the extractor's chunked pass (160-frame chunks) over a full 150-frame video: every frame independent of its batch
    rs = np.random.RandomState(3)
    video = rs.randint(0, 256, size=(150, 224, 224, 3)).astype(np.uint8)
    video[7], video[149] = frames[0], frames[1]
    full = vfr.vgg_fc7(dev(video), VGG19_E, [dev(w) for w in cw], [dev(b) for b in cb],
                       (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    assert torch.equal(full[7], got[0]) and torch.equal(full[149], got[1])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo"), ("n21", 21)])
def test_bf16_scoring_tolerance_on_reference_embeddings(vfr, oracle, golden, tag, clips):
    """BASELINE config 5 (bf16 MFMA operands, fp32 accumulate) on the REFERENCE's own embeddings (fixture G2).  Stated
    tolerance: rank@1 identical to the fp32 path for >= 98 % of the queries and to the reference's argsort top-1 likewise,
    top-10 overlap >= 0.95, every returned distance an EXACT fp32 score of its moment (the list is re-ranked exactly), rank
    counts within 2 % of the number of moments."""
    g = golden(f"g2_scoring_{tag}.npz")
    Q, V, counts = g["query_emb"], g["visual_emb"], g["counts"]
    off = synth.clip_offsets(counts)
    bank = _bank(vfr, V, off)
    d0, i0, _ = vfr.score_topk(dev(Q), bank, 100, mode="exact")
    d1, i1, _ = vfr.score_topk(dev(Q), bank, 100, mode="bf16")
    i0n, i1n = i0.cpu().numpy(), i1.cpu().numpy()
    assert (i0n[:, 0] == i1n[:, 0]).mean() >= 0.98
    assert (i1n[:, 0] == g["top_idx"][:, 0]).mean() >= 0.98
    assert np.mean([len(set(a[:10]) & set(b[:10])) / 10 for a, b in zip(i0n, i1n)]) >= 0.95
    dense = oracle.score_moments(Q, V, off)
    assert np.array_equal(d1.cpu().numpy(), np.take_along_axis(dense, i1n, axis=1))       # exact scores of the listed moments
    assert bool((d1[:, 1:] >= d1[:, :-1]).all())
    rs = np.random.RandomState(2)
    pick = rs.randint(0, dense.shape[1], size=(2, Q.shape[0]))
    rd = np.take_along_axis(dense, pick.T, axis=1).T.copy()
    _, _, c0 = vfr.score_topk(dev(Q), bank, 0, dev(rd), dev(pick.astype(np.int64)), mode="exact")
    _, _, c1 = vfr.score_topk(dev(Q), bank, 0, dev(rd), dev(pick.astype(np.int64)), mode="bf16")
    assert float((c0 - c1).abs().max()) <= 0.02 * dense.shape[1]


@pytest.mark.gpu
def test_bf16_scoring_tolerance_full_corpus(vfr, oracle):
    """Config 5 at BASELINE size (10k videos x 21 clips, 256-query batch): rank@1 agreement >= 0.99, top-10 overlap >= 0.98,
    top-100 overlap >= 0.95 against the fp32 path, rank counts within 1 % of the 2.31 M moments.  The fp32 side of the
    comparison is itself checked against the CPU oracle on four of the queries over the full corpus (bit-identical), and
    the bf16 list of those queries against the oracle's directly."""
    torch.manual_seed(1)
    nv, n, nq, k = 10000, 21, 256, 100
    V = torch.randn(nv * n, 100, device=DEV) * 0.1
    Q = torch.randn(nq, 100, device=DEV) * 0.1
    off = torch.arange(0, nv * n + 1, n, dtype=torch.int32, device=DEV)
    bank = vfr.VideoBank(V, off)
    d0, i0, _ = vfr.score_topk(Q, bank, k, mode="mfma")
    rd = torch.stack([d0[:, 40], d0[:, 99] * 1.05]).contiguous()
    ri = torch.stack([i0[:, 40], i0[:, 99]]).contiguous()
    d0, i0, c0 = vfr.score_topk(Q, bank, k, rd, ri, mode="mfma")
    d1, i1, c1 = vfr.score_topk(Q, bank, k, rd, ri, mode="bf16")
    assert float((i0[:, 0] == i1[:, 0]).float().mean()) >= 0.99
    assert float((i0[:, :10, None] == i1[:, None, :10]).any(-1).float().mean()) >= 0.98
    assert float((i0[:, :, None] == i1[:, None, :]).any(-1).float().mean()) >= 0.95
    assert bool((c0[0] == 40).all())
    assert float((c0 - c1).abs().max()) <= 0.01 * nv * n * (n + 1) / 2
    sel = [0, 77, 128, 255]
    Vh, offh, Qh = V.cpu().numpy(), off.cpu().numpy(), Q[sel].cpu().numpy()
    od, oi = oracle.score_topk(Qh, Vh, offh, k)
    assert np.array_equal(i0[sel].cpu().numpy(), oi) and np.array_equal(d0[sel].cpu().numpy(), od)
    oc = oracle.rank_of(Qh, Vh, offh, od[:, 40].copy(), oi[:, 40].copy())
    assert oc.tolist() == [40] * 4 and c0[0, sel].tolist() == [40] * 4
    bi = i1[sel].cpu().numpy()
    assert (bi[:, 0] == oi[:, 0]).mean() >= 0.75                                    # rank@1 vs the CPU reference path
    assert np.mean([len(set(bi[q, :10]) & set(oi[q, :10])) / 10 for q in range(4)]) >= 0.9


@pytest.mark.gpu
def test_mfma_prefilter_answers_itself_and_falls_back_on_duplicates(vfr, oracle):
    """The fp32 MFMA path must not be the exact kernels in disguise: on a generic corpus no query group is handed to the
    fallback and only a small fraction of the (query, video) pairs is re-scored exactly; on a corpus of duplicated videos
    (hundreds of exactly equal scores inside any margin) the groups ARE handed over -- and the result is still exact."""
    rs = np.random.RandomState(5)
    counts = synth.clip_counts(1500, 21, seed=5)
    off = synth.clip_offsets(counts)
    V = (rs.randn(int(off[-1]), 100) * 0.1).astype(np.float32)
    Q = (rs.randn(200, 100) * 0.1).astype(np.float32)
    bank = _bank(vfr, V, off)
    dense = oracle.score_moments(Q[:8], V, off)
    d, i, _ = vfr.score_topk(dev(Q), bank, 100, mode="exact")
    rd = torch.stack([d[:, 50], d[:, 99] * 1.1]).contiguous()
    ri = torch.stack([i[:, 50], i[:, 99]]).contiguous()
    ws = vfr.topk_workspace(200, 1500, 100, DEV, total_clips=int(off[-1]))
    d1, i1, c1 = vfr.score_topk(dev(Q), bank, 100, rd, ri, workspace=ws, mode="mfma")
    st = vfr.score_mfma_stats(ws, 200, bank, 100)
    assert st["fallback_groups"] == 0 and 0 < st["exact_pair_fraction"] < 0.3, st
    d0, i0, c0 = vfr.score_topk(dev(Q), bank, 100, rd, ri, mode="exact")
    assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(c0, c1)
    order = np.argsort(dense, axis=1, kind="stable")[:, :100]
    assert np.array_equal(i1.cpu().numpy()[:8], order)
    # duplicated videos: 300 copies of one video -> every best moment 300 times
    vfr.set_option("score_mfma_min", 0)
    Vd = np.tile(V[:21], (300, 1))
    offd = synth.clip_offsets(np.full(300, 21))
    bankd = _bank(vfr, Vd, offd)
    wsd = vfr.topk_workspace(200, 300, 100, DEV, total_clips=int(offd[-1]))
    dd, idd, _ = vfr.score_topk(dev(Q), bankd, 100, workspace=wsd, mode="mfma")
    assert vfr.score_mfma_stats(wsd, 200, bankd, 100)["fallback_groups"] > 0
    wd, wi = oracle.score_topk(Q, Vd, offd, 100)
    vfr.set_option("score_mfma_min", 128)
    assert same(idd, wi) and same(dd, wd)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "normlang"])
def test_encoder_backward_hip_matches_reference_gradients(vfr, golden, tag):
    """f2 second half: training-mode forward + backward of the clip MLP and the BiLSTM in HIP (train.py, csrc/train.hip)
    against the REFERENCE's gradients (fixture G10: loss.backward() through model/models.py's CALModel), 1e-4 of the scale."""
    from test_host_logic import _encoder_grad_case
    from vfr_amd import models as vmodels
    assert vmodels.HIP_TRAINING
    _encoder_grad_case(tag, DEV, golden)


@pytest.mark.gpu
def test_encoder_backward_hip_full_size_vs_torch_autograd(vfr):
    """The same at the real sizes (F = 4096, hidden 1000, 24 clips rows x 3, 40 queries), HIP autograd functions vs
    torch.nn's autograd on the same device and weights: forward 1e-4, gradients 2e-4 of their scale."""
    from vfr_amd import models as vmodels
    sd = synth.model_weights(4096, seed=8)
    rs = np.random.RandomState(9)
    x = dev(rs.rand(72, 8194).astype(np.float32))
    tok = dev(synth.query_tokens(40, seed=9))
    wv, wl = dev(rs.randn(72, 100).astype(np.float32)), dev(rs.randn(40, 100).astype(np.float32))
    res = {}
    for hip in (True, False):
        vmodels.HIP_TRAINING = hip
        try:
            m = make_model(sd).to(DEV).train()
            m.visual_fc[3].p = 0.0
            vis, lang = m(x), m(tok, False, DEV)
            ((vis * wv).sum() + (lang * wl).sum()).backward()
            res[hip] = (vis.detach(), lang.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            vmodels.HIP_TRAINING = True
    assert torch.allclose(res[True][0], res[False][0], rtol=0, atol=1e-4) and torch.allclose(res[True][1], res[False][1], rtol=0, atol=1e-4)
    assert set(res[True][2]) == set(res[False][2]) and len(res[True][2]) == 14
    for name, want in res[False][2].items():
        scale = max(1.0, float(want.abs().max()))
        assert float((res[True][2][name] - want).abs().max()) <= 2e-4 * scale, name


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,E,H", [(37, 7, 12, 20), (300, 5, 8, 132), (9, 6, 10, 6), (5, 4, 12, 7)])
def test_bilstm_training_paths_vs_torch_autograd(vfr, B, T, E, H):
    """train.bilstm_final (fused recurrence: E, H multiples of 4; step-by-step kernels otherwise) vs nn.LSTM's autograd:
    h_n 1e-5, gradients of the input and of all eight weights 1e-4 of their scale.  (300, 132) spans several row / column
    tiles and several split-K ranges."""
    from vfr_amd import train
    torch.manual_seed(B * 131 + H)
    lstm = torch.nn.LSTM(E, H, batch_first=True, bidirectional=True).to(DEV)
    x = torch.randn(B, T, E, device=DEV)
    w = torch.randn(B, 2 * H, device=DEV)
    res = []
    for hip in (True, False):
        lstm.zero_grad()
        xi = x.clone().requires_grad_(True)
        if hip:
            hn = train.bilstm_final(xi, lstm)
        else:
            hn = lstm(xi)[1][0].permute(1, 0, 2).reshape(B, 2 * H)
        (hn * w).sum().backward()
        res.append((hn.detach(), xi.grad.clone(), {n: p.grad.clone() for n, p in lstm.named_parameters()}))
    assert torch.allclose(res[0][0], res[1][0], rtol=0, atol=1e-5)
    for got, want, name in [(res[0][1], res[1][1], "x")] + [(res[0][2][n], res[1][2][n], n) for n in res[1][2]]:
        assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), name


@pytest.mark.gpu
def test_extractor_front_end_to_pooled_dataset(vfr, oracle, tmp_path):
    """f4 -> a2 -> a3 end to end: decoded frames -> frame sampling (get_rgb_features.py:45-60) -> HIP VGG stack -> one .npy per
    video (:150-151) -> data.CustomDataset pooling them (model/data.py:163-181).  The selected frames' features == the oracle's
    VGG on exactly the frames the reference's index arithmetic selects; reduced width for speed."""
    from vfr_amd import data as vdata, features
    cfg = SMALL_VGG
    cw, cb, fc6, fc7 = synth.vgg_weights(cfg, (32, 32), 64, seed=2)
    weights = ([dev(w) for w in cw], [dev(b) for b in cb], (dev(fc6[0]), dev(fc6[1])), (dev(fc7[0]), dev(fc7[1])))
    rs = np.random.RandomState(4)
    clips = {"a": (rs.randint(0, 256, (150, 32, 32, 3)).astype(np.uint8), 5.0, 6),     # 30 s at 5 fps: all frames kept? no: 150 of 150
             "b": (rs.randint(0, 256, (819, 32, 32, 3)).astype(np.uint8), 30.0, 6),    # 27.3 s at 30 fps -> 138 frames
             "c": (rs.randint(0, 256, (625, 32, 32, 3)).astype(np.uint8), 25.0, 5)}
    info = [dict(video=k, num_segments=v[2]) for k, v in clips.items()] + [dict(video="broken", num_segments=6)]
    decoder = lambda video, nseg: (None, 0) if video == "broken" else clips[video][:2]
    ft = tmp_path / "features_vgg19"
    written, missed = features.extract_dataset(info, decoder, ft, weights, cfg=cfg, missed_path=tmp_path / "missed.json")
    assert written == ["a", "b", "c"] and missed == ["broken"]
    for name, (frames, fps, nseg) in clips.items():
        got = np.load(ft / f"vgg19_ft_{name}.npy")
        mask = oracle.frame_sample_indices(len(frames), fps, nseg)
        want = oracle.vgg_fc7(frames[mask], cw, cb, fc6, fc7, cfg)
        assert got.shape == want.shape and np.array_equal(got, want), name
    assert np.load(ft / "vgg19_ft_b.npy").shape[0] == 138
    # the serial loop (pipeline=False) writes the same files and lists
    ft2 = tmp_path / "features_serial"
    written2, missed2 = features.extract_dataset(info, decoder, ft2, weights, cfg=cfg, missed_path=tmp_path / "missed2.json", pipeline=False)
    assert (written2, missed2) == (written, missed)
    for name in clips:
        assert np.array_equal(np.load(ft2 / f"vgg19_ft_{name}.npy"), np.load(ft / f"vgg19_ft_{name}.npy"))
    # a decoder that fails mid-way surfaces its exception and leaves no thread behind
    import threading
    before = threading.active_count()

    def bad_decoder(video, nseg):
        if video == "b":
            raise OSError("cannot open b.mp4")
        return clips[video][:2]
    with pytest.raises(OSError, match="b.mp4"):
        features.extract_dataset(info[:3], bad_decoder, tmp_path / "features_bad", weights, cfg=cfg, missed_path=tmp_path / "missed3.json")
    assert threading.active_count() == before


@pytest.mark.gpu
def test_mfma_prefilter_centres_offset_embeddings(vfr):
    """Embeddings that share a large common offset (norms 40x the distances): the pre-filter's margins are taken for the
    CENTRED rows, so it still answers itself -- no fallback group, a small exact fraction -- with the exact kernels' bits."""
    g = torch.Generator(device=DEV).manual_seed(3)
    nv, nq, n, k = 2000, 300, 21, 100
    V = torch.randn((nv * n, 100), device=DEV, generator=g) * 0.1 + 4.0
    Q = torch.randn((nq, 100), device=DEV, generator=g) * 0.1 + 4.0
    off = torch.arange(0, nv * n + 1, n, dtype=torch.int32, device=DEV)
    bank = vfr.VideoBank(V, off)
    d0, i0, _ = vfr.score_topk(Q, bank, k, mode="exact")
    rd = torch.stack([d0[:, 30], d0[:, 99] * 1.02]).contiguous()
    ri = torch.stack([i0[:, 30], i0[:, 99]]).contiguous()
    d0, i0, c0 = vfr.score_topk(Q, bank, k, rd, ri, mode="exact")
    ws = vfr.topk_workspace(nq, nv, k, DEV, total_clips=nv * n)
    d1, i1, c1 = vfr.score_topk(Q, bank, k, rd, ri, workspace=ws, mode="mfma")
    st = vfr.score_mfma_stats(ws, nq, bank, k)
    assert st["fallback_groups"] == 0 and st["exact_pair_fraction"] < 0.3, st
    assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(c0, c1) and bool((c0[0] == 30).all())


# ---------------------------------------------------------------------------------------------------------------------
# f4: the ResNet-152 variant of the extractor (get_rgb_features.py:127-131)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("hw,blocks,width,T", [((64, 64), (1, 2, 2, 1), 8, 3), ((96, 80), (2, 1, 3, 2), 4, 2), ((64, 96), (1, 1, 1, 1), 32, 5),
                                               ((224, 224), (1, 1, 2, 1), 16, 2)])
def test_resnet_stack_bit_exact_reduced(vfr, oracle, hw, blocks, width, T):
    """vfr_resnet_pool_f32 == the oracle bit for bit on reduced stacks that reach every layer form: the 7x7/2 stem through
    im2col, 3x3/2 max-pool, 1x1 dense GEMMs, the 3x3 implicit-GEMM loader (stride 1) and im2col (stride 2), downsample branches of
    stride 1 and 2, the residual epilogue, the global average; widths below and above one MFMA tile, odd spatial sizes."""
    H, W = hw
    frames = synth.frames_u8(T, H, W, seed=21)
    sd = synth.resnet_weights(blocks, width, seed=21)
    want = oracle.resnet_pool(frames, sd, blocks, width)
    got = vfr.resnet_pool(dev(frames), vfr.resnet_pack(sd, blocks, width, DEV), blocks, width)
    assert got.shape == (T, 32 * width) and same(got, want)


@pytest.mark.gpu
def test_resnet152_full_size_bit_exact_and_torch_fixture(vfr, oracle, golden):
    """f4 at FULL size: ResNet-152 (3 / 8 / 36 / 3, width 64, 224x224): frame 0 == the oracle bit for bit, both frames == fixture
    G12 (the same network from torch.nn modules) to 1e-4 of the activation scale; a pass over 170 frames (two chunks of the frame
    loop) reproduces the two-frame result row for row."""
    blocks, width = (3, 8, 36, 3), 64
    sd = synth.resnet_weights(blocks, width, seed=12)
    frames = synth.frames_u8(2, 224, 224, seed=12)
    packed = vfr.resnet_pack(sd, blocks, width, DEV)
    got = vfr.resnet_pool(dev(frames), packed)
    want = golden("g12_resnet_full.npz")["pooled"]
    assert got.shape == (2, 2048)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-4 * max(1.0, float(np.abs(want).max())))
    assert same(got[:1], oracle.resnet_pool(frames[:1], sd))
    many = np.concatenate([frames] * 85)                                     # 170 frames: two chunks of the frame loop (chunk <= 160)
    got170 = vfr.resnet_pool(dev(many), packed)
    assert torch.equal(got170[0::2], got[0:1].expand(85, -1)) and torch.equal(got170[1::2], got[1:2].expand(85, -1))


@pytest.mark.gpu
def test_extractor_resnet152_variant_front_end(vfr, oracle, tmp_path):
    """features.extract_dataset(model_type="resnet152"): frame selection + the ResNet stack + resnet152_ft_<video>.npy files the
    dataset side pools (model/data.py:22: FEATURE_DIM['resnet152'] = 32 * width here)."""
    from vfr_amd import features
    blocks, width = (1, 1, 1, 1), 8
    sd = synth.resnet_weights(blocks, width, seed=9)
    packed = vfr.resnet_pack(sd, blocks, width, DEV)
    rs = np.random.RandomState(5)
    clips = {"a": (rs.randint(0, 256, (150, 32, 32, 3)).astype(np.uint8), 5.0, 6), "b": (rs.randint(0, 256, (250, 32, 32, 3)).astype(np.uint8), 25.0, 2)}
    info = [dict(video=k, num_segments=v[2]) for k, v in clips.items()] + [dict(video="broken", num_segments=6)]
    decoder = lambda video, nseg: (None, 0) if video == "broken" else clips[video][:2]
    ft = tmp_path / "features_resnet152"
    written, missed = features.extract_dataset(info, decoder, ft, packed, model_type="resnet152", cfg=(blocks, width),
                                               missed_path=tmp_path / "missed.json")
    assert written == ["a", "b"] and missed == ["broken"]
    for name, (frames, fps, nseg) in clips.items():
        got = np.load(ft / f"resnet152_ft_{name}.npy")
        mask = oracle.frame_sample_indices(len(frames), fps, nseg)
        want = oracle.resnet_pool(frames[mask], sd, blocks, width)
        # the file has the reference's shape: children()[:-1] ends in the average pool -> [T, C, 1, 1] (get_rgb_features.py:129-131,151)
        assert got.shape == (want.shape[0], 32 * width, 1, 1) and np.array_equal(got.reshape(want.shape), want), name
