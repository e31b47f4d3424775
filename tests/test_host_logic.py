"""Host-side logic on the CPU device: utils / data mirror vs golden fixtures, and both evaluators driven
through the reference's iterator contract (BASELINE config 0: 100 videos x 50 queries, no GPU)."""
import json
import random
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import MemoryDataset, make_model, problem
from vfr_amd import data as vdata
from vfr_amd import synth
from vfr_amd import evaluate as vevaluate
from vfr_amd import evaluate_single as vsingle
from vfr_amd import utils as vutils


def test_moments_and_iou_match_reference(golden):
    g = golden("g3_moments_iou.npz")
    for n in list(range(7)) + [21]:
        mom = vutils.generate_moments(n)
        assert mom == [tuple(r) for r in g[f"moments_{n}"].tolist()]
        assert all(vutils.moment_index(n, s, e) == i for i, (s, e) in enumerate(mom))
    lens, flat, ious = g["times_len"], g["times_flat"], g["iou_flat"]
    p = o = 0
    for L in lens:
        times = flat[p:p + L].tolist()
        got = np.asarray([vutils.get_iou(times, s, e) for s, e in vutils.generate_moments(6)]).T.reshape(-1)
        assert np.array_equal(got, ious[o:o + got.size])
        p += L; o += got.size


def test_tokeniser_and_word_indexer_match_reference():
    g = json.load(open(Path(__file__).parent / "golden" / "g5_tokens.json"))
    words = [vdata.tokenize(d) for d in g["descriptions"]]
    assert words == g["words"]
    assert max(len(w) for w in words) > 20                       # the truncation case is present
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "glove.6B.100d.txt").write_text(g["glove_text"], encoding="UTF-8")
        wi = vdata.WordIndexer(td)
        assert wi.get_items_count() == g["vocab_size"]
        got = [wi.items2tensor([w], 20)[0].tolist() for w in words]
        assert got == g["tensors"]
        assert float(np.abs(wi.get_embeddings().numpy()).sum()) == pytest.approx(g["emb_checksum"], rel=1e-6)


@pytest.mark.parametrize("mode", ["avg", "max"])
def test_custom_dataset_pooling_host_path(golden, mode):
    g = golden("g4_pooling.npz")
    with tempfile.TemporaryDirectory() as td:
        d = Path(td) / "features_vgg19"
        d.mkdir()
        names = []
        for T in (150, 138, 125, 112):
            x = np.random.RandomState(1000 + T).rand(T, 4096).astype(np.float32)
            x[x < 0.3] = 0.0
            np.save(d / f"vgg19_ft_vid{T}.npy", x)
            names.append(f"vid{T}")
        ds = vdata.CustomDataset(names, {}, td, "vgg19", pooling=mode, pool_device="cpu")
        for T, name in zip((150, 138, 125, 112), names):
            vf = ds.video_features[name]
            assert vf["num_segments"] == int(g[f"nseg_{mode}_{T}"])
            np.testing.assert_allclose(vf["segment_features"], g[f"seg_{mode}_{T}"], rtol=0, atol=3e-7)
            np.testing.assert_allclose(vf["context_features"], g[f"ctx_{mode}_{T}"], rtol=0, atol=3e-7)
        feat = ds.make_visual_features("vid150", 0, 5)
        assert feat.shape == (6, 8194) and feat.dtype == torch.float32
        np.testing.assert_allclose(feat[:, -2:].numpy(), [[t / 6, (t + 1) / 6] for t in range(6)], rtol=1e-6)
        bank = ds.feature_bank()
        assert bank.seg.shape == (6 + 6 + 5 + 5, 4096) and bank.clip_off.tolist() == [0, 6, 12, 17, 22]


@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo"), ("n21", 21)])
def test_evaluators_on_cpu_device_match_reference_dicts(golden, tag, clips):
    g = golden(f"g2_scoring_{tag}.npz")
    p = problem(100, 50, clips)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    model = make_model(p["sd"])
    vi, li = ds.iterators()
    got = vevaluate.evaluate(model, vi, li, ds.annotations, "cpu")
    ref = json.loads(str(g["corpus_metrics"]))
    assert set(got) == set(ref)
    for key in ref:
        assert got[key] == pytest.approx(ref[key], abs=1e-9), key
    prior = {int(k): [tuple(m) for m in v] for k, v in json.loads(str(g["prior"])).items()}
    random.seed(123)
    vi, li = ds.iterators()
    got = vsingle.evaluate(model, vi, li, ds.annotations, "cpu", model_types=["model", "chance", "prior"], prior=prior)
    ref = json.loads(str(g["single_metrics"]))
    assert set(got) == set(ref)
    for key in ref:
        assert got[key] == pytest.approx(ref[key], abs=1e-9), key


def test_cal_model_state_dict_surface():
    sd = problem(2, 2, 6)["sd"]
    m = make_model(sd)
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert keys["visual_fc.0.weight"] == (500, 8194) and keys["visual_fc.2.weight"] == (100, 500)
    assert keys["lstm.weight_hh_l0_reverse"] == (4000, 1000) and keys["lang_fc.weight"] == (100, 2000)
    h = m.init_hidden(3, "cpu")
    assert len(h) == 2 and h[0].shape == (2, 3, 1000)
    m.train()
    x = torch.randn(4, 8194, requires_grad=True)
    m(x).sum().backward()                                         # autograd flows in training mode (main.py:66)
    assert x.grad is not None


def test_no_positive_moment_raises_like_reference():
    p = problem(4, 3, 6)
    times = [[[0, 0], [1, 1], [2, 2], [3, 3]]] * 3                # no two annotators agree -> no positive
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], times)
    vi, li = ds.iterators()
    with pytest.raises(IndexError):
        vevaluate.evaluate(make_model(p["sd"]), vi, li, ds.annotations, "cpu")


def _check_validate(got, ref):
    for group in ("CustomRecall", "MedianRank", "MeanReciprocalRank"):
        assert set(got[group]) == set(ref["scalars"][group]), group
        for key, want in ref["scalars"][group].items():
            assert got[group][key] == pytest.approx(want, abs=1e-12), (group, key)
    assert set(got["pr_curve"]) == set(ref["pr_curve"])
    for kind, per_k in ref["pr_curve"].items():
        for k, want in per_k.items():
            assert got["pr_curve"][kind][int(k)] == pytest.approx(want, abs=1e-12), (kind, k)


@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo")])
@pytest.mark.parametrize("size", [25, -1])
def test_validate_epoch_on_cpu_device_matches_reference_scalars(tag, clips, size):
    """Trainer.validate_epoch (main.py:121-212): `>=` thresholds, 1-based ranks, MRR, size limit, 11-point PR sweep."""
    ref = json.load(open(Path(__file__).parent / "golden" / "g6_validate_epoch.json"))[f"{tag}_size{size}"]
    p = problem(60, 40, clips, seed=77)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    vi, li = ds.iterators()
    got = vevaluate.validate_epoch(make_model(p["sd"]), vi, li, ds.annotations, "cpu", size=size)
    _check_validate(got, ref)


@pytest.mark.parametrize("tag,nl", [("plain", False), ("normalized", True)])
def test_ranking_loss_cpu_device_matches_reference(golden, tag, nl):
    """losses.ranking_loss on CPU tensors: the reference's value, n_samples and autograd gradients."""
    from vfr_amd import losses, synth
    g = golden("g7_ranking_loss.npz")
    posit, intra, inter, lang, maskp, maskn = synth.ranking_batch(41)
    t = [torch.from_numpy(a).clone().requires_grad_(True) for a in (posit, intra, inter, lang)]
    loss, n = losses.ranking_loss(*t, torch.from_numpy(maskp), torch.from_numpy(maskn), normalize_loss=nl)
    loss.backward()
    assert n == int(g[f"n_{tag}"]) and loss.item() == pytest.approx(float(g[f"loss_{tag}"]), rel=1e-6)
    for name, x in zip(("posit", "intra", "inter", "lang"), t):
        np.testing.assert_allclose(x.grad.numpy(), g[f"grad_{name}_{tag}"], rtol=1e-5, atol=1e-8)


def _npy_corpus(td, lengths=(150, 138, 125, 112)):
    d = Path(td) / "features_vgg19"
    d.mkdir()
    names = []
    for T in lengths:
        x = np.random.RandomState(1000 + T).rand(T, 4096).astype(np.float32)
        x[x < 0.3] = 0.0
        np.save(d / f"vgg19_ft_vid{T}.npy", x)
        names.append(f"vid{T}")
    return names


@pytest.mark.parametrize("mode", ["avg", "max"])
def test_feature_store_round_trip_and_dataset_equivalence(golden, mode):
    """store.FeatureStore: packing the extractor's .npy files gives the reference's pooled rows (g4),
    and a dataset opened over the packed file is indistinguishable from one that pooled the .npy files itself."""
    from vfr_amd.store import FeatureStore
    g = golden("g4_pooling.npz")
    with tempfile.TemporaryDirectory() as td:
        names = _npy_corpus(td)
        plain = vdata.CustomDataset(names, {}, td, "vgg19", pooling=mode, pool_device="cpu")
        st = FeatureStore.from_npy(Path(td) / "features_vgg19.vfs", names, td, "vgg19", mode, pool_device="cpu")
        assert (st.Nv, st.C, st.F) == (4, 22, 4096) and st.videos == names
        assert st.clip_off.tolist() == [0, 6, 12, 17, 22] and st.num_segments_info == plain.num_segments_info
        for T, name in zip((150, 138, 125, 112), names):
            seg_v, ctx_v = st.video_rows(name)
            np.testing.assert_allclose(seg_v.numpy(), g[f"seg_{mode}_{T}"], rtol=0, atol=3e-7)
            np.testing.assert_allclose(ctx_v.numpy(), g[f"ctx_{mode}_{T}"], rtol=0, atol=3e-7)
        packed = vdata.CustomDataset(names, {}, td, "vgg19", pooling=mode, pool_device="cpu")     # finds the .vfs
        assert packed.store is not None and packed.num_segments_info == plain.num_segments_info
        for name in names:
            n = plain.num_segments_info[name]
            assert torch.equal(packed.make_visual_features(name, 0, n - 1), plain.make_visual_features(name, 0, n - 1))
        a, b = packed.feature_bank(), plain.feature_bank()
        assert torch.equal(a.seg, b.seg) and torch.equal(a.ctx, b.ctx) and torch.equal(a.clip_off, b.clip_off)
        assert a.seg.data_ptr() == st.feature_bank().seg.data_ptr() or a.seg.untyped_storage().size() >= 4 * 22 * 4096
        sub = [names[2], names[0]]                                                              # another order: gathered
        a, b = packed.feature_bank(sub), plain.feature_bank(sub)
        assert a.videos == sub and torch.equal(a.seg, b.seg) and torch.equal(a.ctx, b.ctx) and a.clip_off.tolist() == [0, 5, 11]
        with pytest.raises(ValueError):
            vdata.CustomDataset(names, {}, td, "vgg19", pooling="max" if mode == "avg" else "avg", pool_device="cpu")


def test_feature_store_rejects_damaged_files():
    from vfr_amd.store import FeatureStore
    with tempfile.TemporaryDirectory() as td:
        p = Path(td) / "s.vfs"
        rs = np.random.RandomState(0)
        FeatureStore.write(p, ["a", "b"], [0, 2, 5], rs.rand(2, 8), rs.rand(5, 8))
        st = FeatureStore.open(p)
        assert st.seg.shape == (5, 8) and st.counts.tolist() == [2, 3]
        raw = p.read_bytes()
        (Path(td) / "cut.vfs").write_bytes(raw[:len(raw) - 4096])
        (Path(td) / "magic.vfs").write_bytes(b"NOTSTORE" + raw[8:])
        for bad in ("cut.vfs", "magic.vfs"):
            with pytest.raises(ValueError):
                FeatureStore.open(Path(td) / bad)
        with pytest.raises(ValueError):
            FeatureStore.write(p, ["a", "a"], [0, 2, 5], rs.rand(2, 8), rs.rand(5, 8))
        with pytest.raises(ValueError):
            FeatureStore.write(p, ["a", "b"], [0, 2, 4], rs.rand(2, 8), rs.rand(5, 8))
        FeatureStore.write(p, [], [0], np.zeros((0, 8)), np.zeros((0, 8)))                       # empty corpus
        assert FeatureStore.open(p).feature_bank().seg.shape[0] == 0


def test_engine_exchange_helpers_on_cpu():
    """engine.overlapped degrades to two plain calls off the GPU; _all_gather_rows picks its collective from what the group
    object offers (identically on every rank) and lets a failing collective raise -- no per-rank fallback at call time."""
    from vfr_amd import engine
    assert engine.overlapped("cpu", lambda: 1, lambda: 2) == (1, 2)
    assert engine.overlapped("cuda:0", lambda: "a", lambda: "b", enable=False) == ("a", "b")

    class ListOnly:
        calls = 0

        def all_gather(self, parts, t):
            ListOnly.calls += 1
            for g, p in enumerate(parts):
                p.copy_(t + g)

    class IntoTensor(ListOnly):
        def all_gather_into_tensor(self, out, t):
            assert out.shape == (3 * t.shape[0],) + tuple(t.shape[1:])            # concatenated form
            for g in range(3):
                out[g * t.shape[0]:(g + 1) * t.shape[0]].copy_(t + g)

    class Fails(ListOnly):
        def all_gather_into_tensor(self, out, t):
            raise RuntimeError("collective failed on this rank")

    mine = torch.arange(6, dtype=torch.int64).reshape(2, 3)
    for group in (ListOnly(), IntoTensor()):
        out = torch.empty((3, 2, 3), dtype=torch.int64)
        engine._all_gather_rows(group, out, mine)
        assert all(torch.equal(out[g], mine + g) for g in range(3))
    assert ListOnly.calls == 1
    with pytest.raises(RuntimeError, match="collective failed"):
        engine._all_gather_rows(Fails(), torch.empty((3, 2, 3), dtype=torch.int64), mine)
    assert ListOnly.calls == 1                                   # no silent switch to another collective

    class OldGloo(ListOnly):
        """a torch.distributed whose gloo group refuses the tensor form (older builds): every rank must settle on the list
        form ONCE, through a probe whose outcome is reduced over the ranks -- not per call, not per rank"""
        probes = reduces = 0

        class ReduceOp:
            MIN = "min"

        def get_backend(self): return "gloo"
        def get_world_size(self): return 3

        def all_gather_into_tensor(self, out, t):
            OldGloo.probes += 1
            raise RuntimeError("ProcessGroupGloo does not support _allgather_base")

        def all_reduce(self, t, op=None):
            OldGloo.reduces += 1                                 # (single process: the MIN over the ranks is this rank's value)

    old = OldGloo()
    for _ in range(2):
        out = torch.empty((3, 2, 3), dtype=torch.int64)
        engine._all_gather_rows(old, out, mine)
        assert all(torch.equal(out[g], mine + g) for g in range(3))
    assert (OldGloo.probes, OldGloo.reduces) == (1, 1)           # probed once, then remembered


@pytest.mark.parametrize("clips", [6, "didemo", 21])
def test_gt_label_table_vectorised_equals_get_iou_loop(clips):
    """a11 (model/evaluate.py:59-62): the tabulated/gathered table == the reference's per-moment get_iou expression, with a
    ragged number of annotators, both comparison forms (> for evaluate, >= for validate_epoch) and a threshold of 1.0."""
    from vfr_amd import engine, synth
    counts = synth.clip_counts(300, clips, seed=5)
    own, times = synth.annotations(400, counts, seed=5)
    times = [t if i % 3 else t + [t[1], [0, 0]] for i, t in enumerate(times)]          # 4 or 6 annotators
    for strict, thrs in ((True, [0.5, 0.7]), (False, [0.0, 0.3, 0.5, 1.0])):
        lab = engine.gt_label_table(times, counts[own], thrs, strict=strict)
        assert lab.shape[:2] == (len(thrs), 400)
        for q in range(400):
            mom = vutils.generate_moments(int(counts[own[q]]))
            for r, thr in enumerate(thrs):
                iou = [vutils.get_iou(times[q], s, e) for s, e in mom]
                want = [int(((i > thr) if strict else (i >= thr)).sum() >= 2) for i in iou]
                assert lab[r, q, :len(mom)].astype(int).tolist() == want
                assert not lab[r, q, len(mom):].any()
    t, na = engine.pack_times(times)
    assert t.shape == (400, 6, 2) and na.tolist() == [6 if i % 3 == 0 else 4 for i in range(400)]
    assert engine.gt_label_table([], np.zeros(0, int), [0.5]).shape == (1, 0, 0)
    # spans far outside the clip range (a foreign annotation file) take the per-query form instead of a (T - lo)^2 table
    far = [[[0, 1], [0, 1], [5000, 9000], [2, 4000]] for _ in range(5)] + [[[3, 3], [3, 3], [3, 4], [70, 90]]]
    cf = np.array([6, 6, 6, 6, 6, 5])
    lab = engine.gt_label_table(far, cf, [0.5, 0.7])
    for q in range(6):
        mom = vutils.generate_moments(int(cf[q]))
        for r, thr in enumerate([0.5, 0.7]):
            want = [int((vutils.get_iou(far[q], s, e) > thr).sum() >= 2) for s, e in mom]
            assert lab[r, q, :len(mom)].astype(int).tolist() == want


@pytest.mark.parametrize("tag,clips", [("n6", 6), ("ragged", "didemo")])
def test_evaluate_chance_baseline_matches_reference(tag, clips, capsys):
    """evaluate(model_types=['model', 'chance']) under np.random.seed(123) == the reference's dict (fixture G8): one
    permutation per query shared by both IoU thresholds (evaluate.py:68-72); 130 queries also cross the preliminary print."""
    ref = json.load(open(Path(__file__).parent / "golden" / "g8_chance.json"))[tag]
    p = problem(40, 130, clips, feat_dim=256, seed=88)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    model = make_model(p["sd"], feat_dim=256)
    vi, li = ds.iterators()
    np.random.seed(123)
    got = vevaluate.evaluate(model, vi, li, ds.annotations, "cpu", model_types=["model", "chance"])
    assert set(got) == set(ref)
    for key in ref:
        for name in ref[key]:
            assert got[key][name] == pytest.approx(ref[key][name], abs=1e-9), (key, name)
    assert capsys.readouterr().out.count("chance, IoU=0.5") == 1                     # printed once, at query 101 (li = 100)
    vi, li = ds.iterators()
    np.random.seed(123)
    vevaluate.evaluate(model, vi, li, ds.annotations, "cpu", model_types=["model", "chance"], preliminary=129)
    assert capsys.readouterr().out.count("chance, IoU=0.5") == 1                     # len == m * preliminary + 1 still prints


def _encoder_grad_case(tag, device, golden):
    """The reference's loss.backward() case of fixture G10 through OUR CALModel on `device`."""
    from vfr_amd import models as vmodels, synth
    g = golden("g10_encoder_grads.npz")
    nl = tag == "normlang"
    sd = synth.model_weights(16, vocab=60, hidden=24, seed=31, normalize_lang=nl)
    m = vmodels.CALModel(2 * 16 + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]), hidden_size=24,
                         dropout_rate=0.0, normalize_lang=nl)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(device).train()
    rs = np.random.RandomState(32)
    x = torch.from_numpy(rs.rand(37, 34).astype(np.float32)).to(device)
    tok = torch.from_numpy(synth.query_tokens(11, vocab=60, seed=33)).to(device)
    wv = torch.from_numpy(rs.randn(37, 100).astype(np.float32)).to(device)
    wl = torch.from_numpy(rs.randn(11, 100).astype(np.float32)).to(device)
    vis = m(x)
    lang = m(tok, False, device)
    ((vis * wv).sum() + (lang * wl).sum()).backward()
    np.testing.assert_allclose(vis.detach().cpu().numpy(), g[f"{tag}_vis"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(lang.detach().cpu().numpy(), g[f"{tag}_lang"], rtol=0, atol=1e-4)
    names = [k[len(tag) + 6:] for k in g.files if k.startswith(f"{tag}_grad_")]
    assert len(names) >= 14
    params = dict(m.named_parameters())
    for name in names:
        want = g[f"{tag}_grad_{name}"]
        got = params[name].grad.detach().cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * max(1.0, float(np.abs(want).max())), err_msg=name)


@pytest.mark.parametrize("tag", ["plain", "normlang"])
def test_encoder_gradients_cpu_device_match_reference(golden, tag):
    """f2 second half, CPU device: loss.backward() through CALModel (torch.nn path) == the reference's gradients (G10)."""
    _encoder_grad_case(tag, "cpu", golden)


def test_frame_sampling_matches_reference_arithmetic(oracle):
    """f4: get_rgb_features.py:45-60 -- the vectorised index selection == the reference's running-step loop (oracle
    restatement) over frame counts / frame rates / segment counts, and the counts SURVEY 3.1 derived from it."""
    from vfr_amd import features
    assert len(features.sample_frames(900, 30.0, 6)) == 150 and len(features.sample_frames(750, 30.0, 5)) == 125
    assert len(features.sample_frames(819, 30.0, 6)) == 138                      # 27.3 s: the last clip pools 13 frames
    assert features.sample_frames(0, 30.0, 6).tolist() == []
    rs = np.random.RandomState(0)
    for _ in range(1500):
        fps = float(rs.choice([23.976, 24.0, 25.0, 29.97, 30.0, 15.0, 59.94, 12.5]))
        nseg = int(rs.randint(1, 7))
        nf = int(rs.randint(1, int(fps * 5 * nseg) + 2))
        got = features.sample_frames(nf, fps, nseg)
        assert got.tolist() == oracle.frame_sample_indices(nf, fps, nseg), (nf, fps, nseg)
        assert got.dtype == np.int64 and (np.diff(got) >= 0).all() and got.max() < max(nf, 1)


def g11_norm_input():
    """The seeded uint8 frames the G11 normalisation case was generated from (tools/gen_golden.py::g11_frame_front_end)."""
    fr = synth.frames_u8(150, 16, 16, seed=11)
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    fr[0, :, :, 0], fr[0, :, :, 1], fr[0, :, :, 2] = ramp, ramp[::-1], ramp.T
    return fr


def test_frame_front_end_matches_reference_getitem(golden, oracle):
    """a1 + f4 pinned to the reference ITSELF (fixture G11 = the unmodified DiDeMoDataset.__getitem__,
    get_rgb_features.py:37-78, fed by a stubbed read_video): selected frame indices for 28 (frames, fps, segments) cases --
    full clips, short last segments, one-segment clips, one frame, an empty read -- from both the product's vectorised
    selection and the oracle's restatement; the oracle's normalisation == the reference's tensor bit for bit."""
    from vfr_amd import features
    g = golden("g11_frame_front_end.npz")
    cases = g["cases"]
    assert len(cases) >= 20
    for i, (nf, fps, nseg) in enumerate(cases):
        want = g[f"idx_{i}"].tolist()
        assert features.sample_frames(int(nf), float(fps), int(nseg)).tolist() == want, (i, nf, fps, nseg)
        assert list(oracle.frame_sample_indices(int(nf), float(fps), int(nseg))) == want, (i, nf, fps, nseg)
    assert [len(g[f"idx_{i}"]) for i in range(3)] == [150, 138, 125]            # SURVEY 3.1's counts
    fr = g11_norm_input()
    idx = features.sample_frames(150, 30.0, 1)
    want = g["norm_frames"]
    got = oracle.frames_normalize(fr[idx])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_extract_dataset_resume_and_missed_bookkeeping(tmp_path, monkeypatch):
    """f4: the extraction loop's skip-done / skip-missed / record-unreadable behaviour (get_rgb_features.py:105-116,152-156)
    with the device pass stubbed out (this is the host logic)."""
    from vfr_amd import features
    calls = []

    def fake_extract(frames, fps, nseg, weights, cfg=None, model_type="vgg19"):
        calls.append((len(frames), fps, nseg))
        return torch.full((len(features.sample_frames(len(frames), fps, nseg)), 4), float(nseg))
    monkeypatch.setattr(features, "extract_video", fake_extract)
    info = [dict(video=f"v{i}", num_segments=6 if i % 2 else 5) for i in range(5)]
    ft = tmp_path / "features"
    ft.mkdir()
    np.save(ft / "vgg19_ft_v1", np.zeros((3, 4), np.float32))                   # already done
    missed_file = tmp_path / "missed_videos_features.json"
    missed_file.write_text(json.dumps(["v2"]))                                  # known unreadable
    decoder = lambda video, nseg: (None, 0) if video == "v3" else (np.zeros((25 * 5 * nseg, 2, 2, 3), np.uint8), 25.0)
    written, missed = features.extract_dataset(info, decoder, ft, None, missed_path=missed_file)
    assert written == ["v0", "v4"] and missed == ["v2", "v3"] and json.loads(missed_file.read_text()) == ["v2", "v3"]
    assert np.load(ft / "vgg19_ft_v0.npy").shape == (125, 4) and np.load(ft / "vgg19_ft_v1.npy").shape == (3, 4)
    assert calls == [(625, 25.0, 5), (625, 25.0, 5)]
    # the resnet152 variant (get_rgb_features.py:127-131) keeps its own file prefix: nothing of the vgg19 run counts as done
    calls.clear()
    written, _ = features.extract_dataset(info, decoder, ft, None, model_type="resnet152", missed_path=missed_file)
    assert written == ["v0", "v1", "v4"] and (ft / "resnet152_ft_v1.npy").exists() and len(calls) == 3
    assert np.load(ft / "resnet152_ft_v0.npy").shape == (125, 4, 1, 1)           # the reference's [T, C, 1, 1] (get_rgb_features.py:129-131,151)
    with pytest.raises(ValueError):
        features.extract_dataset(info, decoder, ft, None, model_type="resnet50", missed_path=missed_file)


def test_resnet_packed_is_a_plain_pair_with_a_fold_cache():
    """`_vfr.resnet_pack` returns a tuple subclass: callers (features.extract_video, the tests) unpack it as (convs, bns); the folded
    weights `resnet_pool` computes once per (blocks, width, eps, device) live in its `.folded` dict."""
    from vfr_amd import _vfr
    p = _vfr.ResnetPacked(([1, 2], [3, 4]))
    convs, bns = p
    assert convs == [1, 2] and bns == [3, 4] and isinstance(p, tuple) and len(p) == 2 and p.folded == {}
    p.folded["k"] = 1
    assert _vfr.ResnetPacked(([], [])).folded == {}            # per instance, not shared


def test_resnet_plan_and_frame_selection_host_side(oracle):
    """f4 host logic: the product's convolution order for vfr_resnet_pool_f32 == the oracle's execution plan (names, count =
    1 + sum(3 n + 1)); features.select_frames gathers on the host, passes a pre-selected clip through (fps <= 0) and keeps dtype."""
    from vfr_amd import _vfr, features
    for blocks in ((3, 8, 36, 3), (1, 2, 2, 1), (2, 1, 3, 2)):
        mine = _vfr.resnet_conv_plan(blocks, 64)
        theirs = oracle.resnet_conv_plan(blocks, 64)
        assert [m for m in mine] == [(t[0], t[1]) for t in theirs]
        assert len(mine) == 1 + sum(3 * n + 1 for n in blocks)
        sd = synth.resnet_weights(blocks, 4, seed=2)
        for conv, bn in mine:
            assert conv + ".weight" in sd and all(f"{bn}.{k}" in sd for k in ("weight", "bias", "running_mean", "running_var"))
    fr = np.arange(900 * 2 * 2 * 3, dtype=np.uint8).reshape(900, 2, 2, 3)
    sel = features.select_frames(fr, 30.0, 6)
    idx = features.sample_frames(900, 30.0, 6)
    assert sel.dtype == torch.uint8 and sel.shape == (150, 2, 2, 3) and np.array_equal(sel.numpy(), fr[idx])
    assert features.select_frames(fr[:7], 0.0, 6).shape == (7, 2, 2, 3)
