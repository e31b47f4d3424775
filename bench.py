#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: query x video scorings per second for one full
evaluation pass (model/evaluate.py:28-90) over a synthetic DiDeMo-shaped corpus.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch, inputs resident in HBM: clip encoder over this rank's
videos (packed [n+1, 4096] fc7 features -> 100-d clip embeddings), BiLSTM query encoder over the query
batch, own-video scores for ground truth, then the fused kernel that scores every query against every
moment of every video, keeps the top-k and counts the rank of the best ground-truth moment for both IoU
thresholds.  With N > 1 the 10k videos are sharded contiguously (strong scaling), each
rank encodes 1/N of the queries (all_gather), best-GT keys are all_reduce(MIN)'d, rank counts
all_reduce(SUM)'d and the per-shard top-k lists all_gather'ed and merged (RCCL).

Rank 0 prints ONE JSON line: the driver contract plus
  "roofline"     -- dominant kernel (largest share of device time), algorithmic FLOP per launch / measured
                    launch duration (HIP events on the launch stream, taken inside the timed steps);
  "cpu_baseline" -- the CPU oracle (a port, not the product) timed on the host cores on a bounded sample
                    of the same pass.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP32_PEAK_TFLOPS = 157.3        # MI355X fp32 vector == fp32 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--videos", type=int, default=10000)
    ap.add_argument("--clips", default="21", help="clips per video: 21 (BASELINE-literal), 6 (reference-native) or 'didemo'")
    ap.add_argument("--queries", type=int, default=5000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--feat-dim", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16 / small-batch / VGG sub-records measured after the timed steps")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="vfr_set_option passthrough for parameter sweeps (results must not change: compare the checksums)")
    ap.add_argument("--host-feed", action="store_true",
                    help="also time the pass with the pooled features in pinned HOST memory (PCIe-inclusive; extra field, never `value`)")
    ap.add_argument("--vgg-videos", type=int, default=24, help="videos of the extractor-loop sub-record (BASELINE.md C3's loop)")
    ap.add_argument("--plant-alpha", type=float, default=0.5, help="noise scale of the planted-query sub-record (realistic_gt)")
    ap.add_argument("--cpu-queries", type=int, default=1024, help="queries of the batch the CPU oracle leg runs (against ALL videos)")
    ap.add_argument("--parity-ranks", type=int, default=64, help="queries whose ground-truth rank counts the oracle recomputes")
    return ap.parse_args()


# profiler sites -> the kernel they launch (the dominant kernel is chosen per KERNEL: the scorer's ladder stages B and C
# are the same kernel under two sites)
KERNEL_OF_SITE = {
    "gemm_lstm_rec": "lstm_step_mfma_pair", "gemm_vis_seg": "gemm_nt_mfma<seg+hidden>", "gemm_vis_ctx": "gemm_nt_mfma<ctx>",
    "gemm_vis_out": "gemm_nt_mfma<out>", "gemm_lang_fc": "gemm_nt_mfma<lang_fc>", "gemm_lstm_in": "gemm_nt_mfma<vocab projection>",
    "score_fused": "score_mfma_kernel", "score_prepass": "score_mfma_kernel", "score_rank": "score_mfma_kernel",
    "score_pairs": "score_pairs_video_kernel", "score_finish": "topk_finish_kernel", "score_prep": "mfma_prep kernels",
    "score_fallback": "score_fast_kernel (masked fallback)", "topk_merge": "topk_merge_tasks_kernel", "score_own": "score_own_kernel",
    "exchange": "exchange / label kernels",
}
SCORER_SITES = ("score_fused", "score_prepass", "score_rank", "score_pairs", "score_finish", "score_prep", "score_fallback", "topk_merge")


def site_work(site, cfg):
    """(algorithmic FLOP per launch -- SURVEY.md 8d figures x the units one launch processes --, FLOP the kernel actually
    executes per launch) of an instrumented site.  They differ for the fused LSTM step: the reference's arithmetic has an
    E-wide input part per (query, time) which the kernel reads from the per-vocabulary projection table instead of
    multiplying, and the first step's recurrent part (h_0 = 0) is skipped."""
    B, C, Nv, H, E, F, hid, D, n, Nq, T, vocab = (cfg[k] for k in ("Bq", "C", "Nv", "H", "E", "F", "hid", "D", "n", "Nq", "T", "vocab"))
    M = n * (n + 1) // 2
    nb = (min(640, Nv // 16) // 2 or min(640, Nv // 16)) if Nv >= 256 else 0   # ladder stage B (score.hip: pre_b_videos, halved beside the candidate histogram)
    rows = cfg["lstm_rows_per_step"]
    per_scoring = 2 * n * D + n + 2 * M
    world = cfg.get("world", 1)
    if world > 1 or cfg.get("sharded_protocol"):
        # sharded pass (engine.sharded_search_fused): the top-k sample pass over s_r videos and the seeded main pass over the rest
        # of the shard are both `score_fused` launches (their work averaged), the sample videos' rank-only pass is `score_rank`
        s_r = min(Nv, -(-256 // world))
        score = {"score_fused": (float(Nq) * Nv * per_scoring / 2,) * 2, "score_rank": (float(Nq) * s_r * per_scoring,) * 2,
                 "score_prepass": (None, None)}
    else:
        score = {"score_fused": (float(Nq) * max(Nv - nb, 0) * per_scoring,) * 2,
                 "score_rank": (float(Nq) * Nv * per_scoring,) * 2,          # k = 0 calls only
                 "score_prepass": (float(Nq) * nb / 2 * per_scoring,) * 2}   # two launches (A, B); B's work averaged over both
    table = {
        # fused step [x_t | h] x [Wih | Whh]^T over the rows processed per launch (forward: every query + the all-pad row;
        # reverse: only the queries that have reached a real token), averaged over the T launches
        "gemm_lstm_rec": (2.0 * rows * 4 * H * (E + H), 2.0 * rows * 4 * H * H * (T - 1) / T),
        # emb x W_ih^T for both directions, once per VOCABULARY entry (not per query)
        "gemm_lstm_in": (2.0 * 2 * vocab * 4 * H * E,) * 2,
        "gemm_vis_seg": (2.0 * C * hid * F,) * 2,
        "gemm_vis_ctx": (2.0 * Nv * hid * F,) * 2,
        "gemm_vis_out": (2.0 * C * D * hid,) * 2,
        "gemm_lang_fc": (2.0 * B * D * 2 * H,) * 2,
        # scoring launches (2nD contraction + n norms + 2M moment means per scoring): stage B of the threshold
        # ladder = the first Nv/16 (<= 640) videos, the main launch the rest (stage A's 64-video sample is scored again by B)
    }
    table.update(score)
    return table.get(site, (None, None))


def host_cores():
    """(threads the CPU legs use, cores the host reports).  A one-GPU lease on this pool owns 16 of the host's cores (the
    affinity mask says so when it is narrower than cpu_count); the oracle's OpenMP team is sized to that share."""
    total = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = total
    return min(avail, int(os.environ.get("VFR_BENCH_CPU_THREADS", "16"))), total


def cpu_baseline_and_parity(args, n_clips, seg, ctx, clip_off, tokens, sd, own, times, counts_all, gpu_out, gpu_emb, gpu_Q):
    """The CPU leg, on the BENCH CORPUS ITSELF (same seeded inputs, copied back from HBM): the C oracle's full pass -- clip MLP
    over every video, BiLSTM over the first `cpu-queries` queries, scoring + top-k of those queries against all videos --
    timed on the host cores = `cpu_baseline`; then, untimed, the metric's second half: the GPU step's outputs for those
    queries against the oracle's (`parity`: clip / query embeddings, top-1 / top-5 / top-k moment ids and distances, and the
    rank of the best ground-truth moment at both IoU thresholds for the first `parity-ranks` queries)."""
    from oracle import oracle as orc
    threads, total = host_cores()
    orc.set_threads(threads)
    nq = min(args.cpu_queries, tokens.shape[0])
    seg_h, ctx_h, off_h = seg.cpu().numpy(), ctx.cpu().numpy(), clip_off.cpu().numpy()
    tok_h = tokens[:nq].cpu().numpy()
    nv = len(off_h) - 1
    lstm = {k[5:]: v for k, v in sd.items() if k.startswith("lstm.")}
    t0 = time.perf_counter()
    V = orc.visual_mlp(seg_h, ctx_h, off_h, sd["visual_fc.0.weight"], sd["visual_fc.0.bias"], sd["visual_fc.2.weight"], sd["visual_fc.2.bias"])
    t1 = time.perf_counter()
    Q = orc.bilstm_final(tok_h, sd["word_embedding.weight"], lstm, sd["lang_fc.weight"], sd["lang_fc.bias"])
    t2 = time.perf_counter()
    od, oi = orc.score_topk(Q, V, off_h, args.k)
    t3 = time.perf_counter()
    dt = t3 - t0
    del seg_h, ctx_h
    base = {"value": nq * nv / dt, "unit": "scorings/s", "cores": threads, "cores_of_host": f"{threads} of {total}", "kind": "port",
            "sample": f"the bench corpus itself: first {nq} of the {tokens.shape[0]} queries x all {nv} videos ({n_clips} clips x "
                      f"{args.feat_dim}-d), full pass in {dt:.1f} s with the C oracle (clip MLP {t1 - t0:.1f} s + BiLSTM {t2 - t1:.1f} s + "
                      f"score/top-{args.k} {t3 - t2:.1f} s), OpenMP x{threads} of the host's {total} cores (a one-GPU lease's share)"}
    # ---- parity of the timed steps' outputs (model/evaluate.py:49-80 restated by the oracle), same inputs ----
    ranks_g, dist_g, idx_g = (t[..., :nq].cpu().numpy() if i == 0 else t[:nq].cpu().numpy() for i, t in enumerate(gpu_out))
    pr = min(args.parity_ranks, nq)
    mo = orc.moment_offsets(off_h)
    own_h = np.asarray(own[:pr], np.int32)
    nmax = int(np.max(np.diff(off_h)))
    own_scores = orc.score_own(Q[:pr], V, off_h, own_h, nmax * (nmax + 1) // 2)
    rank_same, rank_total = 0, 0
    for t, thr in enumerate((0.5, 0.7)):
        dstar, istar = np.empty(pr, np.float32), np.empty(pr, np.int64)
        for q in range(pr):
            n = int(counts_all[own_h[q]])
            lab = orc.gt_labels(times[q], n, thr)
            sc = np.where(lab == 1, own_scores[q, :len(lab)], np.float32(np.inf))
            m = int(np.argmin(sc))                                  # first minimum = smallest moment id among ties
            dstar[q], istar[q] = sc[m], mo[own_h[q]] + m
        oc = orc.rank_of(Q[:pr], V, off_h, dstar, istar)
        rank_same += int((oc == ranks_g[t, :pr]).sum()); rank_total += pr
    parity = {"against": "the C oracle (CPU restatement of model/evaluate.py:49-80, pinned to the reference by tests/golden) on the same inputs",
              "queries": nq, "clip_embedding_rows_identical": f"{int((gpu_emb.cpu().numpy() == V).all(axis=1).sum())}/{V.shape[0]}",
              "query_embedding_rows_identical": f"{int((gpu_Q[:nq].cpu().numpy() == Q).all(axis=1).sum())}/{nq}",
              "top1_ids_identical": f"{int((idx_g[:, 0] == oi[:, 0]).sum())}/{nq}",
              "top5_ids_identical": f"{int((idx_g[:, :5] == oi[:, :5]).all(axis=1).sum())}/{nq}",
              f"top{args.k}_ids_identical": f"{int((idx_g == oi).all(axis=1).sum())}/{nq}",
              f"top{args.k}_distances_identical": f"{int((dist_g == od).all(axis=1).sum())}/{nq}",
              "gt_rank_counts_identical": f"{rank_same}/{rank_total} ({pr} queries x IoU 0.5, 0.7)"}
    parity["all_identical"] = all(v.split(" ")[0].split("/")[0] == v.split(" ")[0].split("/")[1]
                                  for k, v in parity.items() if k.endswith("identical"))
    return base, parity


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    import vfr_amd  # noqa: F401
    from vfr_amd import _vfr, engine, models, synth
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the HIP path has no CPU substitute)"
    for item in args.opt:
        name, value = item.split("=")
        _vfr.set_option(name, int(value))
    # rehearsal switches (not used by the driver): VFR_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # VFR_BENCH_BACKEND=gloo swaps RCCL for gloo, so the multi-rank code path can be exercised on a one-GPU box
    if os.environ.get("VFR_BENCH_SAME_DEVICE"):
        local = 0
    backend = os.environ.get("VFR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # VFR_BENCH_FORCE_DIST=1 (rehearsal, one rank): a world-size-1 RCCL group runs the complete sharded protocol -- the sample
    # pass, the three packed all-gathers, the merges -- so the multi-GPU code path executes on RCCL on a one-GPU box; the
    # checksums must equal the plain run's
    force_dist = world == 1 and bool(os.environ.get("VFR_BENCH_FORCE_DIST"))
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        engine.FORCE_COLLECTIVES = True
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist = None

    n_clips = args.clips if args.clips == "didemo" else int(args.clips)
    Nv, Nq, F = args.videos, args.queries, args.feat_dim
    counts_all = synth.clip_counts(Nv, n_clips, seed=123)
    off_all = np.concatenate([[0], np.cumsum(counts_all.astype(np.int64))])
    mom_all = np.concatenate([[0], np.cumsum(counts_all.astype(np.int64) * (counts_all + 1) // 2)])
    lo, hi = engine.shard_range(Nv, rank, world)
    C_loc = int(off_all[hi] - off_all[lo])

    # ---- synthetic inputs, generated on the device (seeded), resident in HBM before timing ------------
    # every rank draws the SAME corpus (one seeded stream, 3.4 GB transient) and keeps its shard's rows, so the results --
    # and the checksums in the JSON line -- are those of the single-GPU run whatever N is
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    raw = torch.rand((int(off_all[-1]), F), generator=gen, device=dev)[int(off_all[lo]):int(off_all[hi])].clone()
    seg = raw / (raw.norm(dim=1, keepdim=True) + 1e-5)
    clip_off = torch.from_numpy((off_all[lo:hi + 1] - off_all[lo]).astype(np.int32)).to(dev)
    nloc = (clip_off[1:] - clip_off[:-1]).long()
    # per-video mean of the raw rows, deterministic (index_add_ would sum with atomics: run-to-run different bits)
    ctx = torch.segment_reduce(raw, "sum", lengths=nloc, axis=0) / nloc[:, None].float()
    ctx = ctx / (ctx.norm(dim=1, keepdim=True) + 1e-5)
    del raw
    tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
    own, times = synth.annotations(Nq, counts_all, seed=123)
    sd = synth.model_weights(F, seed=123)
    model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    ops = engine.HipOps()
    # a11 inputs resident on the device: the annotators' spans of every query, the clip count of its own video
    t_h, na_h = engine.pack_times(times)
    times_dev = (torch.from_numpy(t_h).to(dev), torch.from_numpy(na_h).to(dev))
    n_own_dev = torch.from_numpy(counts_all[own].astype(np.int32)).to(dev)
    nmax_own = int(counts_all[own].max())
    Mmax_own = nmax_own * (nmax_own + 1) // 2
    IOU = [0.5, 0.7]

    def make_shard(emb):
        bank = _vfr.VideoBank(emb, clip_off, int(mom_all[lo]), max_clips=int(counts_all[lo:hi].max()),
                              total_moments=int(mom_all[hi] - mom_all[lo]), min_clips=int(counts_all[lo:hi].min()))
        return engine.CorpusShard(bank, lo, hi, counts_all, mom_all, dev)

    with torch.no_grad():
        gt_idx = engine.gt_index(make_shard(model.encode_clips(seg, ctx, clip_off)), own)    # which queries this shard owns
    ws = _vfr.topk_workspace(Nq, hi - lo, args.k, dev, total_clips=C_loc)

    def ranks_of(shard, Q):
        # a11 inside the step: the label table of the batch (model/evaluate.py:59-65) is rebuilt from the spans every pass
        labels = ops.gt_labels(times_dev, n_own_dev, IOU, True, dev, Mmax=Mmax_own)
        gt = engine.prepare_gt(shard, own, labels, index=gt_idx)
        return engine.corpus_ranks(shard, Q, own, labels, ops, k=args.k, world=world, workspace=ws, gt=gt)

    def subset(nq):                                   # the same pass for the first nq queries (small-batch sub-records)
        idx = engine.gt_index(make_shard(torch.empty((C_loc, 100), device=dev)), own[:nq])
        td = (times_dev[0][:nq].contiguous(), times_dev[1][:nq].contiguous())
        nd = n_own_dev[:nq].contiguous()

        def run(shard, Q):
            labels = ops.gt_labels(td, nd, IOU, True, dev, Mmax=Mmax_own)
            gt = engine.prepare_gt(shard, own[:nq], labels, index=idx)
            return engine.corpus_ranks(shard, Q, own[:nq], labels, ops, k=args.k, world=1, workspace=ws, gt=gt)
        return run
    ranks_of.own, ranks_of.subset, ranks_of.ws = own, subset, ws

    state = {}

    def step():
        with torch.no_grad():
            # a rank's two encoders are small enough from N = 4 on to leave CUs idle in their partial tile rounds: run them side
            # by side there (tools/rank_sim.py: -0.16 ms at 8, -0.09 at 4, +0.08 at 2); below they stay back to back (the per-kernel durations quoted in `roofline` are then undisturbed)
            emb, Q = engine.overlapped(dev, lambda: model.encode_clips(seg, ctx, clip_off),
                                       lambda: engine.encode_queries(model, tokens, dev, ops, rank, world),
                                       enable=world >= 4 or bool(os.environ.get("VFR_BENCH_OVERLAP")))   # (env: rehearsal switch)
            state["shard"], state["Q"] = make_shard(emb), Q
            return ranks_of(state["shard"], Q)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    _vfr.set_option("profile", 0 if os.environ.get("VFR_BENCH_NO_SITES") else 1)   # rehearsal switch: cost of the site events
    _vfr.profile_read(reset=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    _vfr.set_option("profile", 0)
    sites = _vfr.profile_read(reset=True)
    engine.FORCE_COLLECTIVES = False                 # the sub-records after the timed region are plain single-GPU passes
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    host_feed = None
    if args.host_feed and world == 1:
        # PCIe-inclusive variant: rows start in page-locked host memory (store.FeatureStore.pin() layout); chunked H2D on a
        # side stream overlaps the clip encoder, the query encoder runs on a third stream in the copy's shadow.
        seg_h, ctx_h = seg.cpu().pin_memory(), ctx.cpu().pin_memory()
        off_h = (off_all[lo:hi + 1] - off_all[lo])
        qs = torch.cuda.Stream(dev)

        def step_host():
            with torch.no_grad():
                qs.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(qs):
                    Q = engine.encode_queries(model, tokens, dev, ops, rank, world)
                shard = make_shard(engine.encode_clips_streamed(ops, model, seg_h, ctx_h, off_h, dev))
                torch.cuda.current_stream(dev).wait_stream(qs)
                return ranks_of(shard, Q)
        out_h = step_host()
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(args.steps):
            out_h = step_host()
        torch.cuda.synchronize()
        th = (time.perf_counter() - th) / args.steps
        gbytes = (seg_h.numel() + ctx_h.numel()) * 4 / 1e9
        host_feed = {"value": Nq * Nv / th, "unit": "scorings/s", "ms_per_step": th * 1e3, "h2d_GB_per_step": gbytes,
                     "h2d_floor_ms_at_57.6GBps": gbytes / 57.6 * 1e3,
                     "identical_to_resident": bool(torch.equal(out_h[0], out[0]) and torch.equal(out_h[2], out[2]))}
        del seg_h, ctx_h
    ranks = out[0]
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_step = dt * 1e3 / args.steps
    value = Nq * Nv / (dt / args.steps)
    n_eff = float(np.mean(counts_all))
    Bq = -(-Nq // world)
    tok0 = synth.query_tokens(Nq, seed=123)[:Bq]                       # rank 0's query slice (host copy of the tokens)
    T_ = tok0.shape[1]
    qlen = np.where(tok0 != 0, np.arange(1, T_ + 1)[None, :], 0).max(axis=1)
    rev_rows = sum(1 + int((qlen > T_ - 1 - s).sum()) for s in range(T_)) / T_
    cfg = dict(Bq=Bq, C=C_loc, Nv=hi - lo, H=model.hidden_size, E=100, F=F, hid=500, D=100,
               n=int(round(n_eff)), Nq=Nq, T=T_, lstm_rows_per_step=(Bq + 1) + rev_rows,
               vocab=int(sd["word_embedding.weight"].shape[0]), world=world, sharded_protocol=force_dist)
    if not sites:                                  # VFR_BENCH_NO_SITES rehearsal: nothing to attribute
        print(json.dumps({"ms_per_step": ms_step, "value": value, "n_gpus": world, "note": "site events off (rehearsal)"}))
        if dist is not None:
            dist.destroy_process_group()
        return
    kernels, per_kernel, accounting_notes = {}, {}, []
    for name, (ms, cnt) in sites.items():
        fl, fl_exec = site_work(name, cfg)
        tf = (fl * cnt / (ms * 1e-3) / 1e12) if fl else None
        tf_exec = (fl_exec * cnt / (ms * 1e-3) / 1e12) if fl_exec else None
        # no site may claim more than the chip can do: a violation is an accounting bug, not a result.  Single-GPU (the
        # validated formulas) it stops the bench; a sharded run drops the figure and says so rather than lose the measurement
        if tf is not None and tf > FP32_PEAK_TFLOPS:
            assert world > 1 or force_dist, f"site {name}: {tf:.1f} TFLOP/s > fp32 peak -- wrong work formula"
            accounting_notes.append(f"{name}: work formula gave {tf:.1f} TFLOP/s > peak, figure dropped")
            tf = tf_exec = None
        kernels[name] = {"kernel": KERNEL_OF_SITE.get(name, name), "ms_per_step": ms / args.steps,
                         "launches_per_step": cnt / args.steps, "tflops": tf, "tflops_executed": tf_exec}
        k = per_kernel.setdefault(KERNEL_OF_SITE.get(name, name), {"ms": 0.0, "launches": 0, "flop": 0.0, "flop_exec": 0.0, "sites": []})
        k["ms"] += ms; k["launches"] += cnt; k["sites"].append(name)
        k["flop"] += (fl or 0.0) * cnt; k["flop_exec"] += (fl_exec or 0.0) * cnt
    # dominant kernel = largest share of device time, per KERNEL (a kernel launched from two sites counts once)
    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms"])
    pk = per_kernel[dom]
    dom_ms_launch = pk["ms"] / pk["launches"]
    achieved = pk["flop"] / (pk["ms"] * 1e-3) / 1e12 if pk["flop"] else None
    executed = pk["flop_exec"] / (pk["ms"] * 1e-3) / 1e12 if pk["flop_exec"] else None
    # HBM bytes per launch of this kernel: NOT measured in this run -- quoted from the committed PMC passes of the same
    # kernel (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction; profiles/README.md)
    traffic, traffic_src = None, None
    tfile = ROOT / "profiles" / "traffic_latest.json"
    if tfile.exists():
        tj = json.loads(tfile.read_text())
        for site_name in pk["sites"]:
            if tj.get(site_name) is not None:
                traffic, traffic_src = tj[site_name], tj.get("_source", "profiles/traffic_latest.json (committed PMC passes, not this run)")
                break
    roofline = {"kernel": dom, "sites": pk["sites"], "bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": (achieved / FP32_PEAK_TFLOPS) if achieved else None,
                "achieved_executed": executed, "frac_executed": (executed / FP32_PEAK_TFLOPS) if executed else None,
                "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": dom_ms_launch, "launches_per_step": pk["launches"] / args.steps,
                "share_of_step": pk["ms"] / args.steps / ms_step}
    # the scorer as a whole (all its kernels): algorithmic scoring FLOP of the pass / their summed device time
    sc_ms = sum(sites[s_][0] for s_ in SCORER_SITES if s_ in sites) / args.steps
    sc_flop = float(Nq) * (hi - lo) * (2 * cfg["n"] * 100 + cfg["n"] + cfg["n"] * (cfg["n"] + 1))
    scorer = {"mode": _vfr.DEFAULT_SCORE_MODE, "ms_per_step": sc_ms, "tflops": sc_flop / (sc_ms * 1e-3) / 1e12 if sc_ms else None,
              "frac_fp32_peak": sc_flop / (sc_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS if sc_ms else None,
              "sites_ms": {s_: sites[s_][0] / args.steps for s_ in SCORER_SITES if s_ in sites}}
    if world == 1 and _vfr.DEFAULT_SCORE_MODE == "mfma":
        scorer["exact_rescoring"] = _vfr.score_mfma_stats(ws, Nq, state["shard"].bank, args.k)
    line = {
        "metric": "query x video scorings/sec, full evaluate pass (clip MLP + BiLSTM + moment scoring/top-k/rank), DiDeMo-shape",
        "value": value, "unit": "scorings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"full evaluate pass over {Nv} synthetic DiDeMo-shape videos x {args.clips} clips x {F}-d fc7 features, {Nq}-query "
                               f"batch: clip MLP + BiLSTM query encoder + ground-truth labels + every moment of every video scored, "
                               f"top-{args.k} + rank@IoU{{0.5,0.7}}, fp32, videos sharded over {world} GPU(s) "
                               f"(BASELINE.json configs[1]; configs[3] when sharded)",
                   "videos": Nv, "clips": args.clips, "queries": Nq, "k": args.k, "parallelism": f"shard{world}"},
        "gpu_event_ms_per_step": ev0.elapsed_time(ev1) / args.steps,
        "median_rank_check": float(ranks[0].float().median()),
        "ranks_checksum": int(ranks.sum()), "topk_checksum": int(out[2].sum()) if out[2] is not None else None,
        "roofline": roofline, "scorer": scorer, "kernels": kernels,
    }
    if force_dist:
        line["rehearsal"] = ("VFR_BENCH_FORCE_DIST: ONE rank ran the sharded protocol (sample pass, three packed all-gathers, "
                             f"merges) through a world-size-1 '{backend}' process group; not the headline configuration")
    if accounting_notes:
        line["accounting_notes"] = accounting_notes
    if host_feed is not None:
        line["pcie_inclusive"] = host_feed
    if world == 1 and not args.no_extras:
        line.update(extras(args, dev, model, seg, ctx, clip_off, tokens, make_shard, ranks_of, out, Nq, Nv, counts_all, ops))
    if not args.no_cpu_baseline and world == 1:
        # rank 0 at N = 1 only; the CPU leg doubles as the in-run parity check the metric names ("+ rank@1/5 vs CPU ref")
        line["cpu_baseline"], line["parity"] = cpu_baseline_and_parity(args, n_clips, seg, ctx, clip_off, tokens, sd, own, times, counts_all,
                                                                       out, state["shard"].bank.emb, state["Q"])
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def _timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def extras(args, dev, model, seg, ctx, clip_off, tokens, make_shard, ranks_of, out_f32, Nq, Nv, counts_all, ops):
    """Sub-records measured after the timed region (never part of `value`): BASELINE.md C5 (bf16 MFMA scoring, tolerance
    against the fp32 result), the small query batches of C2, and C3's VGG19-fc7 extractor."""
    from vfr_amd import _vfr, engine, synth
    ex = {}
    with torch.no_grad():
        emb = model.encode_clips(seg, ctx, clip_off)
        shard = make_shard(emb)
        Q = engine.encode_queries(model, tokens, dev, ops)
        # ---- bf16 operands, fp32 accumulate: the same step with the scoring in bf16 mode ----
        old = _vfr.DEFAULT_SCORE_MODE
        try:
            _vfr.DEFAULT_SCORE_MODE = "bf16"
            dt_sc, out_b = _timed(lambda: ranks_of(shard, Q), 3)
            _vfr.DEFAULT_SCORE_MODE = old
            dt_f32, _ = _timed(lambda: ranks_of(shard, Q), 3)
        finally:
            _vfr.DEFAULT_SCORE_MODE = old
        r_f, _, i_f = out_f32
        r_b, _, i_b = out_b
        k = i_f.shape[1]
        same10 = (i_f[:, :10, None] == i_b[:, None, :10]).any(-1).float().mean()
        samek = (i_f[:, :, None] == i_b[:, None, :]).any(-1).float().mean()
        total_m = float(int(counts_all.astype(np.int64) @ (counts_all.astype(np.int64) + 1)) // 2)
        ex["bf16"] = {"what": "scoring with bf16 MFMA operands / fp32 accumulate (vfr_score_topk_mfma dtype bf16): rank counts from the "
                              "bf16 distances, top-k = exact re-rank of the k + 28 best bf16 candidates; same encoders (fp32)",
                      "scoring_ms": dt_sc * 1e3, "scoring_ms_f32_mode": dt_f32 * 1e3,
                      "gemm_tflops_over_scoring_time": 2.0 * Nq * emb.shape[0] * 100 / dt_sc / 1e12,
                      "frac_of_bf16_mfma_peak": 2.0 * Nq * emb.shape[0] * 100 / dt_sc / 1e12 / BF16_PEAK_TFLOPS,
                      "bound": "NOT the bf16 matrix pipes: the query x clip GEMM is 2*Nq*clips*100 = "
                               f"{2.0 * Nq * emb.shape[0] * 100 / 1e9:.0f} GFLOP (well under 0.1 ms at the 2.5 PFLOP/s bf16 peak); the pass is bound by the fp32 "
                               "VALU work behind it -- 231 moment sums per (query, video) compared against two rank keys and the "
                               "top-k threshold (the moment triangle), plus candidate merges and the exact re-rank of k + 28 candidates",
                      "scorings_per_s_scoring_only": Nq * Nv / dt_sc,
                      "rank1_agreement": float((i_f[:, 0] == i_b[:, 0]).float().mean()),
                      "top10_overlap": float(same10), f"top{k}_overlap": float(samek),
                      "rank_count_max_abs_diff_over_moments": float((r_f - r_b).abs().max()) / total_m,
                      "R@1_R@10_R@100_f32": [float((r_f[0] < t).float().mean()) for t in (1, 10, 100)],
                      "R@1_R@10_R@100_bf16": [float((r_b[0] < t).float().mean()) for t in (1, 10, 100)],
                      "median_rank_f32_vs_bf16": [float(r_f[0].float().median()), float(r_b[0].float().median())]}
        # ---- small query batches (Nq in {1, 64, 1024}; 8 and 32 added: up to 32 queries the encoder runs as
        # ONE launch and the scoring with lanes = clips / videos): query encoder + labels + scoring against the resident clip
        # bank (what a serving request costs once the corpus is embedded)
        small = {}
        for nq in (1, 8, 32, 64, 1024):
            if nq > Nq:
                continue
            tk = tokens[:nq].contiguous()
            own_s = np.asarray(ranks_of.own[:nq])
            sub = ranks_of.subset(nq)
            dt_q, out_e = _timed(lambda: sub(shard, engine.encode_queries(model, tk, dev, ops)), 5)
            small[str(nq)] = {"ms": dt_q * 1e3, "scorings_per_s": nq * Nv / dt_q}
            if nq <= 64:
                # the same request replayed as a captured HIP graph (engine.GraphedRequest): no per-launch host work
                try:
                    own_all, times_all = synth.annotations(Nq, counts_all, seed=123)
                    gr = engine.GraphedRequest(model, shard, nq, args.k, ops)
                    gr.load(tk.cpu().numpy(), times_all[:nq], own_all[:nq])
                    gr.replay(); torch.cuda.synchronize(); gr.check()
                    dt_g, out_g = _timed(gr.replay, 20)
                    small[str(nq)].update({"graph_replay_ms": dt_g * 1e3,
                                           "graph_identical_to_eager": bool(all(torch.equal(a_, b_) for a_, b_ in zip(out_g, out_e)))})
                    del gr
                except Exception as e:                      # reported, never hidden; the eager figure above stands
                    small[str(nq)]["graph_replay_error"] = f"{type(e).__name__}: {str(e)[:160]}"
        ex["small_batches"] = {"what": "query encoder + labels + fused scoring of Nq queries against the resident clip bank (ms: launched from "
                                       "Python call by call; graph_replay_ms: the same pass captured once as a HIP graph and replayed)", **small}
        # ---- the scorer with RETRIEVABLE ground truth: the bench batch's encoded queries land mid-distribution (random weights:
        # R@100 = 0, the pre-filter's worst case for rank keys); here each query is planted next to the clips of its first
        # annotated span in its own video (centroid + isotropic noise) so the best ground-truth moment sits in the near tail,
        # as with a trained model.  Scoring only (embeddings given); the result is checked against the exact kernels.
        try:
            ex["realistic_gt"] = realistic_gt(args, dev, emb, shard, ranks_of, Nq, counts_all)
        except IndexError as e:
            ex["realistic_gt"] = {"error": str(e)[:200]}
    # ---- VGG19-fc7 extractor: one 150-frame video of 224x224 frames, full-width random weights ----
    try:
        cfgv = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
        g = torch.Generator(device=dev); g.manual_seed(7)
        cw, cb, cin = [], [], 3
        for c in cfgv:
            if c == "M":
                continue
            cw.append(torch.randn((c, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5)
            cb.append(torch.zeros(c, device=dev))
            cin = c
        fc6 = (torch.randn((4096, 512 * 49), device=dev, generator=g) * 0.01, torch.zeros(4096, device=dev))
        fc7 = (torch.randn((4096, 4096), device=dev, generator=g) * 0.01, torch.zeros(4096, device=dev))
        frames = torch.randint(0, 256, (150, 224, 224, 3), device=dev, dtype=torch.uint8, generator=g)
        dt_v, feat = _timed(lambda: _vfr.vgg_fc7(frames, cfgv, cw, cb, fc6, fc7), 2)
        gflop = 39.26 * 150
        ex["vgg"] = {"what": "get_rgb_features.py path: 150 uint8 frames 224x224x3 -> VGG19 fc7 [150, 4096], fp32, random full-width weights",
                     "ms_per_video": dt_v * 1e3, "frames_per_s": 150 / dt_v, "videos_per_s": 1 / dt_v,
                     "tflops": gflop / dt_v / 1e3, "frac_fp32_mfma_peak": gflop / dt_v / 1e3 / FP32_PEAK_TFLOPS,
                     "finite": bool(torch.isfinite(feat).all())}
        # the LOOP of get_rgb_features.py:134-153 (BASELINE.md C3 is "1k videos", not one): features.extract_dataset over
        # `--vgg-videos` synthetic videos of 900 decoded frames at 30 fps (-> 150 selected), decoder = a seeded frame store in host
        # memory (decode speed is the codec's, not ours), one .npy per video written to a scratch directory
        import tempfile
        from vfr_amd import features
        nvid = args.vgg_videos
        host_videos = [frames.cpu().numpy()[torch.randperm(150, generator=torch.Generator().manual_seed(i)).numpy()][np.repeat(np.arange(150), 6)]
                       for i in range(2)]                                   # two distinct 900-frame clips, reused round-robin
        info = [dict(video=f"v{i:04d}", num_segments=6) for i in range(nvid)]
        decoder = lambda video, nseg: (host_videos[int(video[1:]) % 2], 30.0)
        weights = (cw, cb, fc6, fc7)
        loop = {}
        for tag, pipe in (("pipelined", True), ("serial", False)):
            with tempfile.TemporaryDirectory() as td:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                written, _ = features.extract_dataset(info, decoder, Path(td) / "features_vgg19", weights, missed_path=Path(td) / "missed.json",
                                                      pipeline=pipe)
                torch.cuda.synchronize()
                dt_l = time.perf_counter() - t0
                nfr = sum(np.load(Path(td) / "features_vgg19" / f"vgg19_ft_{v}.npy", mmap_mode="r").shape[0] for v in written)
            loop[tag] = {"videos": len(written), "frames": int(nfr), "s": dt_l, "frames_per_s": nfr / dt_l, "videos_per_s": len(written) / dt_l}
        ex["vgg"]["loop"] = {"what": f"features.extract_dataset over {nvid} videos x 900 decoded frames (150 kept): decode stub -> pinned H2D -> frame "
                                     "selection + VGG19-fc7 -> D2H -> np.save per video; pipelined = reader / device / writer overlapped",
                             **loop, "loop_frames_per_s": loop["pipelined"]["frames_per_s"],
                             "frac_of_kernel_only_rate": loop["pipelined"]["frames_per_s"] / (150 / dt_v)}
    except RuntimeError as e:                                   # e.g. out of memory on a shared device: report, do not hide
        ex["vgg"] = {"error": str(e)[:200]}
    ex["cpu_baseline_loop"] = cpu_loop_baseline(emb, clip_off, Q, counts_all)
    ex["didemo_shape"] = didemo_shape(args)
    return ex


def didemo_shape(args):
    """The same batch at the REAL dataset's clip counts (a DiDeMo video has 5 or 6 clips -- 21 is the number of
    moments of a 6-clip video): n = 6 and the 86 % / 14 % six- / five-clip mix of didemo_video_info.json.  Each shape is this
    script run again as a CHILD process (fresh corpus of that shape, 2 warm-up + 5 timed steps, no CPU leg, no sub-records)
    after the parent's timed region; the child's line is condensed here."""
    import subprocess
    rec = {"what": "bench.py --clips 6 / --clips didemo (same videos x queries x k, 2 + 5 steps) as child processes: the real dataset's shapes"}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "VFR_BENCH_FORCE_DIST")}
    for tag, clips in (("n6", "6"), ("ragged_86_14", "didemo")):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "5", "--warmup", "2", "--clips", clips, "--videos", str(args.videos),
               "--queries", str(args.queries), "--k", str(args.k), "--feat-dim", str(args.feat_dim), "--no-cpu-baseline", "--no-extras"]
        cmd += [x for item in args.opt for x in ("--opt", item)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            rf, ks = line["roofline"], line["kernels"]
            top = sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"])[:4]
            rec[tag] = {"ms_per_step": line["ms_per_step"], "scorings_per_s": line["value"],
                        "dominant_kernel": rf["kernel"], "dominant_share_of_step": rf["share_of_step"],
                        "dominant_frac": rf["frac"], "dominant_frac_executed": rf["frac_executed"],
                        "scorer_ms": line["scorer"]["ms_per_step"], "scorer_frac_fp32_peak": line["scorer"]["frac_fp32_peak"],
                        "top_sites_ms": {k_: round(v["ms_per_step"], 4) for k_, v in top},
                        "ranks_checksum": line["ranks_checksum"], "topk_checksum": line["topk_checksum"]}
        except Exception as e:                                  # a failed child is reported, never hidden
            rec[tag] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return rec


def realistic_gt(args, dev, emb, shard, ranks_of, Nq, counts_all):
    from vfr_amd import _vfr, synth
    own, times = synth.annotations(Nq, counts_all, seed=123)
    off = np.concatenate([[0], np.cumsum(counts_all.astype(np.int64))])
    s0 = np.asarray([t[0][0] for t in times]); e0 = np.asarray([t[0][1] for t in times])
    first = torch.from_numpy(off[own] + s0).to(dev); last = torch.from_numpy(off[own] + e0).to(dev)
    csum = torch.cat([torch.zeros((1, emb.shape[1]), device=dev, dtype=torch.float64), emb.double().cumsum(0)])
    centre = ((csum[last + 1] - csum[first]) / (last + 1 - first)[:, None].double()).float()
    g = torch.Generator(device=dev); g.manual_seed(4321)
    a, b = torch.randint(0, emb.shape[0], (2, 8192), device=dev, generator=g)
    d_typ = float((emb[a] - emb[b]).norm(dim=1).median())                     # typical clip-to-clip distance of the corpus
    sigma = args.plant_alpha * d_typ / emb.shape[1] ** 0.5
    Qp = (centre + sigma * torch.randn(centre.shape, device=dev, generator=g)).contiguous()
    rec = {"what": "scoring pass with planted queries: query embedding = centroid of the clips of its first annotated span + N(0, sigma^2 I), "
                   f"sigma = {args.plant_alpha} x (median clip-to-clip distance {d_typ:.4g}) / sqrt(D); embeddings given, scoring + labels only",
           "plant_alpha": args.plant_alpha}
    old = _vfr.DEFAULT_SCORE_MODE
    res = {}
    for mode in ("mfma", "exact"):
        _vfr.DEFAULT_SCORE_MODE = mode
        try:
            dt, out = _timed(lambda: ranks_of(shard, Qp), 3)
            if mode == "mfma":
                _vfr.set_option("profile", 1); _vfr.profile_read(reset=True)
                ranks_of(shard, Qp); torch.cuda.synchronize()
                _vfr.set_option("profile", 0)
                sites = _vfr.profile_read(reset=True)
                rec["scorer_sites_ms"] = {s_: sites[s_][0] for s_ in SCORER_SITES if s_ in sites}
                rec["scorer_ms"] = sum(rec["scorer_sites_ms"].values())
                rec["exact_rescoring"] = _vfr.score_mfma_stats(ranks_of.ws, Nq, shard.bank, args.k)
        finally:
            _vfr.DEFAULT_SCORE_MODE = old
        res[mode] = (dt, out)
    (dt_m, (r_m, d_m, i_m)), (dt_x, (r_x, d_x, i_x)) = res["mfma"], res["exact"]
    rec.update({"pass_ms_mfma_prefilter": dt_m * 1e3, "pass_ms_exact_kernels": dt_x * 1e3,
                "R@1_R@10_R@100_IoU0.5": [float((r_m[0] < t).float().mean()) for t in (1, 10, 100)],
                "R@1_R@10_R@100_IoU0.7": [float((r_m[1] < t).float().mean()) for t in (1, 10, 100)],
                "median_rank": float(r_m[0].float().median()),
                "identical_to_exact_kernels": bool(torch.equal(r_m, r_x) and torch.equal(i_m, i_x) and torch.equal(d_m, d_x))})
    return rec


def cpu_loop_baseline(emb, clip_off, Q, counts_all, n_queries=10, n_videos=1000):
    """the scoring loop with the reference's own structure (model/evaluate.py:42-80) on the host --
    per query, per video one F.pairwise_distance over the video's clips, per moment index_select -> mean -> .item(), then
    np.argsort over all moments -- on the embeddings of the first n_videos videos, single thread like the reference's loop.
    A stated subsample, extrapolated as a rate.  (`cpu_baseline` above is the multi-threaded C oracle, the stronger baseline.)"""
    import torch.nn.functional as F
    from vfr_amd.utils import generate_moments
    off = clip_off[:n_videos + 1].cpu().numpy()
    V = emb[:int(off[-1])].float().cpu()
    Qc = Q[:n_queries].float().cpu()
    moments = {int(n): [torch.arange(s, e + 1) for s, e in generate_moments(int(n))] for n in set(int(x) for x in counts_all[:n_videos])}
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        for q in range(Qc.shape[0]):
            distances = []
            for v in range(n_videos):
                vis = V[int(off[v]):int(off[v + 1])]
                d = F.pairwise_distance(vis, Qc[q:q + 1].expand_as(vis))
                for idx in moments[vis.shape[0]]:
                    distances.append(d.index_select(0, idx).mean().item())
            np.argsort(np.asarray(distances))
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    return {"value": Qc.shape[0] * n_videos / dt, "unit": "scorings/s", "cores": 1, "kind": "port",
            "sample": f"{Qc.shape[0]} queries x {n_videos} videos of the bench corpus (embeddings given), the reference's loop structure "
                      f"(pairwise_distance per video, index_select/mean/.item() per moment, np.argsort per query) in {dt:.1f} s",
            "host": {"cpu_count": os.cpu_count(), "torch_threads": threads, "torch": torch.__version__}}


if __name__ == "__main__":
    main()
