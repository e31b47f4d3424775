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
thresholds.  With N > 1 the 10k videos are sharded contiguously (strong scaling, BASELINE config 4), each
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
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="vfr_set_option passthrough for parameter sweeps (results must not change: compare the checksums)")
    ap.add_argument("--host-feed", action="store_true",
                    help="also time the pass with the pooled features in pinned HOST memory (PCIe-inclusive; extra field, never `value`)")
    ap.add_argument("--cpu-sample", default="1024x4000", help="queries x videos for the CPU baseline sample")
    return ap.parse_args()


def site_work(site, cfg):
    """Algorithmic FLOP per launch of an instrumented site (SURVEY.md 8d figures), and which roof bounds it."""
    B, C, Nv, H, E, F, hid, D, n, Nq = (cfg[k] for k in ("Bq", "C", "Nv", "H", "E", "F", "hid", "D", "n", "Nq"))
    M = n * (n + 1) // 2
    # threshold ladder of the top-k pass (score.hip: PRE_VIDEOS, PRE_LEVELS, pre_b_videos): stage A = the 1- and 2-clip moments of
    # 64 videos (its work is not counted: the sample is scored again in full by stage B), stage B = Nv/16 <= 640 videos
    nb = min(640, Nv // 16) if Nv >= 256 else 0
    table = {
        # fused step [x_t | h] x [Wih | Whh]^T; rows actually processed per launch, averaged over the T steps: the
        # forward direction steps every query (+1 all-pad row), the reverse direction only the queries that have
        # reached a real token (their trailing-pad prefix is shared through the all-pad row)
        "gemm_lstm_rec": 2.0 * cfg["lstm_rows_per_step"] * 4 * H * (E + H),
        "gemm_lstm_in": 2.0 * 2 * B * 4 * H * E,
        "gemm_vis_seg": 2.0 * C * hid * F,
        "gemm_vis_ctx": 2.0 * Nv * hid * F,
        "gemm_vis_out": 2.0 * C * D * hid,
        "gemm_lang_fc": 2.0 * B * D * 2 * H,
        # scoring launches (2nD contraction + n norms + 2M moment means per scoring).  With top-k the first
        # Nv/16 (<= 640) videos are stage B of the threshold ladder (site score_prepass, together with stage A's short-moment
        # pass over 32 of them); the main fused launch (top-k + rank keys, one distance pass) covers the rest.
        "score_fused": float(Nq) * max(Nv - nb, 0) * (2 * n * D + n + 2 * M),
        "score_rank": float(Nq) * Nv * (2 * n * D + n + 2 * M),          # k = 0 calls only
        "score_prepass": float(Nq) * nb / 2 * (2 * n * D + n + 2 * M),   # two launches (A, B); B's work averaged over both
    }
    return table.get(site)


def cpu_baseline(args, n_clips):
    """The oracle's full pass (clip MLP + BiLSTM + scoring/top-k) on a bounded sample, host cores."""
    from oracle import oracle as orc
    from vfr_amd import synth
    nq, nv = (int(x) for x in args.cpu_sample.split("x"))
    threads = min(os.cpu_count() or 1, 16)
    orc.set_threads(threads)
    counts = synth.clip_counts(nv, n_clips, seed=1)
    off = synth.clip_offsets(counts)
    seg, ctx = synth.video_features(counts, args.feat_dim, seed=1)
    tokens = synth.query_tokens(nq, seed=1)
    sd = synth.model_weights(args.feat_dim, seed=1)
    lstm = {k[5:]: v for k, v in sd.items() if k.startswith("lstm.")}
    t0 = time.perf_counter()
    V = orc.visual_mlp(seg, ctx, off, sd["visual_fc.0.weight"], sd["visual_fc.0.bias"], sd["visual_fc.2.weight"], sd["visual_fc.2.bias"])
    Q = orc.bilstm_final(tokens, sd["word_embedding.weight"], lstm, sd["lang_fc.weight"], sd["lang_fc.bias"])
    orc.score_topk(Q, V, off, args.k)
    dt = time.perf_counter() - t0
    return {"value": nq * nv / dt, "unit": "scorings/s", "cores": threads, "kind": "port",
            "sample": f"{nq} queries x {nv} videos ({n_clips} clips x {args.feat_dim}-d), full pass (clip MLP + BiLSTM + "
                      f"score/top-{args.k}) in {dt:.1f} s with the C oracle, OpenMP x{threads}"}


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
    if world > 1:
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
    ws = _vfr.topk_workspace(Nq, hi - lo, args.k, dev)

    def ranks_of(shard, Q):
        # a11 inside the step: the label table of the batch (model/evaluate.py:59-65) is rebuilt from the spans every pass
        labels = ops.gt_labels(times_dev, n_own_dev, IOU, True, dev, Mmax=Mmax_own)
        gt = engine.prepare_gt(shard, own, labels, index=gt_idx)
        return engine.corpus_ranks(shard, Q, own, labels, ops, k=args.k, world=world, workspace=ws, gt=gt)

    def step():
        with torch.no_grad():
            # a rank's two encoders are small enough from N = 4 on to leave CUs idle in their partial tile rounds: run them side
            # by side there (tools/rank_sim.py: -0.16 ms at 8, -0.09 at 4, +0.08 at 2); below they stay back to back (the per-kernel durations quoted in `roofline` are then undisturbed)
            emb, Q = engine.overlapped(dev, lambda: model.encode_clips(seg, ctx, clip_off),
                                       lambda: engine.encode_queries(model, tokens, dev, ops, rank, world), enable=world >= 4)
            return ranks_of(make_shard(emb), Q)

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
               n=int(round(n_eff)), Nq=Nq, T=T_, lstm_rows_per_step=(Bq + 1) + rev_rows)
    if not sites:                                  # VFR_BENCH_NO_SITES rehearsal: nothing to attribute
        print(json.dumps({"ms_per_step": ms_step, "value": value, "n_gpus": world, "note": "site events off (rehearsal)"}))
        if dist is not None:
            dist.destroy_process_group()
        return
    kernels, dom, dom_ms = {}, None, -1.0
    for name, (ms, cnt) in sites.items():
        fl = site_work(name, cfg)
        kernels[name] = {"ms_per_step": ms / args.steps, "launches_per_step": cnt / args.steps,
                         "tflops": (fl * cnt / (ms * 1e-3) / 1e12) if fl else None}
        if ms > dom_ms:
            dom, dom_ms = name, ms
    dom_ms_launch = dom_ms / sites[dom][1]
    fl = site_work(dom, cfg)
    achieved = fl / (dom_ms_launch * 1e-3) / 1e12 if fl else None
    traffic = None                      # HBM bytes/launch of this kernel from the committed PMC passes (profiles/README.md)
    tfile = ROOT / "profiles" / "traffic_latest.json"
    if tfile.exists():
        traffic = json.loads(tfile.read_text()).get(dom)
    roofline = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (achieved / FP32_PEAK_TFLOPS) if achieved else None, "traffic": traffic,
                "avg_launch_ms": dom_ms_launch, "launches_per_step": sites[dom][1] / args.steps,
                "share_of_step": dom_ms / args.steps / ms_step}
    line = {
        "metric": "query x video scorings/sec, full evaluate pass (clip MLP + BiLSTM + moment scoring/top-k/rank), DiDeMo-shape",
        "value": value, "unit": "scorings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config 1: {Nv} videos x {args.clips} clips x {F}-d fc7 features, {Nq}-query batch, "
                               f"top-{args.k} + rank@IoU{{0.5,0.7}}, fp32, videos sharded over {world} GPU(s)",
                   "videos": Nv, "clips": args.clips, "queries": Nq, "k": args.k, "parallelism": f"shard{world}"},
        "gpu_event_ms_per_step": ev0.elapsed_time(ev1) / args.steps,
        "median_rank_check": float(ranks[0].float().median()),
        "ranks_checksum": int(ranks.sum()), "topk_checksum": int(out[2].sum()) if out[2] is not None else None,
        "roofline": roofline, "kernels": kernels,
    }
    if host_feed is not None:
        line["pcie_inclusive"] = host_feed
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, n_clips)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
