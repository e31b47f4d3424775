#!/usr/bin/env python3
"""The scoring pass of the bench corpus alone, A/B over library options (GPU box).

    python tools/scorer_ab.py [--videos 10000] [--queries 5000] [--clips 21] name=v1,v2,... [name2=...]

Builds the bench corpus's embeddings (seeded features through the clip / query encoders, as bench.py does), then times the
scoring pass -- labels + own-video scores + fused top-100 + rank counts at IoU 0.5 / 0.7, what `engine.corpus_ranks` runs --
in two regimes: `bench` (the encoded queries: rank keys mid-distribution) and `planted` (bench.py's realistic_gt: every query
next to the clips of its first annotated span: rank keys in the near tail).  Every combination of the option values given on
the command line is run (`--zip`: the i-th values of every option together); per combination the profiler sites of the scorer are printed, and the outputs (rank counts, top-k ids
and distances) must be IDENTICAL across all combinations (the first is the reference) -- a mismatch is an error."""
import itertools
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vfr_amd  # noqa: E402,F401
from vfr_amd import _vfr, engine, models, synth  # noqa: E402

SITES = ("score_fused", "score_prepass", "score_rank", "score_pairs", "score_finish", "score_prep", "score_fallback", "topk_merge")


def main():
    args = sys.argv[1:]
    Nv, Nq, clips, k, F = 10000, 5000, "21", 100, 4096
    sweeps = []
    zipped = False                          # --zip: the i-th values of every option together (e.g. all switches off, then all on)
    i = 0
    while i < len(args):
        if args[i] == "--videos": Nv = int(args[i + 1]); i += 2
        elif args[i] == "--queries": Nq = int(args[i + 1]); i += 2
        elif args[i] == "--clips": clips = args[i + 1]; i += 2
        elif args[i] == "--feat-dim": F = int(args[i + 1]); i += 2
        elif args[i] == "--zip": zipped = True; i += 1
        else:
            name, vals = args[i].split("=")
            sweeps.append((name, [int(v) for v in vals.split(",")]))
            i += 1
    dev = torch.device("cuda", 0)
    n_clips = clips if clips == "didemo" else int(clips)
    counts = synth.clip_counts(Nv, n_clips, seed=123)
    off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    mom = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * (counts + 1) // 2)])
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    raw = torch.rand((int(off[-1]), F), generator=gen, device=dev)
    seg = raw / (raw.norm(dim=1, keepdim=True) + 1e-5)
    clip_off = torch.from_numpy(off.astype(np.int32)).to(dev)
    nloc = (clip_off[1:] - clip_off[:-1]).long()
    ctx = torch.segment_reduce(raw, "sum", lengths=nloc, axis=0) / nloc[:, None].float()
    ctx = ctx / (ctx.norm(dim=1, keepdim=True) + 1e-5)
    del raw
    tokens = torch.from_numpy(synth.query_tokens(Nq, seed=123)).to(dev)
    own, times = synth.annotations(Nq, counts, seed=123)
    sd = synth.model_weights(F, seed=123)
    model = models.CALModel(2 * F + 2, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
    model.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()})
    model = model.to(dev).eval()
    ops = engine.HipOps()
    with torch.no_grad():
        emb = model.encode_clips(seg, ctx, clip_off)
        Q = engine.encode_queries(model, tokens, dev, ops)
    del seg, ctx
    bank = _vfr.VideoBank(emb, clip_off, 0, max_clips=int(counts.max()), total_moments=int(mom[-1]), min_clips=int(counts.min()))
    shard = engine.CorpusShard(bank, 0, Nv, counts, mom, dev)
    t_h, na_h = engine.pack_times(times)
    times_dev = (torch.from_numpy(t_h).to(dev), torch.from_numpy(na_h).to(dev))
    n_own_dev = torch.from_numpy(counts[own].astype(np.int32)).to(dev)
    nmax = int(counts[own].max())
    gt_idx = engine.gt_index(shard, own)
    ws = _vfr.topk_workspace(Nq, Nv, k, dev, total_clips=int(off[-1]))

    def run(Qx):
        labels = ops.gt_labels(times_dev, n_own_dev, [0.5, 0.7], True, dev, Mmax=nmax * (nmax + 1) // 2)
        gt = engine.prepare_gt(shard, own, labels, index=gt_idx)
        return engine.corpus_ranks(shard, Qx, own, labels, ops, k=k, world=1, workspace=ws, gt=gt)

    # planted queries (bench.py realistic_gt)
    s0 = np.asarray([t[0][0] for t in times]); e0 = np.asarray([t[0][1] for t in times])
    first = torch.from_numpy(off[own] + s0).to(dev); last = torch.from_numpy(off[own] + e0).to(dev)
    csum = torch.cat([torch.zeros((1, 100), device=dev, dtype=torch.float64), emb.double().cumsum(0)])
    centre = ((csum[last + 1] - csum[first]) / (last + 1 - first)[:, None].double()).float()
    g = torch.Generator(device=dev); g.manual_seed(4321)
    a, b = torch.randint(0, emb.shape[0], (2, 8192), device=dev, generator=g)
    d_typ = float((emb[a] - emb[b]).norm(dim=1).median())
    Qp = (centre + 0.5 * d_typ / 10.0 * torch.randn(centre.shape, device=dev, generator=g)).contiguous()

    names = [n for n, _ in sweeps]
    combos = (list(zip(*[v for _, v in sweeps])) if zipped else list(itertools.product(*[v for _, v in sweeps]))) or [()]
    print(f"# scorer_ab: {Nv} videos x {clips} clips, {Nq} queries, k = {k}; options {names or '(defaults)'}")
    for regime, Qx in (("bench", Q), ("planted", Qp)):
        ref = None
        for combo in combos:
            for n_, v_ in zip(names, combo):
                _vfr.set_option(n_, v_)
            with torch.no_grad():
                out = run(Qx); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    out = run(Qx)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 3 * 1e3
                _vfr.set_option("profile", 1); _vfr.profile_read(reset=True)
                run(Qx); torch.cuda.synchronize()
                _vfr.set_option("profile", 0)
                sites = _vfr.profile_read(reset=True)
            st = _vfr.score_mfma_stats(ws, Nq, bank, k) if _vfr.DEFAULT_SCORE_MODE == "mfma" else {}
            row = {s_: round(sites[s_][0], 3) for s_ in SITES if s_ in sites}
            same = "reference" if ref is None else ("IDENTICAL" if all(torch.equal(x, y) for x, y in zip(out, ref)) else "*** DIFFERENT ***")
            if ref is None:
                ref = out
            print(f"{regime:8s} {dict(zip(names, combo))}: pass {ms:7.3f} ms  scorer sites {sum(row.values()):7.3f} ms  {row}  "
                  f"exact pairs {st.get('exact_pair_fraction', 0) * 100:.3f} %  fallback groups {st.get('fallback_groups')}  "
                  f"median rank {float(out[0][0].float().median()):.0f}  checksums {int(out[0].sum())} / {int(out[2].sum())}  {same}", flush=True)
            if "DIFFERENT" in same:
                raise SystemExit("results differ between option settings")


if __name__ == "__main__":
    main()
