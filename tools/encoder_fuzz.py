#!/usr/bin/env python3
"""Randomised differential check of the encoders and the pooling against the CPU oracle (bit for bit), GPU box.
Random shapes: clip encoder (F, hidden, D, ragged clip counts incl. empty videos), BiLSTM (B, T, E, H, vocab, query lengths
incl. all-pad rows, normalised-length variant), pooling (T, F, avg/max).  usage: encoder_fuzz.py [iterations] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vfr_amd  # noqa
from vfr_amd import _vfr
from oracle import oracle

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0


def report(what, cfg, ok):
    global bad
    print(f"{what:10s} {cfg}  {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1


for it in range(iters):
    # ---- clip encoder
    F = int(rs.choice([4, 36, 100, 128, 260, 1024]))
    hid = int(rs.choice([1, 7, 64, 129, 500]))
    D = int(rs.choice([1, 5, 100, 130]))
    Nv = int(rs.choice([1, 2, 17, 90, 400]))
    counts = rs.randint(0 if Nv > 1 else 1, int(rs.choice([2, 7, 22, 70])), size=Nv)
    if counts.sum() == 0:
        counts[0] = 3
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    seg = rs.randn(int(off[-1]), F).astype(np.float32)
    ctx = rs.randn(Nv, F).astype(np.float32)
    W1 = (rs.randn(hid, 2 * F + 2) * 0.1).astype(np.float32); b1 = rs.randn(hid).astype(np.float32)
    W2 = (rs.randn(D, hid) * 0.1).astype(np.float32); b2 = rs.randn(D).astype(np.float32)
    want = oracle.visual_mlp(seg, ctx, off, W1, b1, W2, b2)
    got = _vfr.visual_mlp(d(seg), d(ctx), d(off), d(W1), d(b1), d(W2), d(b2)).cpu().numpy()
    report("clip_mlp", f"F={F} hid={hid} D={D} Nv={Nv} C={int(off[-1])}", np.array_equal(got, want))

    # ---- BiLSTM
    B = int(rs.choice([1, 3, 64, 65, 200, 700]))
    T = int(rs.choice([1, 2, 7, 20, 33]))
    E = int(rs.choice([4, 8, 100, 52]))
    H = int(rs.choice([4, 12, 32, 36, 100, 260]))
    vocab = int(rs.choice([2, 5, 60, 400, 5000]))
    Dq = int(rs.choice([1, 100, 33]))
    tokens = np.zeros((B, T), np.int64)
    for b in range(B):
        L = int(rs.randint(0, T + 1))                      # 0 = an all-pad query
        tokens[b, :L] = rs.randint(1, vocab, size=L)
    emb = rs.randn(vocab, E).astype(np.float32); emb[0] = 0
    lstm = {}
    for suf in ("", "_reverse"):
        lstm["weight_ih_l0" + suf] = (rs.randn(4 * H, E) * 0.3).astype(np.float32)
        lstm["weight_hh_l0" + suf] = (rs.randn(4 * H, H) * 0.3).astype(np.float32)
        lstm["bias_ih_l0" + suf] = (rs.randn(4 * H) * 0.1).astype(np.float32)
        lstm["bias_hh_l0" + suf] = (rs.randn(4 * H) * 0.1).astype(np.float32)
    Wfc = (rs.randn(Dq, 2 * H) * 0.2).astype(np.float32); bfc = rs.randn(Dq).astype(np.float32)
    lt = (rs.rand(vocab, 1).astype(np.float32) + 0.5) if rs.randint(2) else None
    want = oracle.bilstm_final(tokens, emb, lstm, Wfc, bfc, lt)
    multi = int(rs.randint(2))                             # the multi-step launch (experiment; taken where the shape qualifies)
    _vfr.set_option("lstm_multi", multi)
    try:
        got = _vfr.bilstm_final(d(tokens), d(emb), {k: d(v) for k, v in lstm.items()}, d(Wfc), d(bfc),
                                d(lt) if lt is not None else None).cpu().numpy()
    finally:
        _vfr.set_option("lstm_multi", 0)
    report("bilstm", f"B={B} T={T} E={E} H={H} vocab={vocab} D={Dq} normlen={lt is not None} lstm_multi={multi}", np.array_equal(got, want))

    # ---- BiLSTM at the model's width (E = 100, H = 1000), 1 .. 34 queries: the single-launch sequence kernels (vector chains for
    # one or two queries, the matrix pipe for 3 .. 32) and the tile steps just above them
    if it % 3 == 0:
        B = int(rs.choice([1, 2, 3, 5, 9, 16, 17, 31, 32, 34]))
        T = int(rs.choice([1, 2, 5, 20]))
        E, H, vocab, Dq = 100, 1000, int(rs.choice([3, 60, 400])), int(rs.choice([1, 100, 260]))
        tokens = np.zeros((B, T), np.int64)
        for b in range(B):
            L = int(rs.randint(0, T + 1))
            tokens[b, :L] = rs.randint(1, vocab, size=L)
        emb = rs.randn(vocab, E).astype(np.float32); emb[0] = 0
        lstm = {}
        for suf in ("", "_reverse"):
            lstm["weight_ih_l0" + suf] = (rs.randn(4 * H, E) * 0.05).astype(np.float32)
            lstm["weight_hh_l0" + suf] = (rs.randn(4 * H, H) * 0.03).astype(np.float32)
            lstm["bias_ih_l0" + suf] = (rs.randn(4 * H) * 0.1).astype(np.float32)
            lstm["bias_hh_l0" + suf] = (rs.randn(4 * H) * 0.1).astype(np.float32)
        Wfc = (rs.randn(Dq, 2 * H) * 0.05).astype(np.float32); bfc = rs.randn(Dq).astype(np.float32)
        lt = (rs.rand(vocab, 1).astype(np.float32) + 0.5) if rs.randint(2) else None
        want = oracle.bilstm_final(tokens, emb, lstm, Wfc, bfc, lt)
        got = _vfr.bilstm_final(d(tokens), d(emb), {k: d(v) for k, v in lstm.items()}, d(Wfc), d(bfc),
                                d(lt) if lt is not None else None).cpu().numpy()
        report("bilstm1000", f"B={B} T={T} vocab={vocab} D={Dq} normlen={lt is not None}", np.array_equal(got, want))

    # ---- pooling
    Tf = int(rs.choice([1, 24, 25, 26, 150, 333]))
    Fp = int(rs.choice([4, 100, 2048, 4096]))
    mode = "avg" if rs.randint(2) else "max"
    fr = rs.rand(Tf, Fp).astype(np.float32); fr[fr < 0.3] = 0
    ws, wc = oracle.segment_pool_norm(fr, 25, mode)
    gs, gc = _vfr.segment_pool_norm(d(fr), 25, mode)
    report("pool", f"T={Tf} F={Fp} {mode}", np.array_equal(gs.cpu().numpy(), ws) and np.array_equal(gc.cpu().numpy(), wc))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
