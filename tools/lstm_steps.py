#!/usr/bin/env python3
"""Per-launch duration of the fused LSTM step over one pass (20 launches) from a rocprofv3 --kernel-trace CSV of bench.py,
beside the number of active 64-row tiles of each launch (forward 5001 rows + the reverse rows past their pads) and the rounds
of the chip's 768 workgroup slots that is.  usage: lstm_steps.py <kernel_trace.csv> [queries=5000]"""
import csv
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vfr_amd import synth  # noqa: E402

Nq = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lstm_step_mfma_pair" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-20:]
tok = synth.query_tokens(Nq, seed=123)
T = tok.shape[1]
qlen = np.where(tok != 0, np.arange(1, T + 1)[None, :], 0).max(axis=1)
tot = 0.0
for s, r in enumerate(last):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fwd, rev = Nq + 1, 1 + int((qlen > T - 1 - s).sum())
    tiles = (-(-fwd // 64) + -(-rev // 64)) * 32
    tot += us
    print(f"step {s:2d}: {us:7.1f} us   rows {fwd} + {rev:4d}   workgroups {tiles:5d} = {tiles / 768:5.2f} rounds of 768   {us / tiles * 768:6.1f} us per full round"
          f"   {2.0 * (fwd + rev) * 4000 * 1000 / us / 1e6:6.1f} TF executed")
print(f"sum {tot / 1e3:.3f} ms")
