#!/usr/bin/env python3
"""The multi-step BiLSTM launch (EXPERIMENT `lstm_multi` 1: all T steps in one kernel; with `lstm_tile` 2: 128-row tiles) against one
launch per step (`lstm_multi` 0, the default): bits and time of the whole query-encoder pass.

    python tools/lstm_multi_check.py [B ...]      (default: 64 130 313 626 1250 2500 5000)
"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vfr_amd  # noqa: E402,F401
from vfr_amd import _vfr, synth  # noqa: E402


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [64, 130, 313, 626, 1250, 2500, 5000]
    dev = torch.device("cuda", 0)
    sd = synth.model_weights(4096, seed=7)
    lstm = {k.split("lstm.")[1]: torch.from_numpy(v).to(dev) for k, v in sd.items() if k.startswith("lstm.")}
    emb = torch.from_numpy(sd["word_embedding.weight"]).to(dev)
    wfc, bfc = torch.from_numpy(sd["lang_fc.weight"]).to(dev), torch.from_numpy(sd["lang_fc.bias"]).to(dev)
    for B in Bs:
        tokens = torch.from_numpy(synth.query_tokens(B, seed=B)).to(dev)
        row = {}
        outs = {}
        for multi in (0, 1, 2):
            _vfr.set_option("lstm_multi", 1 if multi else 0)
            _vfr.set_option("lstm_tile", 2 if multi == 2 else 0)
            out = _vfr.bilstm_final(tokens, emb, lstm, wfc, bfc); torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 12
            for _ in range(n):
                out = _vfr.bilstm_final(tokens, emb, lstm, wfc, bfc)
            torch.cuda.synchronize()
            row[multi] = (time.perf_counter() - t0) / n * 1e3
            outs[multi] = out
        _vfr.set_option("lstm_multi", 0); _vfr.set_option("lstm_tile", 0)
        same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)) and torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32))
        print(f"B = {B:5d}: per-step launches {row[0]:7.3f} ms   one launch {row[1]:7.3f} ms   one launch, 128-row tiles {row[2]:7.3f} ms   {'IDENTICAL' if same else '*** DIFFERENT ***'}"
              f"   faults {_vfr.poll_faults()}", flush=True)
        if not same:
            d = (outs[0] - outs[1]).abs()
            print("   max |diff|", float(d.max()), " NaNs", int(torch.isnan(outs[1]).sum()))


if __name__ == "__main__":
    main()
