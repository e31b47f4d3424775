#!/usr/bin/env python3
"""ms per query-encoder pass (per-step launches) at the given batch sizes -- for timing-experiment builds (VFR_LIB=.../x_NAME.so;
their results are WRONG by construction, only the time means something).  usage: lstm_pass_time.py [B ...]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa: E402,F401
from vfr_amd import _vfr, synth  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.model_weights(4096, seed=7)
lstm = {k.split("lstm.")[1]: torch.from_numpy(v).to(dev) for k, v in sd.items() if k.startswith("lstm.")}
emb = torch.from_numpy(sd["word_embedding.weight"]).to(dev)
wfc, bfc = torch.from_numpy(sd["lang_fc.weight"]).to(dev), torch.from_numpy(sd["lang_fc.bias"]).to(dev)
for B in [int(x) for x in sys.argv[1:]] or [626, 5000]:
    tokens = torch.from_numpy(synth.query_tokens(B, seed=B)).to(dev)
    for _ in range(2):
        _vfr.bilstm_final(tokens, emb, lstm, wfc, bfc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        _vfr.bilstm_final(tokens, emb, lstm, wfc, bfc)
    torch.cuda.synchronize()
    print(f"B = {B:5d}: {(time.perf_counter() - t0) / 12 * 1e3:7.3f} ms per pass  (lib {_vfr.LIB_PATH.name})", flush=True)
