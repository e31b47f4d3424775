#!/usr/bin/env python3
"""Query encoder time with the fused LSTM step forced to 64- (1), 128- (2) or 32-row (3) tiles (GPU box).  usage: lstm_tile_sweep.py [B ...]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, models, synth
import os
if os.environ.get("VFR_LIB"):
    _vfr.LIB_PATH = Path(os.environ["VFR_LIB"]).resolve()          # A/B against another build of libvfr
dev = "cuda:0"
sd = synth.model_weights(4096, seed=123)
model = models.CALModel(8194, pretrained_emb=torch.from_numpy(sd["word_embedding.weight"]))
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev).eval()
for B in [int(x) for x in sys.argv[1:]] or [5000, 2500, 1250, 625]:
    tokens = torch.from_numpy(synth.query_tokens(B, seed=123)).to(dev)
    ref = None
    for mode, xcd in ((1, 1), (1, 0), (2, 1), (2, 0), (3, 1), (3, 0)):
        _vfr.set_option("lstm_xcd", xcd)
        _vfr.set_option("lstm_tile", mode)
        with torch.no_grad():
            q = model.encode_queries(tokens); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3): q = model.encode_queries(tokens)
            torch.cuda.synchronize()
        ref = q if ref is None else ref
        print(f"B={B:5d} lstm_tile={mode} xcd={xcd}: {(time.perf_counter() - t) / 3 * 1e3:8.3f} ms  same bits as tile 1: {torch.equal(q, ref)}", flush=True)
_vfr.set_option("lstm_tile", 0)
_vfr.set_option("lstm_xcd", 1)
