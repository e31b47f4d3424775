#!/usr/bin/env python3
"""Feature-store feed of the clip encoder at BASELINE configs[1] size (10k videos x 21 clips x 4096 f32 = 3.6 GB).

Times, on the GPU box: (a) the clip encoder with rows resident in HBM (what bench.py's value uses), (b) one blocking
H2D of the whole bank from pinned memory followed by the encoder, (c) engine.encode_clips_streamed from pinned memory
(chunked copies overlapped with the encoder), (d) the same from the pageable mmap of the store file.
Usage: store_bench.py [Nv] [clips] [reps]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import vfr_amd  # noqa
from vfr_amd import engine, synth
from vfr_amd.store import FeatureStore
from helpers import make_model

Nv = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 21
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
F = 4096
counts = np.full(Nv, clips, np.int64)
off = synth.clip_offsets(counts)
C = int(off[-1])
g = torch.Generator().manual_seed(1)
seg = torch.rand((C, F), generator=g)
seg /= seg.norm(dim=1, keepdim=True) + 1e-5
ctx = torch.rand((Nv, F), generator=g)
ctx /= ctx.norm(dim=1, keepdim=True) + 1e-5
td = tempfile.mkdtemp(dir="/tmp")
t0 = time.time()
FeatureStore.write(Path(td) / "c.vfs", [f"v{i}" for i in range(Nv)], off, ctx.numpy(), seg.numpy())
t1 = time.time()
st_map = FeatureStore.open(Path(td) / "c.vfs")
t2 = time.time()
st_pin = FeatureStore.open(Path(td) / "c.vfs", pin=True)
t3 = time.time()
gb = (C + Nv) * F * 4 / 1e9
print(f"corpus {Nv} x {clips} clips = {gb:.2f} GB   write {t1 - t0:.2f}s  open(mmap) {1e3 * (t2 - t1):.1f} ms  "
      f"open+pin {t3 - t2:.2f}s ({gb / (t3 - t2):.1f} GB/s)", flush=True)
model = make_model(synth.model_weights(F, seed=1)).to(dev)
ops = engine.ops_for(dev)
off_dev = st_pin.clip_off.to(dev)


def timed(fn, n=reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, r


seg_d, ctx_d = st_pin.seg.to(dev), st_pin.ctx.to(dev)
ta, ref = timed(lambda: model.encode_clips(seg_d, ctx_d, off_dev))
del seg_d, ctx_d
tb, rb = timed(lambda: model.encode_clips(st_pin.seg.to(dev, non_blocking=True), st_pin.ctx.to(dev, non_blocking=True), off_dev))
tcopy, _ = timed(lambda: (st_pin.seg.to(dev, non_blocking=True), st_pin.ctx.to(dev, non_blocking=True)))
print(f"(a) resident encode          {ta:8.2f} ms")
print(f"    H2D alone (pinned)       {tcopy:8.2f} ms  {gb / tcopy * 1e3:.1f} GB/s")
print(f"(b) H2D then encode          {tb:8.2f} ms  bit-equal {torch.equal(rb, ref)}")
for chunk in (1 << 13, 1 << 14, 1 << 15, 1 << 16):
    tc, rc = timed(lambda: engine.encode_clips_streamed(ops, model, st_pin.seg, st_pin.ctx, off, dev, chunk_clips=chunk))
    print(f"(c) streamed pinned  chunk {chunk:6d} clips {tc:8.2f} ms  {gb / tc * 1e3:.1f} GB/s  bit-equal {torch.equal(rc, ref)}", flush=True)
td_, rd = timed(lambda: engine.encode_clips_streamed(ops, model, st_map.seg, st_map.ctx, off, dev), n=2)
print(f"(d) streamed from mmap (pageable)   {td_:8.2f} ms  {gb / td_ * 1e3:.1f} GB/s  bit-equal {torch.equal(rd, ref)}")
import shutil
shutil.rmtree(td)
