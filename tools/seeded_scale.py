#!/usr/bin/env python3
"""Seeded main pass of the sharded search vs shard size: T(Nv) for 5000 queries, top-100 + 2 rank keys, thr_seed = k-th
key of a 256-video global sample (what a rank runs at N GPUs).  Shows the fixed cost per pass.  usage: seeded_scale.py [score_tasks values, comma separated]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfr_amd  # noqa
from vfr_amd import _vfr, engine

dev = "cuda:0"
Nq, n, D, k = 5000, 21, 100, 100
g = torch.Generator(device=dev).manual_seed(3)
Q = torch.randn((Nq, D), device=dev, generator=g) * 0.1
Vall = torch.randn((10000 * n, D), device=dev, generator=g) * 0.1
counts_all = np.full(10000, n, np.int64)
off_all = torch.arange(0, 10000 * n + 1, n, dtype=torch.int32, device=dev)
full = _vfr.VideoBank(Vall, off_all, 0, max_clips=n, total_moments=10000 * 231, min_clips=n)
sample = _vfr.slice_bank(full, counts_all, 0, 256)
sd, si, _ = _vfr.score_topk(Q, sample, k)
seed = engine._pack_key(sd[:, k - 1].contiguous(), si[:, k - 1])
rd = torch.stack([sd[:, 0], sd[:, 3]]).contiguous(); ri = torch.stack([si[:, 0], si[:, 3]]).contiguous()
ws = _vfr.topk_workspace(Nq, 10000, k, dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


TASKS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
for Nv, tasks in [(nv, t) for nv in (312, 625, 1218, 2468, 4968, 9744) for t in TASKS]:
    _vfr.set_option("score_tasks", tasks)
    b = _vfr.slice_bank(full, counts_all, 256, 256 + Nv)
    _vfr.set_option("profile", 1); _vfr.profile_read(reset=True)
    t_seed = timed(lambda: _vfr.score_topk(Q, b, k, rd, ri, workspace=ws, thr_seed=seed))
    sites = _vfr.profile_read(reset=True); _vfr.set_option("profile", 0)
    t_un = timed(lambda: _vfr.score_topk(Q, b, k, rd, ri, workspace=ws))
    t_rank = timed(lambda: _vfr.score_topk(Q, b, 0, rd, ri, workspace=ws))
    detail = "  ".join(f"{nm} {ms / 11:.3f}x{c // 11}" for nm, (ms, c) in sites.items())
    print(f"tasks={tasks:5d} Nv={Nv:5d}  seeded {t_seed:7.3f} ms ({t_seed / Nv * 1e3:6.2f} us/video)  unseeded {t_un:7.3f}  rank-only {t_rank:7.3f}   [{detail}]", flush=True)
