"""Multi-rank path on CPU: world_size 2 over gloo.  Videos are sharded contiguously, the
best-GT key is all_reduce(MIN)'d, rank counts all_reduce(SUM)'d, per-shard top-k lists all_gather'ed and
merged.  Every rank must end with exactly the single-process answer."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import MemoryDataset, make_model, problem


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, clips, out_dir, sample=256, force=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfr_amd import engine
        from vfr_amd import evaluate as vevaluate
        engine.FORCE_COLLECTIVES = force       # one rank running the whole exchange protocol (the RCCL rehearsal's switch)
        engine.SAMPLE_VIDEOS = sample          # small sample: both the sample part and the seeded main part are non-empty
        p = problem(37, 23, clips, feat_dim=64, hidden=16)          # odd sizes: uneven shards, padded query split
        ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
        model = make_model(p["sd"], feat_dim=64, hidden=16)
        vi, li = ds.iterators()
        metrics, (td, ti) = vevaluate.evaluate(model, vi, li, ds.annotations, "cpu", rank=rank, world=world,
                                               return_topk=50)
        vi, li = ds.iterators()
        val = vevaluate.validate_epoch(model, vi, li, ds.annotations, "cpu", size=-1, rank=rank, world=world)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), metrics=json.dumps(metrics), td=td.numpy(), ti=ti.numpy(),
                 val=json.dumps(val))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("clips,sample,world", [(6, 256, 2), ("didemo", 256, 2), ("didemo", 8, 2), ("didemo", 8, 1)])
def test_two_rank_sharded_evaluate_equals_single_process(tmp_path, clips, sample, world):
    """world = 1: ONE rank with engine.FORCE_COLLECTIVES -- the form the GPU suite runs over RCCL on a one-GPU box."""
    from vfr_amd import evaluate as vevaluate
    mp.spawn(_worker, args=(world, _free_port(), clips, str(tmp_path), sample, world == 1), nprocs=world, join=True)
    p = problem(37, 23, clips, feat_dim=64, hidden=16)
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    vi, li = ds.iterators()
    ref_metrics, (rd, ri) = vevaluate.evaluate(make_model(p["sd"], feat_dim=64, hidden=16), vi, li, ds.annotations,
                                               "cpu", return_topk=50)
    vi, li = ds.iterators()
    ref_val = vevaluate.validate_epoch(make_model(p["sd"], feat_dim=64, hidden=16), vi, li, ds.annotations, "cpu", size=-1)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        assert json.loads(str(got["metrics"])) == json.loads(json.dumps(ref_metrics))
        assert json.loads(str(got["val"])) == json.loads(json.dumps(ref_val))      # 11-threshold sweep, sharded
        assert np.array_equal(got["ti"], ri.numpy())
        assert np.array_equal(got["td"], rd.numpy())


def test_shard_ranges_cover_and_balance():
    from vfr_amd import engine
    for nv in (1, 7, 10000):
        for world in (1, 2, 3, 8):
            spans = [engine.shard_range(nv, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nv
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
