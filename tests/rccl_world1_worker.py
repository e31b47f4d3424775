"""Worker of test_gpu_parity.py::test_rccl_world1_runs_the_sharded_protocol: ONE rank, backend "nccl" (= RCCL on ROCm).
With engine.FORCE_COLLECTIVES the world-size-1 group runs the complete sharded protocol of SURVEY.md 8e -- query-slice
all_gather, sample pass + packed [sample keys | best-GT keys] all_gather, seeded main pass, packed [final keys | rank counts]
all_gather, key merges; and the five-collective form (all_reduce MIN / SUM) under validate_epoch's 11 thresholds -- through
the real RCCL calls on device tensors.  Every result must equal the plain single-GPU pass.  Prints one JSON line."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import vfr_amd  # noqa: F401
    from helpers import MemoryDataset, make_model, problem
    from vfr_amd import engine
    from vfr_amd import evaluate as vevaluate
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    try:
        p = problem(400, 150, "didemo", seed=31)
        ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
        model = make_model(p["sd"]).to(dev)
        want, (wd, wi) = vevaluate.evaluate(model, *ds.iterators(), ds.annotations, dev, return_topk=100)
        wv = vevaluate.validate_epoch(model, *ds.iterators(), ds.annotations, dev, size=-1)
        calls = {"all_gather_into_tensor": 0, "all_reduce": 0}
        real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

        def ag(*a, **k):
            calls["all_gather_into_tensor"] += 1
            assert a[0].is_cuda and a[1].is_cuda
            return real_ag(*a, **k)

        def ar(*a, **k):
            calls["all_reduce"] += 1
            assert a[0].is_cuda
            return real_ar(*a, **k)
        dist.all_gather_into_tensor, dist.all_reduce = ag, ar
        engine.FORCE_COLLECTIVES = True
        engine.SAMPLE_VIDEOS = 64                     # sample part and seeded main part both non-empty
        try:
            got, (gd, gi) = vevaluate.evaluate(model, *ds.iterators(), ds.annotations, dev, rank=0, world=1, return_topk=100)
            out["fused_collectives"] = dict(calls)
            gv = vevaluate.validate_epoch(model, *ds.iterators(), ds.annotations, dev, size=-1, rank=0, world=1)
        finally:
            engine.FORCE_COLLECTIVES = False
            dist.all_gather_into_tensor, dist.all_reduce = real_ag, real_ar
        torch.cuda.synchronize()
        out.update(backend=str(dist.get_backend()), gather_form=engine._gather_form(dist), calls=calls,
                   evaluate_equal=bool(got == want), validate_equal=bool(gv == wv),
                   topk_ids_equal=bool(torch.equal(gi, wi)), topk_dist_equal=bool(torch.equal(gd, wd)))
    finally:
        dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out))


if __name__ == "__main__":
    main()
