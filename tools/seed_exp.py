import sys, time
sys.path.insert(0, '/root/repo')
import torch, vfr_amd
from vfr_amd import _vfr, engine
nq, nv, n = 5000, 10000, 21
dev = "cuda:0"
torch.manual_seed(0)
V = torch.randn(nv * n, 100, device=dev) * 0.1
Q = torch.randn(nq, 100, device=dev) * 0.1
off = torch.arange(0, nv * n + 1, n, dtype=torch.int32, device=dev)
bank = _vfr.VideoBank(V, off)
ws = _vfr.topk_workspace(nq, nv, 100, dev)
d, i, _ = _vfr.score_topk(Q, bank, 100, workspace=ws)
def timed(label, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t) / reps * 1e3:9.3f} ms", flush=True)
for pos, name in ((99, "exact 100th"), (49, "50th (too tight, wrong result)")):
    pass
seed = engine._pack_key(d[:, 99].contiguous(), i[:, 99]).contiguous()
timed("top-100, own prepass", lambda: _vfr.score_topk(Q, bank, 100, workspace=ws))
timed("top-100, thr_seed = exact 100th key", lambda: _vfr.score_topk(Q, bank, 100, workspace=ws, thr_seed=seed))
for mult in (1.02, 1.05, 1.1):
    s2 = engine._pack_key((d[:, 99] * mult).contiguous(), i[:, 99]).contiguous()
    timed(f"top-100, thr_seed = 100th dist x {mult}", lambda: _vfr.score_topk(Q, bank, 100, workspace=ws, thr_seed=s2))
d2, i2, _ = _vfr.score_topk(Q, bank, 100, workspace=ws, thr_seed=seed)
print("same result with seed:", bool((i2 == i).all()), bool((d2 == d).all()))
