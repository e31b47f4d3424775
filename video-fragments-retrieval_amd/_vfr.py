"""ctypes binding of ``lib/libvfr.so`` (C ABI: ``include/vfr.h``) for torch tensors on a ROCm device.

PyTorch is plumbing here: it owns device memory and the stream; every kernel is ours.  Wrappers take
``torch`` CUDA(=HIP) tensors, pass ``data_ptr()`` / sizes / ``torch.cuda.current_stream().cuda_stream``
and raise ``RuntimeError`` with ``vfr_last_error()`` on a non-zero return.  There is NO fallback: if the
shared library is missing or a tensor is not on the GPU these functions raise.
"""
from __future__ import annotations

import ctypes
import itertools
import os
import subprocess
from pathlib import Path

import numpy as np
import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["VFR_LIB"]).resolve() if os.environ.get("VFR_LIB") else _PKG / "lib" / "libvfr.so"   # VFR_LIB: A/B another build
_lib = None

_vp, _i32, _i64, _f32, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/vfr.h declares (checked by tests)
SIGNATURES = {
    "vfr_version": (_i32, []),
    "vfr_last_error": (ctypes.c_char_p, []),
    "vfr_set_option": (_i32, [ctypes.c_char_p, _i32]),
    "vfr_get_option": (_i32, [ctypes.c_char_p]),
    "vfr_set_fault_word": (_i32, [_vp]),
    "vfr_poll_faults": (_i32, []),
    "vfr_profile_sites": (_i32, []),
    "vfr_profile_site_name": (ctypes.c_char_p, [_i32]),
    "vfr_profile_read": (_i32, [_i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), _i32]),
    "vfr_math_f32": (_i32, [_i32, _vp, _vp, _vp, _i64, _vp]),
    "vfr_segment_pool_norm_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vfr_segment_pool_norm_batch_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vfr_visual_mlp_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "vfr_visual_mlp_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vfr_linear_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "vfr_bilstm_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32]),
    "vfr_bilstm_final_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp] + [_vp] * 8 + [_i32, _i32, _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "vfr_score_moments_f32": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _f32, _vp, _vp]),
    "vfr_score_own_f32": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp]),
    "vfr_score_topk_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "vfr_score_topk_f32": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i64, _i32, _vp, _vp, _i32,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vfr_score_topk_mfma_prefilter": (_i32, [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "vfr_score_topk_mfma_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "vfr_score_topk_mfma": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i64, _i32, _vp, _vp, _i32,
                                   _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "vfr_score_topk_mfma_stats": (_i32, [_vp, _i64, _i32, _i32, _i32, ctypes.POINTER(ctypes.c_int64), _vp]),
    "vfr_mfma_selfcheck": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "vfr_topk_merge_f32": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "vfr_topk_pack_keys": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "vfr_topk_merge_keys": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "vfr_topk_merge_keys_strided": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "vfr_gt_best_keys_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "vfr_gt_rank_keys_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vfr_gt_labels_u8": (_i32, [_vp, _vp, _vp, _i64, _i32, ctypes.POINTER(ctypes.c_double), _i32, _i32, _i32, _vp, _vp]),
    "vfr_ranking_loss_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "vfr_ranking_loss_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _sz, _vp]),
    "vfr_ranking_loss_grad_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp]),
    "vfr_transpose_f32": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "vfr_colsum_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "vfr_relu_backward_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "vfr_lstm_cell_forward_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "vfr_lstm_cell_backward_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "vfr_bilstm_train_forward_f32": (_i32, [_vp, _i64, _i32, _i32, _i32] + [_vp] * 8 + [_vp, _vp, _vp, _vp]),
    "vfr_bilstm_train_backward_workspace_bytes": (_sz, [_i64, _i32]),
    "vfr_bilstm_train_backward_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "vfr_frames_normalize_f32": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "vfr_conv3x3_relu_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "vfr_maxpool2_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vfr_adaptive_avgpool7_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vfr_vgg_fc7_workspace_bytes": (_sz, [_i32, _i32, _i32, _vp, _i32, _i32]),
    "vfr_resnet_pool_workspace_bytes": (_sz, [_i32, _i32, _i32, _vp, _i32]),
    "vfr_resnet_pool_f32": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _f32, _vp, _vp, _sz, _vp]),
    "vfr_resnet_folded_bytes": (_sz, [_vp, _i32]),
    "vfr_resnet_fold_f32": (_i32, [_vp, _i32, _vp, _vp, _f32, _vp, _sz, _vp]),
    "vfr_resnet_pool_folded_f32": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _sz, _vp, _vp, _sz, _vp]),
    "vfr_vgg_fc7_f32": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
}


def build(force: bool = False) -> Path:
    """Compile the HIP sources for gfx950 (``make`` drives hipcc; cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_PKG / "csrc"), "-j8"] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libvfr.so build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no fallback path for the HIP kernels)")
        l = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().vfr_last_error().decode()}")


def _dev(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the ROCm device (no CPU fallback in the HIP path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def set_option(name: str, value: int) -> None:
    _check(lib().vfr_set_option(name.encode(), int(value)), "vfr_set_option")


def get_option(name: str) -> int:
    return int(lib().vfr_get_option(name.encode()))


FAULT_SEQ_RESCUED = 1                      # include/vfr.h VFR_FAULT_SEQ_RESCUED
_FAULT_WORD = None                         # page-locked int32 word the kernels raise fault bits in (registered once per process)
FAULT_LOG = []                             # (bits, message) of every fault poll_faults() has reported


def _register_fault_word() -> None:
    """Give the library its fault word (include/vfr.h vfr_set_fault_word): device-side recoveries -- a single-launch sequence
    encoder that gave up and was re-encoded by the rescue kernel -- are reported through it, lazily, by ``poll_faults``."""
    global _FAULT_WORD
    if _FAULT_WORD is None:
        word = torch.zeros(1, dtype=torch.int32).pin_memory()
        _check(lib().vfr_set_fault_word(word.data_ptr()), "vfr_set_fault_word")
        _FAULT_WORD = word


def poll_faults() -> int:
    """Fault bits raised on the device since the last poll (0: none).  Host-only, never synchronises: call it after a point
    where the work of interest has completed (``.cpu()`` / ``.item()`` / ``torch.cuda.synchronize()``); the wrappers of the
    entry points that can raise a fault also poll on entry, so a fault is reported at the latest by the next such call.
    Every reported fault is a ``RuntimeWarning`` carrying ``vfr_last_error()`` and an entry of ``FAULT_LOG`` -- the results
    of the affected call were REPAIRED on the device (include/vfr.h), nothing needs re-running."""
    if _FAULT_WORD is None:
        return 0
    bits = int(lib().vfr_poll_faults())
    if bits:
        import warnings
        msg = lib().vfr_last_error().decode()
        FAULT_LOG.append((bits, msg))
        warnings.warn(f"vfr: device-side recovery (fault bits {bits:#x}): {msg}", RuntimeWarning, stacklevel=2)
    return bits


def profile_read(reset: bool = True) -> dict:
    """{site name: (total device ms, launches)} since the last reset.  Synchronise the device first."""
    l = lib()
    out = {}
    n = l.vfr_profile_sites()
    for site in range(1, n):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        _check(l.vfr_profile_read(site, ctypes.byref(ms), ctypes.byref(cnt), 0), "vfr_profile_read")
        if cnt.value:
            out[l.vfr_profile_site_name(site).decode()] = (ms.value, cnt.value)
    if reset:
        _check(l.vfr_profile_read(1, None, None, 1), "vfr_profile_read")
    return out


# ------------------------------------------------------------------------------------------------
def math_f32(op: int, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    y = _dev(y, torch.float32, "y") if y is not None else None
    out = torch.empty_like(x)
    _check(lib().vfr_math_f32(op, x.data_ptr(), _ptr(y), out.data_ptr(), x.numel(), _stream()), "vfr_math_f32")
    return out


def segment_pool_norm(frames: torch.Tensor, seg_len: int = 25, mode: str = "avg"):
    """[T,F] fc7 frames -> (seg [ceil(T/seg_len),F], ctx [F])  (model/data.py:163-181)."""
    frames = _dev(frames, torch.float32, "frames")
    T, F = frames.shape
    nseg = (T + seg_len - 1) // seg_len
    seg = torch.empty((nseg, F), dtype=torch.float32, device=frames.device)
    ctx = torch.empty((F,), dtype=torch.float32, device=frames.device)
    _check(lib().vfr_segment_pool_norm_f32(frames.data_ptr(), T, F, seg_len, 0 if mode == "avg" else 1,
                                           seg.data_ptr(), ctx.data_ptr(), _stream()), "vfr_segment_pool_norm_f32")
    return seg, ctx


def segment_pool_norm_batch(frames: torch.Tensor, frame_counts, seg_len: int = 25, mode: str = "avg"):
    """frames [sum T, F] for many videos, frame_counts = list of T_v -> (seg [sum n,F], ctx [Nv,F], clip_counts)."""
    frames = _dev(frames, torch.float32, "frames")
    F = frames.shape[1]
    counts = torch.as_tensor(list(frame_counts), dtype=torch.int64)
    nseg = (counts + seg_len - 1) // seg_len
    foff = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32).to(frames.device)
    soff = torch.cat([torch.zeros(1, dtype=torch.int64), nseg.cumsum(0)]).to(torch.int32).to(frames.device)
    total = int(nseg.sum())
    Nv = len(counts)
    seg = torch.empty((total, F), dtype=torch.float32, device=frames.device)
    ctx = torch.empty((Nv, F), dtype=torch.float32, device=frames.device)
    _check(lib().vfr_segment_pool_norm_batch_f32(frames.data_ptr(), foff.data_ptr(), soff.data_ptr(), Nv, total, F,
                                                 seg_len, 0 if mode == "avg" else 1, seg.data_ptr(), ctx.data_ptr(),
                                                 _stream()), "vfr_segment_pool_norm_batch_f32")
    return seg, ctx, nseg.to(torch.int32)


def visual_mlp(seg, ctx, clip_off, W1, b1, W2, b2) -> torch.Tensor:
    seg, ctx = _dev(seg, torch.float32, "seg"), _dev(ctx, torch.float32, "ctx")
    W1, b1, W2, b2 = (_dev(t, torch.float32, n) for t, n in ((W1, "W1"), (b1, "b1"), (W2, "W2"), (b2, "b2")))
    clip_off = _dev(clip_off, torch.int32, "clip_off")
    C, F = seg.shape
    Nv, hid, D = ctx.shape[0], W1.shape[0], W2.shape[0]
    if W1.shape[1] != 2 * F + 2 or clip_off.numel() != Nv + 1 or W2.shape[1] != hid:
        raise RuntimeError("visual_mlp: inconsistent shapes")
    out = torch.empty((C, D), dtype=torch.float32, device=seg.device)
    ws_bytes = lib().vfr_visual_mlp_workspace_bytes(C, Nv, F, hid)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=seg.device)
    _check(lib().vfr_visual_mlp_f32(seg.data_ptr(), ctx.data_ptr(), clip_off.data_ptr(), Nv, C, F, W1.data_ptr(),
                                    b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), hid, D, out.data_ptr(), ws.data_ptr(),
                                    ws_bytes, _stream()), "vfr_visual_mlp_f32")
    return out


def linear(A, W, b=None, relu: bool = False) -> torch.Tensor:
    A, W = _dev(A, torch.float32, "A"), _dev(W, torch.float32, "W")
    b = _dev(b, torch.float32, "b") if b is not None else None
    out = torch.empty((A.shape[0], W.shape[0]), dtype=torch.float32, device=A.device)
    _check(lib().vfr_linear_f32(A.data_ptr(), A.shape[0], A.shape[1], W.data_ptr(), _ptr(b), W.shape[0], int(relu),
                                out.data_ptr(), _stream()), "vfr_linear_f32")
    return out


_LSTM_NAMES = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")


def bilstm_final(tokens, emb, lstm: dict, Wfc, bfc, len_tab=None) -> torch.Tensor:
    """tokens int64 [B,T] -> query embeddings [B,D]  (model/models.py:61-66)."""
    tokens = _dev(tokens, torch.int64, "tokens")
    emb = _dev(emb, torch.float32, "emb")
    ws_ = [_dev(lstm[n], torch.float32, n) for n in _LSTM_NAMES] + \
          [_dev(lstm[n + "_reverse"], torch.float32, n + "_reverse") for n in _LSTM_NAMES]
    Wfc, bfc = _dev(Wfc, torch.float32, "Wfc"), _dev(bfc, torch.float32, "bfc")
    lt = _dev(len_tab, torch.float32, "len_tab") if len_tab is not None else None
    B, T = tokens.shape
    E, H, D = emb.shape[1], ws_[1].shape[1], Wfc.shape[0]
    _register_fault_word()
    poll_faults()                              # (a give-up of an EARLIER call's sequence kernel: repaired there, reported here)
    out = torch.empty((B, D), dtype=torch.float32, device=tokens.device)
    nbytes = lib().vfr_bilstm_workspace_bytes(B, T, E, H, emb.shape[0])
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=tokens.device)
    _check(lib().vfr_bilstm_final_f32(tokens.data_ptr(), B, T, emb.data_ptr(), emb.shape[0], _ptr(lt),
                                      *[w.data_ptr() for w in ws_], E, H, Wfc.data_ptr(), bfc.data_ptr(), D,
                                      out.data_ptr(), ws.data_ptr(), nbytes, _stream()), "vfr_bilstm_final_f32")
    return out


def moment_offsets(clip_off: torch.Tensor) -> torch.Tensor:
    n = (clip_off[1:] - clip_off[:-1]).to(torch.int64)
    return torch.cat([torch.zeros(1, dtype=torch.int64, device=clip_off.device), (n * (n + 1) // 2).cumsum(0)])


class VideoBank:
    """Clip embeddings of a (shard of a) corpus resident in HBM, with the CSR offsets the kernels take.

    ``emb`` [sum n, D] fp32, ``clip_off`` [Nv+1] int32, ``mom_off`` [Nv+1] int64 (all on the device);
    ``max_clips`` / ``total_moments`` are host ints computed once at construction (one D2H of two scalars).
    """

    def __init__(self, emb: torch.Tensor, clip_off: torch.Tensor, id_base: int = 0, max_clips: int | None = None,
                 total_moments: int | None = None, min_clips: int | None = None, mom_off: torch.Tensor | None = None):
        self.emb = _dev(emb, torch.float32, "emb")
        self.clip_off = _dev(clip_off, torch.int32, "clip_off")
        self.mom_off = moment_offsets(self.clip_off) if mom_off is None else _dev(mom_off, torch.int64, "mom_off")
        self.num_videos = int(self.clip_off.numel() - 1)
        if max_clips is None or total_moments is None or min_clips is None:   # scalar D2H reads, skipped when the host knows
            n = self.clip_off[1:] - self.clip_off[:-1]
            max_clips = int(n.max()) if self.num_videos else 0
            min_clips = int(n.min()) if self.num_videos else 0
            total_moments = int(self.mom_off[-1])
        self.max_clips, self.min_clips, self.total_moments = int(max_clips), int(min_clips), int(total_moments)
        self.total_clips = int(self.emb.shape[0])
        self.id_base = int(id_base)
        self.dim = int(self.emb.shape[1])
        self.serial = next(VideoBank._counter)   # identity of this bank object for the workspace-side cache (score_topk); atomic under the GIL

    _counter = itertools.count(1)

    def invalidate(self) -> None:
        """Forget every workspace-side product computed for this bank (call after writing ``emb`` / ``clip_off`` through
        anything torch does not version: ``.data``, DLPack / numpy views, a raw-pointer kernel).  Not needed for
        correctness -- the library hashes the bank on the device before it reuses anything (include/vfr.h,
        VFR_MFMA_BANK_READY) -- but it saves that call the hash + compare."""
        self.serial = next(VideoBank._counter)

    def prep_token(self, dtype: int, eps: float = 1e-6):
        """When the host may CLAIM the bank-side products of the MFMA pre-filter as reusable: this object, no torch-versioned
        in-place edit of its tensors since, the base dtype, the same eps.  The claim is verified on the device."""
        return (self.serial, self.emb.data_ptr(), self.emb._version, self.clip_off.data_ptr(), self.clip_off._version,
                self.num_videos, self.total_clips, int(dtype), float(eps))


def slice_bank(bank: VideoBank, counts, v0: int, v1: int) -> VideoBank:
    """Videos [v0, v1) of ``bank`` as a bank of their own (``counts`` = host clip counts of ``bank``'s videos)."""
    import numpy as np
    counts = np.asarray(counts, np.int64)
    off = np.concatenate([[0], np.cumsum(counts)])
    mom = np.concatenate([[0], np.cumsum(counts * (counts + 1) // 2)])
    sub = counts[v0:v1]
    clip_off = bank.clip_off[v0:v1 + 1] if v0 == 0 else bank.clip_off[v0:v1 + 1] - bank.clip_off[v0]   # on the device
    mom_off = bank.mom_off[v0:v1 + 1] if v0 == 0 else bank.mom_off[v0:v1 + 1] - bank.mom_off[v0]
    return VideoBank(bank.emb[int(off[v0]):int(off[v1])], clip_off.contiguous(), bank.id_base + int(mom[v0]),
                     max_clips=int(sub.max()) if len(sub) else 0, total_moments=int(mom[v1] - mom[v0]),
                     min_clips=int(sub.min()) if len(sub) else 0, mom_off=mom_off.contiguous())


def score_moments(Q: torch.Tensor, bank: VideoBank, eps: float = 1e-6) -> torch.Tensor:
    Q = _dev(Q, torch.float32, "Q")
    out = torch.empty((Q.shape[0], bank.total_moments), dtype=torch.float32, device=Q.device)
    _check(lib().vfr_score_moments_f32(Q.data_ptr(), Q.shape[0], bank.emb.data_ptr(), bank.clip_off.data_ptr(),
                                       bank.mom_off.data_ptr(), bank.num_videos, bank.max_clips, bank.total_moments,
                                       bank.dim, eps, out.data_ptr(), _stream()), "vfr_score_moments_f32")
    return out


def score_own(Q: torch.Tensor, bank: VideoBank, own: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    Q = _dev(Q, torch.float32, "Q")
    own = _dev(own, torch.int32, "own")
    Mmax = bank.max_clips * (bank.max_clips + 1) // 2
    out = torch.empty((Q.shape[0], Mmax), dtype=torch.float32, device=Q.device)
    _check(lib().vfr_score_own_f32(Q.data_ptr(), Q.shape[0], bank.emb.data_ptr(), bank.clip_off.data_ptr(),
                                   own.data_ptr(), bank.max_clips, bank.dim, eps, Mmax, out.data_ptr(), _stream()),
           "vfr_score_own_f32")
    return out


# scoring mode of score_topk: "mfma" = fp32 MFMA pre-filter + exact re-scoring (bit-identical to "exact", the default),
# "exact" = the exact VALU kernels only, "bf16" = bf16 MFMA operands, approximate
SCORE_MODES = {"exact": None, "mfma": 0, "bf16": 1}
MFMA_BANK_READY = 0x100                    # include/vfr.h VFR_MFMA_BANK_READY
DEFAULT_SCORE_MODE = os.environ.get("VFR_SCORE_MODE", "mfma")


_MFMA_CHECKED = {}                         # device index -> bool, one self-check per process and device


def mfma_selfcheck(device=None, reversed_reference: bool = False) -> int:
    """Number of elements (of 256) on which the matrix pipe's 16x16xK product differs from an explicit k-ascending fmaf chain
    (include/vfr.h vfr_mfma_selfcheck) on adversarial rows: exponents spread over 2^-40 .. 2^40, every second product the
    near-negation of its neighbour (cancellation), denormal operands.  0 on a conforming device.  SYNCHRONISES."""
    device = torch.device(device if device is not None else "cuda")
    rs = np.random.RandomState(2024)
    K = 100

    def rows():
        m = rs.uniform(1.0, 2.0, (16, K)) * np.exp2(rs.randint(-40, 41, (16, K))) * rs.choice([-1.0, 1.0], (16, K))
        m[:, 7::13] = 1e-41 * rs.randint(1, 9, m[:, 7::13].shape)          # denormal operands
        return m.astype(np.float32)
    A, B = rows(), rows()
    A[:, 1::2] = -A[:, 0::2] * (1.0 + np.float32(2.0 ** -20))               # a_{2j+1} b_{2j+1} ~ -(a_2j b_2j): cancellation
    B[:, 1::2] = B[:, 0::2]
    a, b = torch.from_numpy(A).to(device), torch.from_numpy(B).to(device)
    mis = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _check(lib().vfr_mfma_selfcheck(a.data_ptr(), b.data_ptr(), K, 1 if reversed_reference else 0, mis.data_ptr(), _stream()),
               "vfr_mfma_selfcheck")
    return int(mis.item())


def _mfma_mode_ok(device) -> bool:
    """The first "mfma" / "bf16" scoring call on a device runs the self-check; a device that fails it gets the exact kernels
    (and a warning) for the rest of the process: the pre-filter's margins assume the checked arithmetic."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ok = _MFMA_CHECKED.get(idx)
    if ok is None:
        ok = _MFMA_CHECKED[idx] = mfma_selfcheck(device) == 0
        if not ok:
            import warnings
            warnings.warn("vfr: the MFMA self-check failed on this device (v_mfma_f32_16x16x4_f32 is not a k-ascending fp32 fma "
                          "chain here); scoring falls back to the exact kernels")
    return ok


def score_topk(Q: torch.Tensor, bank: VideoBank, k: int, rank_dist=None, rank_idx=None, count_lt=None,
               eps: float = 1e-6, workspace: torch.Tensor | None = None, thr_seed: torch.Tensor | None = None,
               mode: str | None = None):
    """Fused scoring + top-k (+ rank counting for up to 4 keys per query).

    rank_dist / rank_idx: [R, Nq] (or [Nq]) -> count_lt [R, Nq] int64 is ADDED to (allocated zeroed when None).
    ``mode``: see SCORE_MODES (None = DEFAULT_SCORE_MODE).
    Returns (dist [Nq,k] | None, idx [Nq,k] int64 | None, count_lt | None)."""
    mode = mode or DEFAULT_SCORE_MODE
    if mode not in SCORE_MODES:
        raise RuntimeError(f"score_topk: unknown mode {mode!r}")
    Q = _dev(Q, torch.float32, "Q")
    if mode != "exact" and not _mfma_mode_ok(Q.device):
        mode = "exact"
    Nq = Q.shape[0]
    od = torch.empty((Nq, k), dtype=torch.float32, device=Q.device) if k > 0 else None
    oi = torch.empty((Nq, k), dtype=torch.int64, device=Q.device) if k > 0 else None
    R = 0
    if rank_dist is not None:
        rank_dist = _dev(rank_dist, torch.float32, "rank_dist").reshape(-1, Nq)
        rank_idx = _dev(rank_idx, torch.int64, "rank_idx").reshape(-1, Nq)
        R = rank_dist.shape[0]
        if count_lt is None:
            count_lt = torch.zeros((R, Nq), dtype=torch.int64, device=Q.device)
        count_lt = _dev(count_lt, torch.int64, "count_lt")
    if thr_seed is not None:
        thr_seed = _dev(thr_seed, torch.int64, "thr_seed")
    dtype = SCORE_MODES[mode]
    if dtype is None:
        nbytes = lib().vfr_score_topk_workspace_bytes(Nq, bank.num_videos, k)
    else:
        nbytes = lib().vfr_score_topk_mfma_workspace_bytes(Nq, bank.num_videos, bank.total_clips, k)
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=Q.device)
    common = (Q.data_ptr(), Nq, bank.emb.data_ptr(), bank.clip_off.data_ptr(), bank.mom_off.data_ptr(), bank.num_videos,
              bank.total_clips, bank.min_clips, bank.max_clips, bank.dim, eps, bank.id_base, k, _ptr(od), _ptr(oi), R,
              _ptr(rank_dist), _ptr(rank_idx), _ptr(count_lt), _ptr(thr_seed))
    if dtype is None:
        workspace._vfr_bank = None               # the exact kernels carve the workspace their own way
        _check(lib().vfr_score_topk_f32(*common, workspace.data_ptr(), nbytes, _stream()), "vfr_score_topk_f32")
    else:
        # the bank-side products of the pre-filter stay in the workspace: a later call on the SAME bank object (and untouched
        # tensors) with this workspace reuses them (VFR_MFMA_BANK_READY); anything else recomputes
        token = bank.prep_token(dtype, eps)
        ready = getattr(workspace, "_vfr_bank", None) == token
        pre = lib().vfr_score_topk_mfma_prefilter(Nq, bank.num_videos, bank.total_clips, bank.max_clips, bank.dim, R, k, dtype)
        workspace._vfr_bank = None
        _check(lib().vfr_score_topk_mfma(*common, dtype | (MFMA_BANK_READY if ready and pre else 0), workspace.data_ptr(), nbytes,
                                         _stream()), "vfr_score_topk_mfma")
        if pre:
            workspace._vfr_bank = token
    return od, oi, count_lt


def score_mfma_stats(workspace: torch.Tensor, Nq: int, bank: VideoBank, k: int) -> dict:
    """What the last ``mode="mfma"`` call on ``workspace`` left to the exact kernels (synchronises)."""
    out = (ctypes.c_int64 * 4)()
    _check(lib().vfr_score_topk_mfma_stats(workspace.data_ptr(), Nq, bank.num_videos, bank.total_clips, k, out, _stream()),
           "vfr_score_topk_mfma_stats")
    return {"groups": out[0], "fallback_groups": out[1], "exact_pairs": out[2], "markable_pairs": out[3],
            "exact_pair_fraction": out[2] / float(max(1, Nq * bank.num_videos))}


def topk_workspace(Nq: int, num_videos: int, k: int, device, total_clips: int | None = None) -> torch.Tensor:
    """Workspace large enough for every scoring mode of a bank of ``num_videos`` videos / ``total_clips`` clips
    (64 clips per video assumed when not given)."""
    total_clips = int(total_clips) if total_clips is not None else 64 * num_videos
    n = max(lib().vfr_score_topk_workspace_bytes(Nq, num_videos, k),
            lib().vfr_score_topk_mfma_workspace_bytes(Nq, num_videos, total_clips, k), 1)
    return torch.empty((n,), dtype=torch.uint8, device=device)


def topk_merge(part_dist: torch.Tensor, part_idx: torch.Tensor):
    """[G,Nq,k] shard lists -> merged [Nq,k] in (distance, id) order."""
    pd, pi = _dev(part_dist, torch.float32, "part_dist"), _dev(part_idx, torch.int64, "part_idx")
    G, Nq, k = pd.shape
    od = torch.empty((Nq, k), dtype=torch.float32, device=pd.device)
    oi = torch.empty((Nq, k), dtype=torch.int64, device=pd.device)
    _check(lib().vfr_topk_merge_f32(pd.data_ptr(), pi.data_ptr(), G, Nq, k, od.data_ptr(), oi.data_ptr(), _stream()),
           "vfr_topk_merge_f32")
    return od, oi


KEY_EMPTY = 0x7F800000FFFFFFFF     # VFR_KEY_EMPTY: (+inf, max id)


def topk_pack_keys(dist: torch.Tensor, idx: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """(dist, idx) lists -> int64 exchange keys of the same shape (idx < 0 -> KEY_EMPTY)."""
    d, i = _dev(dist, torch.float32, "dist"), _dev(idx, torch.int64, "idx")
    keys = out if out is not None else torch.empty(d.shape, dtype=torch.int64, device=d.device)
    if out is not None:
        _dev(out, torch.int64, "out")
    _check(lib().vfr_topk_pack_keys(d.data_ptr(), i.data_ptr(), d.numel(), keys.data_ptr(), _stream()), "vfr_topk_pack_keys")
    return keys


def topk_merge_keys(part_keys: torch.Tensor, want_lists: bool = True, want_keys: bool = False):
    """[G,Nq,k] int64 key lists -> (dist [Nq,k] | None, idx [Nq,k] | None, keys [Nq,k] | None).  The slots may be views into
    wider rows (each [Nq,k] dense, any slot stride >= Nq*k): the lists inside a packed exchange buffer are merged in place."""
    pk = part_keys
    G, Nq, k = pk.shape
    if not (pk.is_cuda and pk.dtype == torch.int64 and pk.stride(2) == 1 and pk.stride(1) == k and pk.stride(0) >= Nq * k):
        pk = _dev(part_keys, torch.int64, "part_keys")
    od = torch.empty((Nq, k), dtype=torch.float32, device=pk.device) if want_lists else None
    oi = torch.empty((Nq, k), dtype=torch.int64, device=pk.device) if want_lists else None
    ok = torch.empty((Nq, k), dtype=torch.int64, device=pk.device) if want_keys else None
    _check(lib().vfr_topk_merge_keys_strided(pk.data_ptr(), pk.stride(0) if G > 1 else Nq * k, G, Nq, k, _ptr(od), _ptr(oi), _ptr(ok),
                                             _stream()), "vfr_topk_merge_keys")
    return od, oi, ok


def gt_best_keys(own_scores: torch.Tensor, labels: torch.Tensor, id_base: torch.Tensor, sel: torch.Tensor, Nq: int):
    """own_scores [n_sel, Ms] (``score_own``), labels bool/uint8 [R, n_sel, Ml], id_base / sel int64 [n_sel] ->
    keys int64 [R, Nq]: best positive (score, id) per threshold and query, KEY_EMPTY elsewhere."""
    sc = _dev(own_scores, torch.float32, "own_scores")
    lab = labels.view(torch.uint8) if labels.dtype == torch.bool else labels
    lab = _dev(lab, torch.uint8, "labels")
    base, sel = _dev(id_base, torch.int64, "id_base"), _dev(sel, torch.int64, "sel")
    R, n_sel, Ml = lab.shape
    if sc.shape[0] != n_sel or base.numel() != n_sel or sel.numel() != n_sel:
        raise RuntimeError("gt_best_keys: inconsistent shapes")
    keys = torch.empty((R, Nq), dtype=torch.int64, device=sc.device)
    _check(lib().vfr_gt_best_keys_f32(sc.data_ptr(), n_sel, min(sc.shape[1], Ml), sc.shape[1], lab.data_ptr(), R, Ml,
                                      base.data_ptr(), sel.data_ptr(), Nq, keys.data_ptr(), _stream()), "vfr_gt_best_keys_f32")
    return keys


def gt_rank_keys(own_scores: torch.Tensor, labels: torch.Tensor, id_base: torch.Tensor, sel: torch.Tensor, Nq: int):
    """``gt_best_keys`` plus, from the same two launches: the keys unpacked as (rank_dist f32 [R, Nq], rank_idx int64 [R, Nq]), a
    zeroed count buffer int64 [R, Nq] and the device flag (int32 [1]) "some selected query has no positive moment"."""
    sc = _dev(own_scores, torch.float32, "own_scores")
    lab = labels.view(torch.uint8) if labels.dtype == torch.bool else labels
    lab = _dev(lab, torch.uint8, "labels")
    base, sel = _dev(id_base, torch.int64, "id_base"), _dev(sel, torch.int64, "sel")
    R, n_sel, Ml = lab.shape
    if sc.shape[0] != n_sel or base.numel() != n_sel or sel.numel() != n_sel:
        raise RuntimeError("gt_rank_keys: inconsistent shapes")
    keys = torch.empty((R, Nq), dtype=torch.int64, device=sc.device)
    rd = torch.empty((R, Nq), dtype=torch.float32, device=sc.device)
    ri = torch.empty((R, Nq), dtype=torch.int64, device=sc.device)
    cnt = torch.empty((R, Nq), dtype=torch.int64, device=sc.device)
    missing = torch.empty((1,), dtype=torch.int32, device=sc.device)
    _check(lib().vfr_gt_rank_keys_f32(sc.data_ptr(), n_sel, min(sc.shape[1], Ml), sc.shape[1], lab.data_ptr(), R, Ml, base.data_ptr(),
                                      sel.data_ptr(), Nq, keys.data_ptr(), rd.data_ptr(), ri.data_ptr(), cnt.data_ptr(), missing.data_ptr(),
                                      _stream()), "vfr_gt_rank_keys_f32")
    return keys, rd, ri, cnt, missing


def gt_labels(times: torch.Tensor, nannot: torch.Tensor, n_own: torch.Tensor, thresholds, strict: bool, Mmax: int):
    """a11 on the device: times int32 [Nq, A, 2], nannot / n_own int32 [Nq] -> labels bool [R, Nq, Mmax]."""
    t, na, no = _dev(times, torch.int32, "times"), _dev(nannot, torch.int32, "nannot"), _dev(n_own, torch.int32, "n_own")
    Nq, A = t.shape[0], t.shape[1]
    R = len(thresholds)
    thr = (ctypes.c_double * R)(*[float(x) for x in thresholds])
    lab = torch.empty((R, Nq, Mmax), dtype=torch.uint8, device=t.device)
    _check(lib().vfr_gt_labels_u8(t.data_ptr(), na.data_ptr(), no.data_ptr(), Nq, A, thr, R, int(bool(strict)), Mmax,
                                  lab.data_ptr(), _stream()), "vfr_gt_labels_u8")
    return lab.view(torch.bool)


def frames_normalize(frames_thwc: torch.Tensor) -> torch.Tensor:
    fr = _dev(frames_thwc, torch.uint8, "frames")
    T, H, W, _ = fr.shape
    out = torch.empty((T, 3, H, W), dtype=torch.float32, device=fr.device)
    _check(lib().vfr_frames_normalize_f32(fr.data_ptr(), T, H, W, out.data_ptr(), _stream()), "vfr_frames_normalize_f32")
    return out


def conv3x3_relu(x, w, b) -> torch.Tensor:
    x, w, b = _dev(x, torch.float32, "x"), _dev(w, torch.float32, "w"), _dev(b, torch.float32, "b")
    B, Cin, H, W = x.shape
    y = torch.empty((B, w.shape[0], H, W), dtype=torch.float32, device=x.device)
    _check(lib().vfr_conv3x3_relu_f32(x.data_ptr(), B, Cin, H, W, w.data_ptr(), b.data_ptr(), w.shape[0], y.data_ptr(),
                                      _stream()), "vfr_conv3x3_relu_f32")
    return y


def maxpool2(x) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _check(lib().vfr_maxpool2_f32(x.data_ptr(), B, C, H, W, y.data_ptr(), _stream()), "vfr_maxpool2_f32")
    return y


def adaptive_avgpool7(x) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, C, 7, 7), dtype=torch.float32, device=x.device)
    _check(lib().vfr_adaptive_avgpool7_f32(x.data_ptr(), B, C, H, W, y.data_ptr(), _stream()), "vfr_adaptive_avgpool7_f32")
    return y


def vgg_fc7(frames_thwc, cfg, conv_w, conv_b, fc6, fc7) -> torch.Tensor:
    """uint8 [T,H,W,3] -> fc7 features [T, fc_dim]  (get_rgb_features.py:64-69,122-126,145-147)."""
    fr = _dev(frames_thwc, torch.uint8, "frames")
    T, H, W, _ = fr.shape
    cfg_i = (ctypes.c_int * len(cfg))(*[0 if c == "M" else int(c) for c in cfg])
    cw = [_dev(w, torch.float32, "conv_w") for w in conv_w]
    cb = [_dev(b, torch.float32, "conv_b") for b in conv_b]
    wp = (ctypes.c_void_p * len(cw))(*[w.data_ptr() for w in cw])
    bp = (ctypes.c_void_p * len(cb))(*[b.data_ptr() for b in cb])
    f6w, f6b = _dev(fc6[0], torch.float32, "fc6_w"), _dev(fc6[1], torch.float32, "fc6_b")
    f7w, f7b = _dev(fc7[0], torch.float32, "fc7_w"), _dev(fc7[1], torch.float32, "fc7_b")
    fc_dim = f6w.shape[0]
    out = torch.empty((T, fc_dim), dtype=torch.float32, device=fr.device)
    nbytes = lib().vfr_vgg_fc7_workspace_bytes(T, H, W, ctypes.cast(cfg_i, ctypes.c_void_p), len(cfg), fc_dim)
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=fr.device)
    _check(lib().vfr_vgg_fc7_f32(fr.data_ptr(), T, H, W, ctypes.cast(cfg_i, ctypes.c_void_p), len(cfg),
                                 ctypes.cast(wp, ctypes.c_void_p), ctypes.cast(bp, ctypes.c_void_p), f6w.data_ptr(),
                                 f6b.data_ptr(), f7w.data_ptr(), f7b.data_ptr(), fc_dim, out.data_ptr(), ws.data_ptr(),
                                 nbytes, _stream()), "vfr_vgg_fc7_f32")
    return out


RESNET152_BLOCKS = (3, 8, 36, 3)


def resnet_conv_plan(blocks=RESNET152_BLOCKS, width: int = 64):
    """The convolutions of a torchvision Bottleneck ResNet in execution order: (conv prefix, bn prefix) of the state dict --
    the stem, then per block conv1, conv2, conv3 and, for the first block of a layer, downsample.0 / downsample.1."""
    plan = [("conv1", "bn1")]
    for li, nb in enumerate(blocks):
        for b in range(nb):
            pre = f"layer{li + 1}.{b}"
            plan += [(f"{pre}.conv1", f"{pre}.bn1"), (f"{pre}.conv2", f"{pre}.bn2"), (f"{pre}.conv3", f"{pre}.bn3")]
            if b == 0:
                plan.append((f"{pre}.downsample.0", f"{pre}.downsample.1"))
    return plan


def resnet_pack(state_dict, blocks=RESNET152_BLOCKS, width: int = 64, device="cuda:0"):
    """torchvision-style resnet state dict (tensors or numpy) -> (conv weights, packed BatchNorm tensors [4, C]) on the device,
    in the order ``vfr_resnet_pool_f32`` takes them.  Done once per model; the result is passed to ``resnet_pool``."""
    def t(x):
        return torch.as_tensor(x, dtype=torch.float32).to(device).contiguous()
    convs, bns = [], []
    for conv, bn in resnet_conv_plan(blocks, width):
        convs.append(t(state_dict[conv + ".weight"]))
        bns.append(torch.stack([t(state_dict[bn + ".weight"]), t(state_dict[bn + ".bias"]), t(state_dict[bn + ".running_mean"]),
                                t(state_dict[bn + ".running_var"])]).contiguous())
    return ResnetPacked((convs, bns))


class ResnetPacked(tuple):
    """``(conv weights, packed BatchNorm tensors)`` as ``resnet_pack`` returns them, plus the folded form ``resnet_pool`` computes
    once per (blocks, width, eps) and keeps here (``vfr_resnet_fold_f32``): a plain tuple everywhere else."""
    def __new__(cls, pair):
        self = super().__new__(cls, pair)
        self.folded = {}
        return self


def resnet_pool(frames_thwc, packed, blocks=RESNET152_BLOCKS, width: int = 64, eps: float = 1e-5) -> torch.Tensor:
    """uint8 [T,H,W,3] -> pooled ResNet features [T, 32 * width]  (get_rgb_features.py:64-69,127-131, 147)."""
    fr = _dev(frames_thwc, torch.uint8, "frames")
    T, H, W, _ = fr.shape
    convs, bns = packed
    convs = [_dev(w, torch.float32, "conv_w") for w in convs]
    bns = [_dev(b, torch.float32, "bn") for b in bns]
    nconv = 1 + sum(3 * n + 1 for n in blocks)
    if len(convs) != nconv or len(bns) != nconv:
        raise RuntimeError(f"resnet_pool: expected {nconv} convolutions for blocks {tuple(blocks)}, got {len(convs)}")
    bl = (ctypes.c_int * 4)(*[int(b) for b in blocks])
    wp = (ctypes.c_void_p * nconv)(*[w.data_ptr() for w in convs])
    bp = (ctypes.c_void_p * nconv)(*[b.data_ptr() for b in bns])
    out = torch.empty((T, 32 * width), dtype=torch.float32, device=fr.device)
    nbytes = lib().vfr_resnet_pool_workspace_bytes(T, H, W, ctypes.cast(bl, ctypes.c_void_p), width)
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=fr.device)
    if isinstance(packed, ResnetPacked):
        # the model's weights folded once (BatchNorm into the convolutions, tap-major), reused by every later call
        key = (tuple(int(b) for b in blocks), int(width), float(eps), fr.device)
        folded = packed.folded.get(key)
        if folded is None:
            fb = lib().vfr_resnet_folded_bytes(ctypes.cast(bl, ctypes.c_void_p), width)
            folded = torch.empty((max(fb, 1),), dtype=torch.uint8, device=fr.device)
            _check(lib().vfr_resnet_fold_f32(ctypes.cast(bl, ctypes.c_void_p), width, ctypes.cast(wp, ctypes.c_void_p),
                                             ctypes.cast(bp, ctypes.c_void_p), eps, folded.data_ptr(), folded.numel(), _stream()),
                   "vfr_resnet_fold_f32")
            packed.folded[key] = folded
        _check(lib().vfr_resnet_pool_folded_f32(fr.data_ptr(), T, H, W, ctypes.cast(bl, ctypes.c_void_p), width, folded.data_ptr(),
                                                folded.numel(), out.data_ptr(), ws.data_ptr(), nbytes, _stream()),
               "vfr_resnet_pool_folded_f32")
        return out
    _check(lib().vfr_resnet_pool_f32(fr.data_ptr(), T, H, W, ctypes.cast(bl, ctypes.c_void_p), width, ctypes.cast(wp, ctypes.c_void_p),
                                     ctypes.cast(bp, ctypes.c_void_p), eps, out.data_ptr(), ws.data_ptr(), nbytes, _stream()),
           "vfr_resnet_pool_f32")
    return out


def transpose(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    r, c = x.shape
    out = torch.empty((c, r), dtype=torch.float32, device=x.device)
    _check(lib().vfr_transpose_f32(x.data_ptr(), r, c, out.data_ptr(), _stream()), "vfr_transpose_f32")
    return out


def colsum(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    _check(lib().vfr_colsum_f32(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream()), "vfr_colsum_f32")
    return out


def relu_backward(grad: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    grad, act = _dev(grad, torch.float32, "grad"), _dev(act, torch.float32, "act")
    out = torch.empty_like(grad)
    _check(lib().vfr_relu_backward_f32(grad.data_ptr(), act.data_ptr(), grad.numel(), out.data_ptr(), _stream()), "vfr_relu_backward_f32")
    return out


def lstm_cell_forward(pre, xproj_t, x_stride, c_prev, gates_out, c_out, h_out):
    """pre [B,4H]; xproj_t = view whose row b starts b * x_stride floats after its first element."""
    B, H = c_prev.shape
    _check(lib().vfr_lstm_cell_forward_f32(pre.data_ptr(), xproj_t.data_ptr(), x_stride, c_prev.data_ptr(), B, H, gates_out.data_ptr(),
                                           c_out.data_ptr(), h_out.data_ptr(), _stream()), "vfr_lstm_cell_forward_f32")


def lstm_cell_backward(dh, dc, gates, c_prev, c_cur, dpre_out):
    B, H = c_prev.shape
    _check(lib().vfr_lstm_cell_backward_f32(dh.data_ptr(), dc.data_ptr(), gates.data_ptr(), c_prev.data_ptr(), c_cur.data_ptr(), B, H,
                                            dpre_out.data_ptr(), _stream()), "vfr_lstm_cell_backward_f32")


def bilstm_train_forward(x, ws):
    """x [B, T, E] f32, ws = the eight nn.LSTM tensors (forward four, reverse four) -> (gates [2,T,B,4H], cs, hs [2,T+1,B,H])."""
    x = _dev(x, torch.float32, "x")
    ws = [_dev(w, torch.float32, "lstm weight") for w in ws]
    B, T, E = x.shape
    H = ws[1].shape[1]
    gates = torch.empty((2, T, B, 4 * H), dtype=torch.float32, device=x.device)
    cs = torch.empty((2, T + 1, B, H), dtype=torch.float32, device=x.device)
    hs = torch.empty((2, T + 1, B, H), dtype=torch.float32, device=x.device)
    _check(lib().vfr_bilstm_train_forward_f32(x.data_ptr(), B, T, E, H, *[w.data_ptr() for w in ws], gates.data_ptr(), cs.data_ptr(),
                                              hs.data_ptr(), _stream()), "vfr_bilstm_train_forward_f32")
    return gates, cs, hs


def bilstm_train_backward(gout, gates, cs, whhT_f, whhT_b):
    """gout [B, 2H] -> dpre [2, T, B, 4H] (gradient of every step's gate pre-activations, step-major per direction)."""
    gout = _dev(gout, torch.float32, "gout")
    _, T, B, G = gates.shape
    H = G // 4
    nbytes = lib().vfr_bilstm_train_backward_workspace_bytes(B, H)
    wsb = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=gout.device)
    dpre = torch.empty_like(gates)
    _check(lib().vfr_bilstm_train_backward_f32(gout.data_ptr(), gates.data_ptr(), cs.data_ptr(), whhT_f.data_ptr(), whhT_b.data_ptr(),
                                               B, T, H, dpre.data_ptr(), wsb.data_ptr(), nbytes, _stream()),
           "vfr_bilstm_train_backward_f32")
    return dpre


def ranking_loss_forward(posit, intra, inter, lang, maskp, maskn, n_samples: int, b: float, lamb: float, eps: float = 1e-6):
    """-> (loss [1] f32, workspace) ; the workspace (row distances, per-sample terms) feeds ranking_loss_backward."""
    posit, intra, inter, lang = (_dev(x, torch.float32, n) for x, n in ((posit, "posit"), (intra, "intra"), (inter, "inter"),
                                                                         (lang, "lang")))
    maskp, maskn = _dev(maskp, torch.int64, "maskp"), _dev(maskn, torch.int64, "maskn")
    P, Nn, D = posit.shape[0], intra.shape[0], posit.shape[1]
    nbytes = lib().vfr_ranking_loss_workspace_bytes(P, Nn, n_samples)
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=posit.device)
    loss = torch.empty((1,), dtype=torch.float32, device=posit.device)
    _check(lib().vfr_ranking_loss_f32(posit.data_ptr(), intra.data_ptr(), inter.data_ptr(), lang.data_ptr(), maskp.data_ptr(),
                                      maskn.data_ptr(), P, Nn, n_samples, D, b, lamb, eps, loss.data_ptr(), ws.data_ptr(),
                                      nbytes, _stream()), "vfr_ranking_loss_f32")
    return loss, ws


def ranking_loss_backward(posit, intra, inter, lang, maskp, maskn, n_samples: int, lamb: float, grad_loss, ws,
                          eps: float = 1e-6):
    posit, intra, inter, lang = (_dev(x, torch.float32, n) for x, n in ((posit, "posit"), (intra, "intra"), (inter, "inter"),
                                                                         (lang, "lang")))
    maskp, maskn = _dev(maskp, torch.int64, "maskp"), _dev(maskn, torch.int64, "maskn")
    g = _dev(grad_loss.reshape(1), torch.float32, "grad_loss")
    outs = [torch.empty_like(x) for x in (posit, intra, inter, lang)]
    _check(lib().vfr_ranking_loss_grad_f32(posit.data_ptr(), intra.data_ptr(), inter.data_ptr(), lang.data_ptr(),
                                           maskp.data_ptr(), maskn.data_ptr(), posit.shape[0], intra.shape[0], n_samples,
                                           posit.shape[1], lamb, eps, g.data_ptr(), ws.data_ptr(), *[o.data_ptr() for o in outs],
                                           _stream()), "vfr_ranking_loss_grad_f32")
    return outs
