"""The encoders of ``CALModel`` in training mode on a ROCm device: forward AND backward in HIP kernels
(second half -- ``loss.backward()`` of ``Trainer.train_epoch``, ``model/main.py:58-67``).

Three ``torch.autograd.Function``s; autograd is plumbing (it calls ``backward`` in the right order and owns the
``.grad`` accumulation), every contraction is the chain GEMM behind ``vfr_linear_f32`` and every elementwise step one
of the kernels in ``csrc/train.hip``:

* ``linear(x, W, b)``                       -- ``nn.Linear`` (``lang_fc``, the BERT projection): dX = dY W, dW = dY^T X, db = colsum dY
* ``visual_mlp(x, W1, b1, W2, b2)``         -- ``visual_fc[0..2]``: Linear / ReLU / Linear with the hidden layer saved
* ``bilstm_final(x, lstm weights, H)``      -- ``nn.LSTM(bidirectional)`` over the embedded words, returning h_n [B, 2H]:
  forward = T launches of the inference path's fused step for both directions (``[x_t | h] [W_ih | W_hh]^T`` + cell in one
  kernel, here also keeping the gates and the states of all T steps); backward = T x (one cell-backward launch, one
  split-K GEMM grid ``dh = dpre W_hh``, both directions each) and then ONE GEMM per weight over all steps,
  dW = DP^T [h | x].  E or H not a multiple of 4: the same recurrence step by step from Python (unfused kernels).

``dX = dY W`` is ``linear(dY, W^T)``; ``dW = dY^T X`` is ``linear(dY^T, X^T)`` (the GEMM contracts the trailing dimension of
both operands), hence the transposes.  The dropout after ``visual_fc[2]`` stays ``torch.nn.functional.dropout`` (its mask
has to come from torch's generator to follow the reference's RNG stream).
"""
from __future__ import annotations

import torch

from . import _vfr


def _lin(a, w, b=None, relu=False):
    return _vfr.linear(a.contiguous(), w.contiguous(), b, relu)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.save_for_backward(x2, W)
        ctx.shape, ctx.has_b = x.shape, b is not None
        return _lin(x2, W, b).reshape(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, W = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1]).contiguous()
        gx = _lin(g, _vfr.transpose(W)).reshape(ctx.shape) if ctx.needs_input_grad[0] else None
        gW = _lin(_vfr.transpose(g), _vfr.transpose(x2)) if ctx.needs_input_grad[1] else None
        gb = _vfr.colsum(g) if ctx.has_b and ctx.needs_input_grad[2] else None
        return gx, gW, gb


class _VisualMLPFn(torch.autograd.Function):
    """``visual_fc``: Linear(2F+2, 500) -> ReLU -> Linear(500, emb) (``model/models.py:21-26``)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        h = _lin(x2, W1, b1, relu=True)
        ctx.save_for_backward(x2, h, W1, W2)
        ctx.shape = x.shape
        return _lin(h, W2, b2).reshape(*x.shape[:-1], W2.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, h, W1, W2 = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1]).contiguous()
        gT = _vfr.transpose(g)
        gW2 = _lin(gT, _vfr.transpose(h))                                   # [emb, 500]
        gb2 = _vfr.colsum(g)
        gh = _vfr.relu_backward(_lin(g, _vfr.transpose(W2)), h)             # [M, 500]
        ghT = _vfr.transpose(gh)
        gW1 = _lin(ghT, _vfr.transpose(x2))                                 # [500, 2F+2]
        gb1 = _vfr.colsum(gh)
        gx = _lin(gh, _vfr.transpose(W1)).reshape(ctx.shape) if ctx.needs_input_grad[0] else None
        return gx, gW1, gb1, gW2, gb2


class _BiLSTMFn(torch.autograd.Function):
    """x [B, T, E] -> h_n [B, 2H] = [forward final | reverse final], zero initial state (``model/models.py:50-52,65``).

    Weights in ``nn.LSTM`` order: (W_ih, W_hh, b_ih, b_hh) forward, then the four ``_reverse`` ones."""

    @staticmethod
    def forward(ctx, x, *ws):
        B, T, E = x.shape
        H = ws[1].shape[1]
        dev = x.device
        x2 = x.contiguous().reshape(B * T, E)
        ctx.dims = (B, T, E, H)
        ctx.fused = E % 4 == 0 and H % 4 == 0
        if ctx.fused:
            gates, cs, hs = _vfr.bilstm_train_forward(x2.reshape(B, T, E), [w.contiguous() for w in ws])
            ctx.save_for_backward(x2, *ws, gates, cs, hs)
            return torch.cat([hs[0, T], hs[1, T]], dim=1)
        saved = []
        out = torch.empty((B, 2 * H), dtype=torch.float32, device=dev)
        for d in range(2):
            W_ih, W_hh, b_ih, b_hh = ws[4 * d:4 * d + 4]
            xproj = _lin(x2, W_ih, b_ih)                                    # [B*T, 4H], row (b, t) at b*T + t
            gates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
            cs = torch.zeros((T + 1, B, H), dtype=torch.float32, device=dev)
            hs = torch.zeros((T + 1, B, H), dtype=torch.float32, device=dev)
            for s in range(T):
                t = s if d == 0 else T - 1 - s
                pre = _lin(hs[s], W_hh, b_hh)
                _vfr.lstm_cell_forward(pre, xproj[t:], T * 4 * H, cs[s], gates[s], cs[s + 1], hs[s + 1])
            out[:, d * H:(d + 1) * H] = hs[T]
            saved += [gates, cs, hs]
        ctx.save_for_backward(x2, *ws, *saved)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, T, E, H = ctx.dims
        st = ctx.saved_tensors
        x2, ws, saved = st[0], st[1:9], st[9:]
        dev = x2.device
        gout = gout.contiguous()
        need_x = ctx.needs_input_grad[0]
        gx = torch.zeros((B, T, E), dtype=torch.float32, device=dev) if need_x else None
        grads = []
        x_tm = x2.reshape(B, T, E).transpose(0, 1).contiguous()              # [T, B, E] time-major
        W_hhT = [_vfr.transpose(ws[1].contiguous()), _vfr.transpose(ws[5].contiguous())]     # [H, 4H]
        if ctx.fused:
            gates2, cs2, hs2 = saved
            DP2 = _vfr.bilstm_train_backward(gout, gates2, cs2, W_hhT[0], W_hhT[1])
        for d in range(2):
            W_ih = ws[4 * d]
            W_ihT = _vfr.transpose(W_ih) if need_x else None                # [E, 4H]
            if ctx.fused:
                hs, DP = hs2[d], DP2[d]
            else:
                gates, cs, hs = saved[3 * d:3 * d + 3]
                dh = gout[:, d * H:(d + 1) * H].contiguous()
                dc = torch.zeros((B, H), dtype=torch.float32, device=dev)
                DP = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
                for s in range(T - 1, -1, -1):
                    _vfr.lstm_cell_backward(dh, dc, gates[s], cs[s], cs[s + 1], DP[s])
                    if s > 0:
                        dh = _lin(DP[s], W_hhT[d])                           # gradient reaching h of step s - 1
            DPf = DP.reshape(T * B, 4 * H)
            DPT = _vfr.transpose(DPf)                                        # [4H, T*B]
            # inputs in STEP order: step s of the reverse direction read time T-1-s
            xs = x_tm if d == 0 else torch.flip(x_tm, dims=[0])
            gW_hh = _lin(DPT, _vfr.transpose(hs[:T].reshape(T * B, H)))      # [4H, H]
            gW_ih = _lin(DPT, _vfr.transpose(xs.reshape(T * B, E)))          # [4H, E]
            gb = _vfr.colsum(DPf)
            grads += [gW_ih, gW_hh, gb, gb.clone()]
            if need_x:
                gxs = _lin(DPf, W_ihT).reshape(T, B, E)                      # step-major
                if d == 1:
                    gxs = torch.flip(gxs, dims=[0])
                gx += gxs.transpose(0, 1)
        return (gx, *grads)


def linear(x, W, b=None):
    return _LinearFn.apply(x, W, b)


def visual_mlp(x, W1, b1, W2, b2):
    return _VisualMLPFn.apply(x, W1, b1, W2, b2)


def bilstm_final(x, lstm: torch.nn.LSTM):
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
    ws = [getattr(lstm, n) for n in names] + [getattr(lstm, n + "_reverse") for n in names]
    return _BiLSTMFn.apply(x, *ws)
