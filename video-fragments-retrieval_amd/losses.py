"""The triplet ranking loss of the reference's training / test loops (``Trainer.ranking_loss``, ``model/main.py:214-232``).

``ranking_loss(posit_emb, intra_emb, inter_emb, lang_emb, maskp, maskn, b=0.1, lamb=0.4, normalize_loss=False)`` takes the
method's arguments (the three Trainer attributes it reads become keywords) and returns the same ``(loss, n_samples)``.
On a ROCm device the per-sample Python loop -- three boolean-mask gathers and three ``pairwise_distance`` calls per sample --
is two HIP launches forward and two backward (``csrc/loss.hip``), wired into autograd so ``loss.backward()`` reaches the
encoders exactly as in ``train_epoch`` (``main.py:63-66``).  CPU tensors take the reference's formula with torch ops.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class _RankingLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, posit, intra, inter, lang, maskp, maskn, n_samples, b, lamb):
        from . import _vfr
        loss, ws = _vfr.ranking_loss_forward(posit, intra, inter, lang, maskp, maskn, n_samples, b, lamb)
        ctx.save_for_backward(posit, intra, inter, lang, maskp, maskn, ws)
        ctx.n_samples, ctx.lamb = n_samples, lamb
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        from . import _vfr
        posit, intra, inter, lang, maskp, maskn, ws = ctx.saved_tensors
        gp, gn, gi, gl = _vfr.ranking_loss_backward(posit, intra, inter, lang, maskp, maskn, ctx.n_samples, ctx.lamb,
                                                    grad_out.contiguous(), ws)
        return gp, gn, gi, gl, None, None, None, None, None


def _normalize(x):
    return x.div(x.norm(dim=1, keepdim=True) + 1e-5)                      # main.py:219-223


def ranking_loss(posit_emb, intra_emb, inter_emb, lang_emb, maskp, maskn, b=0.1, lamb=0.4, normalize_loss=False):
    n_samples = int(maskp.max().item()) + 1                                 # main.py:217 (the same host sync)
    if normalize_loss:
        posit_emb, intra_emb, inter_emb, lang_emb = (_normalize(x) for x in (posit_emb, intra_emb, inter_emb, lang_emb))
    if posit_emb.is_cuda:
        loss = _RankingLossFn.apply(posit_emb.contiguous(), intra_emb.contiguous(), inter_emb.contiguous(),
                                    lang_emb.contiguous(), maskp.contiguous(), maskn.contiguous(), n_samples, float(b), float(lamb))
        return loss, n_samples
    loss = 0
    for i in range(n_samples):                                              # main.py:225-231
        mp, mn = maskp == i, maskn == i
        c_posit = F.pairwise_distance(posit_emb[mp], lang_emb[i].repeat(int(mp.sum()), 1)).mean()
        c_intra = F.pairwise_distance(intra_emb[mn], lang_emb[i].repeat(int(mn.sum()), 1)).mean()
        c_inter = F.pairwise_distance(inter_emb[mp], lang_emb[i].repeat(int(mp.sum()), 1)).mean()
        loss = loss + (F.relu(c_posit - c_intra + b) + lamb * F.relu(c_posit - c_inter + b))
    return loss, n_samples
