"""``models.CALModel`` with the reference's constructor, sub-module tree and ``state_dict`` keys
(``model/models.py:7-68``), so ``main.py`` and existing ``last.pth`` checkpoints keep working.

Dispatch inside ``forward``:

* ``model.eval()`` under ``torch.no_grad()`` with the input on a ROCm device  ->  the hand-written gfx950
  kernels (``_vfr``): chain-GEMM clip MLP, gather + BiLSTM + ``lang_fc`` query encoder.  If ``libvfr.so`` is
  missing this raises -- there is no silent substitute for the HIP path.
* training mode or grad enabled, input on a ROCm device  ->  the same arithmetic as ``torch.autograd.Function``s whose
  forward AND backward are HIP kernels (``train.py``: chain GEMMs + ``csrc/train.hip``), so ``loss.backward()`` of
  ``main.py:66`` runs on our kernels too; ``models.HIP_TRAINING = False`` switches this path back to the ``torch.nn``
  sub-modules (A/B checks).
* input on the CPU  ->  the same ``torch.nn`` sub-modules (an ``nn.Module`` has to run where its tensors
  live; BASELINE.md C1 is a CPU run).

``encode_clips`` / ``encode_queries`` are the batched entry points the evaluators use: a whole corpus /
a whole query batch per call, with the factored clip encoder that never materialises the ``[n, 2F+2]``
concat of ``data.make_visual_features``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .data import EMBEDDING_DIM

_LSTM_KEYS = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
HIP_TRAINING = True      # grad-enabled / training-mode forward on a ROCm device through the HIP autograd functions (train.py)


def init_weights(m):
    """U(-0.08, 0.08) weights and zero bias for every ``nn.Linear`` (applied via ``Module.apply``)."""
    if type(m) == nn.Linear:
        nn.init.uniform_(m.weight, -0.08, 0.08)
        nn.init.constant_(m.bias, 0)


class CALModel(nn.Module):
    def __init__(self, visual_input_dim, pretrained_emb=None, emb_dim=EMBEDDING_DIM, hidden_size=1000, bert_emb=768,
                 dropout_rate=0.3, normalize_lang=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.normalize_lang = normalize_lang

        # clip branch: Linear(2F+2 -> 500) / ReLU / Linear(500 -> emb) / Dropout   (indices 0..3 as in the reference)
        self.visual_fc = nn.Sequential(nn.Linear(visual_input_dim, 500), nn.ReLU(), nn.Linear(500, emb_dim),
                                       nn.Dropout(p=dropout_rate))
        self.visual_fc.apply(init_weights)

        if pretrained_emb is None:
            # BERT branch: a single projection of the pooled 768-d vector (default torch init, as in the reference)
            self.lang_fc = nn.Linear(bert_emb, emb_dim)
        else:
            self.word_embedding = nn.Embedding.from_pretrained(pretrained_emb, freeze=True, padding_idx=0)
            if normalize_lang:
                self.learnable_length = nn.Embedding.from_pretrained(torch.ones(pretrained_emb.size(0), 1),
                                                                     freeze=False, padding_idx=0)
            self.lstm = nn.LSTM(input_size=pretrained_emb.size(1), hidden_size=hidden_size, num_layers=1,
                                batch_first=True, bidirectional=True)
            self.lang_fc = nn.Linear(2 * hidden_size, emb_dim)
            self.lang_fc.apply(init_weights)

    # ------------------------------------------------------------------------------------------
    def init_hidden(self, batch_size, device):
        zeros = lambda: torch.zeros(2, batch_size, self.hidden_size, device=device)
        return [zeros(), zeros()]

    def _use_hip(self, t: torch.Tensor) -> bool:
        return t.is_cuda and not self.training and not torch.is_grad_enabled()

    def _use_hip_train(self, t: torch.Tensor) -> bool:
        return t.is_cuda and HIP_TRAINING and (self.training or torch.is_grad_enabled())

    def _lstm_weights(self):
        sd = {k: getattr(self.lstm, k) for k in _LSTM_KEYS}
        sd.update({k + "_reverse": getattr(self.lstm, k + "_reverse") for k in _LSTM_KEYS})
        return {k: v.detach() for k, v in sd.items()}

    # ------------------------------------------------------------------------------------------
    def forward(self, batch, visual=True, device=None, bert=False):
        if visual:
            if self._use_hip(batch):
                from . import _vfr
                fc1, fc2 = self.visual_fc[0], self.visual_fc[2]
                x = batch.reshape(-1, batch.shape[-1]).float()
                h = _vfr.linear(x, fc1.weight.detach(), fc1.bias.detach(), relu=True)
                out = _vfr.linear(h, fc2.weight.detach(), fc2.bias.detach())
                return out.reshape(*batch.shape[:-1], out.shape[-1])
            if self._use_hip_train(batch):
                from . import train
                fc1, fc2 = self.visual_fc[0], self.visual_fc[2]
                return self.visual_fc[3](train.visual_mlp(batch.float(), fc1.weight, fc1.bias, fc2.weight, fc2.bias))
            return self.visual_fc(batch)
        if bert:
            if self._use_hip(batch):
                from . import _vfr
                return _vfr.linear(batch.float(), self.lang_fc.weight.detach(), self.lang_fc.bias.detach())
            if self._use_hip_train(batch):
                from . import train
                return train.linear(batch.float(), self.lang_fc.weight, self.lang_fc.bias)
            return self.lang_fc(batch)
        if self._use_hip(batch):
            return self.encode_queries(batch)
        embedded = self.word_embedding(batch)
        if self.normalize_lang:
            length = self.learnable_length(batch)
            embedded = embedded.div(embedded.norm(dim=-1, keepdim=True) + 1e-5) * length
        if self._use_hip_train(batch):
            from . import train
            return train.linear(train.bilstm_final(embedded, self.lstm), self.lang_fc.weight, self.lang_fc.bias)
        _, hidden = self.lstm(embedded, self.init_hidden(batch.size(0), device))
        h_n = hidden[0].transpose(0, 1).reshape(batch.size(0), 2 * self.hidden_size)
        return self.lang_fc(h_n)

    # ------------------------------------------------------------------------------------------
    # batched HIP entry points (no torch.nn arithmetic below)
    def encode_queries(self, tokens: torch.Tensor) -> torch.Tensor:
        """int64 [B, T] token ids on the device -> [B, emb_dim] query embeddings."""
        from . import _vfr
        len_tab = self.learnable_length.weight.detach() if self.normalize_lang else None
        return _vfr.bilstm_final(tokens, self.word_embedding.weight.detach(), self._lstm_weights(),
                                 self.lang_fc.weight.detach(), self.lang_fc.bias.detach(), len_tab)

    def encode_clips(self, seg: torch.Tensor, ctx: torch.Tensor, clip_off: torch.Tensor) -> torch.Tensor:
        """Packed clip features (``data.FeatureBank`` fields on the device) -> [sum n, emb_dim] clip embeddings."""
        from . import _vfr
        fc1, fc2 = self.visual_fc[0], self.visual_fc[2]
        return _vfr.visual_mlp(seg, ctx, clip_off, fc1.weight.detach(), fc1.bias.detach(), fc2.weight.detach(),
                               fc2.bias.detach())
