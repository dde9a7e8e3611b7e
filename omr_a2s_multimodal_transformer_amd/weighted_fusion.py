"""Weighted late fusion (reference: src/multimodal/weighted_multimodal/test.py:21-70): an image model and an audio model --
two unimodal `Transformer`s with the same vocabulary -- decode in lock-step from the same prefix; at every step the two
next-token distributions are mixed, alpha * softmax(image logits) + (1 - alpha) * softmax(audio logits), and the argmax token
is fed back to BOTH decoders.

The reference re-runs both decoders over the whole prefix per token; here each model keeps its KV cache
(Decoder.init_decode / decode_step: cross-attention K|V projected once, self-attention K|V appended per step) and the mixing +
argmax of a step is one kernel (omr_weighted_argmax).  Same tokens as the reference (tests/golden/f15_weighted.npz).
"""
from __future__ import annotations

from typing import List

import torch

from . import kernels as K
from .synthetic import EOS_TOKEN, SOS_TOKEN


@torch.no_grad()
def weighted_prediction(xi: torch.Tensor, xa: torch.Tensor, img_model, audio_model, alpha: float = 0.5) -> List[str]:
    """weighted_multimodal/test.py:21-70, same signature and return value (the predicted words, <eos> included when reached).
    Like the reference, the loop runs for max(img_model.max_seq_len, audio_model.max_seq_len) steps and a model whose
    positional table is shorter than that raises when the sequence outgrows it."""
    assert xi.size(0) == 1, "Inference only supports batch_size = 1"
    mem_i = img_model.encode(xi)                       # encoder -> 2-D PE -> flatten (test.py:28-37)
    mem_a = audio_model.encode(xa)
    st_i = img_model.decoder.init_decode(mem_i)
    st_a = audio_model.decoder.init_decode(mem_a)
    tok = torch.full((1, 1), img_model.w2i[SOS_TOKEN], dtype=torch.int64, device=mem_i.device)
    yhat: List[str] = []
    for _ in range(max(img_model.max_seq_len, audio_model.max_seq_len)):
        li = img_model.decoder.decode_step(tok, st_i).contiguous()          # fp32 logits of the last position [V]
        la = audio_model.decoder.decode_step(tok, st_a).contiguous()
        idx, _ = K.weighted_argmax(li, la, alpha)
        word = img_model._i2w(int(idx.item()))          # both models share the vocabulary (test.py:62)
        yhat.append(word)
        if word == EOS_TOKEN:
            break
        tok = idx.view(1, 1)
    return yhat
