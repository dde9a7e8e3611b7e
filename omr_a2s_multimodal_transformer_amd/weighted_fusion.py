"""Weighted late fusion (reference: src/multimodal/weighted_multimodal/test.py:21-70): an image model and an audio model --
two unimodal `Transformer`s with the same vocabulary -- decode in lock-step from the same prefix; at every step the two
next-token distributions are mixed, alpha * softmax(image logits) + (1 - alpha) * softmax(audio logits), and the argmax token
is fed back to BOTH decoders.

The reference re-runs both decoders over the whole prefix per token and reads every token back; here each model keeps its KV
cache (Decoder.init_decode: cross-attention K|V projected once, self-attention K|V appended per step), the mixing + argmax of
a step is one kernel (omr_weighted_argmax) and a CHUNK of positions is one host call (omr_weighted_decode_steps: the picked
token reaches both models' next position through device memory; the host reads a chunk of tokens back at a time and cuts
the sequence after <eos>).  Same tokens as the reference (tests/golden/f15_weighted.npz).
"""
from __future__ import annotations

from typing import List

import ctypes

import torch

from ._lib import cur_stream, lib, ptr
from .synthetic import EOS_TOKEN, SOS_TOKEN


@torch.no_grad()
def weighted_prediction(xi: torch.Tensor, xa: torch.Tensor, img_model, audio_model, alpha: float = 0.5, chunk: int = 16) -> List[str]:
    """weighted_multimodal/test.py:21-70, same signature and return value (the predicted words, <eos> included when reached).
    Like the reference, the loop runs for max(img_model.max_seq_len, audio_model.max_seq_len) steps and a model whose
    positional table is shorter than that raises when the sequence outgrows it."""
    assert xi.size(0) == 1, "Inference only supports batch_size = 1"
    mem_i = img_model.encode(xi)                       # encoder -> 2-D PE -> flatten (test.py:28-37)
    mem_a = audio_model.encode(xa)
    st_i = img_model.decoder.init_decode(mem_i)
    st_a = audio_model.decoder.init_decode(mem_a)
    assert st_i.V == st_a.V, "both models share the vocabulary (test.py:62)"
    dev = mem_i.device
    tok = torch.full((1,), img_model.w2i[SOS_TOKEN], dtype=torch.int64, device=dev)
    yhat: List[str] = []
    left = max(img_model.max_seq_len, audio_model.max_seq_len)
    while left > 0:
        n = min(chunk, left, st_i.max_len - st_i.t, st_a.max_len - st_a.t)
        if n <= 0:
            raise RuntimeError("weighted_prediction beyond a model's max_seq_len (positional-encoding table exhausted)")
        toks = torch.empty(n, dtype=torch.int64, device=dev)
        lib().call("omr_weighted_decode_steps", ctypes.byref(st_i.desc), ctypes.byref(st_a.desc), float(alpha), ptr(tok), st_i.t, n, ptr(toks), None,
                   ptr(st_i.logits), ptr(st_a.logits), cur_stream())
        st_i.t += n
        st_a.t += n
        for token in toks.cpu().tolist():              # one device sync per chunk
            word = img_model._i2w(token)
            yhat.append(word)
            if word == EOS_TOKEN:
                return yhat
        left -= n
    return yhat
