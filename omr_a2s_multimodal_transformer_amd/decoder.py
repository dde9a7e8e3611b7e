"""Transformer decoder on HIP kernels -- same class names, constructor signature, attribute and state-dict
names as the reference's src/transformer/decoder.py (which wraps nn.TransformerDecoder, post-norm, ReLU).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .runtime import next_seed


def sinusoid_1d(max_len: int, emb_dim: int) -> torch.Tensor:
    """decoder.py:21-27 -> [1, max_len, emb_dim]."""
    pos = torch.arange(max_len).unsqueeze(1)
    den = torch.pow(10000, torch.arange(0, emb_dim, 2) / emb_dim)
    pe = torch.zeros(1, max_len, emb_dim)
    pe[0, :, 0::2] = torch.sin(pos / den)
    pe[0, :, 1::2] = torch.cos(pos / den)
    return pe


def _dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    if not training or p <= 0.0:
        return x
    return Fn.DropoutFn.apply(x, p, next_seed("rows", p), False, False)


class PositionalEncoding1D(nn.Module):
    """decoder.py:7-32."""

    def __init__(self, max_len: int, emb_dim: int, dropout_p: float = 0.1):
        super().__init__()
        self.dropout_p = dropout_p
        self.register_buffer("pe", sinusoid_1d(max_len, emb_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, d = x.shape
        y = K.add_pe2d(x.contiguous().view(B, 1, T, d), self.pe[0].view(1, -1, d)).view(B, T, d)
        return _dropout(y, self.dropout_p, self.training)


class Embedding(nn.Module):
    """nn.Embedding parameter holder: N(0,1) init with the padding row zeroed (decoder.py:73-77)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx
        w = torch.randn(num_embeddings, embedding_dim)
        if padding_idx is not None:
            w[padding_idx].zero_()
        self.weight = nn.Parameter(w)


class Linear(nn.Module):
    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.linear(x, self.weight, self.bias)


class LayerNorm(nn.Module):
    def __init__(self, d: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class MultiheadAttention(nn.Module):
    """nn.MultiheadAttention parameter layout (packed in_proj rows [Wq; Wk; Wv], out_proj) and default init
    (Xavier-uniform in_proj_weight, zero in_proj_bias / out_proj.bias)."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        bound = math.sqrt(6.0 / (4 * embed_dim))
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim).uniform_(-bound, bound))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        with torch.no_grad():
            self.out_proj.bias.zero_()

    def self_attention(self, x, causal: bool, window: int, key_bias, training: bool):
        qkv = Fn.linear(x, self.in_proj_weight, self.in_proj_bias)
        p = self.dropout if training else 0.0
        o = Fn.AttentionFn.apply(qkv, None, self.num_heads, causal, window, key_bias, None, None, p, next_seed("attn", p) if p > 0 else 0)
        return Fn.linear(o, self.out_proj.weight, self.out_proj.bias)

    def project_kv(self, memory):
        d = self.embed_dim
        return Fn.linear(memory, self.in_proj_weight, self.in_proj_bias, rows=(d, 3 * d))

    def cross_attention(self, x, kv, key_bias, training: bool, blk_lq=None, blk_lkv=None):
        d = self.embed_dim
        q = Fn.linear(x, self.in_proj_weight, self.in_proj_bias, rows=(0, d))
        p = self.dropout if training else 0.0
        o = Fn.AttentionFn.apply(q, kv, self.num_heads, False, -1, key_bias, blk_lq, blk_lkv, p, next_seed("attn", p) if p > 0 else 0)
        return Fn.linear(o, self.out_proj.weight, self.out_proj.bias)


class TransformerDecoderLayer(nn.Module):
    """Post-norm nn.TransformerDecoderLayer(relu, batch_first) math, torch nn/modules/transformer.py:1129-1199."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, dropout: float):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = LayerNorm(d_model), LayerNorm(d_model), LayerNorm(d_model)
        self.dropout_p = dropout

    def forward(self, x, memory, window: int, self_key_bias, mem_key_bias, kv=None):
        """kv: this layer's cross-attention K|V of `memory` when the decoder projected all layers at once."""
        tr, p = self.training, self.dropout_p
        drop = (lambda: (p, next_seed("rows", p))) if (tr and p > 0.0) else (lambda: None)      # dropout1/2/3 ride inside the add+LayerNorm kernels
        sa = self.self_attn.self_attention(x, True, window, self_key_bias, tr)
        x = Fn.AddLayerNormFn.apply(sa, x, self.norm1.weight, self.norm1.bias, drop())
        if kv is None:
            kv = self.multihead_attn.project_kv(memory)
        ca = self.multihead_attn.cross_attention(x, kv, mem_key_bias, tr)
        x = Fn.AddLayerNormFn.apply(ca, x, self.norm2.weight, self.norm2.bias, drop())
        h = Fn.linear(x, self.linear1.weight, self.linear1.bias, relu=True, mask_own=True, drop=drop())     # FFN dropout in the GEMM epilogue
        ff = Fn.linear(h, self.linear2.weight, self.linear2.bias)
        return Fn.AddLayerNormFn.apply(ff, x, self.norm3.weight, self.norm3.bias, drop())


class TransformerDecoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        # nn.TransformerDecoder deep-copies ONE layer, so all layers start identical
        first = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout)
        layers = [first]
        for _ in range(num_layers - 1):
            l = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout)
            l.load_state_dict(first.state_dict())
            layers.append(l)
        self.layers = nn.ModuleList(layers)


class Conv1d(nn.Module):
    """nn.Conv1d(k=1) parameter holder: weight [out, in, 1] (decoder.py:98-102)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        bound = 1.0 / math.sqrt(in_channels)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))


class Decoder(nn.Module):
    """decoder.py:35-148.  forward(tgt [B,T] int64, memory [B,S,d], memory_len) -> logits [B, V, T]
    (a stride permutation of the row-major [B*T, V] buffer the head GEMM writes)."""

    def __init__(self, output_size: int, max_seq_len: int, num_embeddings: int, embedding_dim: int = 256, padding_idx: int = 0,
                 ff_dim: int = 256, dropout_p: float = 0.1, nhead: int = 4, num_transformer_layers: int = 8, attn_window: int = -1):
        super().__init__()
        if (embedding_dim // nhead) not in (32, 64) or embedding_dim % nhead:
            raise NotImplementedError("HIP attention kernels cover head_dim 32 and 64 (reference: 256/4 = 64)")
        self.embedding = Embedding(num_embeddings, embedding_dim, padding_idx)
        self.pos_1d = PositionalEncoding1D(max_seq_len, embedding_dim, dropout_p)
        self.attn_window = attn_window
        self.transformer_decoder = TransformerDecoder(embedding_dim, nhead, ff_dim, dropout_p, num_transformer_layers)
        self.out_layer = Conv1d(embedding_dim, output_size)
        self.padding_idx = padding_idx
        self.output_size = output_size

    # ---- mask builders (host logic; vectorised restatement of decoder.py:150-254) ---------------------------
    def get_memory_key_padding_mask(self, memory: torch.Tensor, memory_len: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """None -> None; bool [B,S] -> cloned bool mask (-inf semantics); integer lengths -> FLOAT32 0/1 mask
        that is ADDED to the scores (+1.0 on padded keys: reference quirk 1, decoder.py:186-188)."""
        if memory_len is None:
            return None
        if memory_len.dtype == torch.bool:
            assert memory_len.shape[0] == memory.shape[0], f"Different batch sizes for memory and memory_len: {memory.shape[0]} != {memory_len.shape[0]}"
            assert memory_len.shape[1] == memory.shape[1], f"Different sequence lengths for memory and memory_len: {memory.shape[1]} != {memory_len.shape[1]}"
            return memory_len.clone()
        pos = torch.arange(memory.shape[1], device=memory.device).unsqueeze(0)
        return (pos >= memory_len.to(memory.device).long().unsqueeze(1)).to(torch.float32)

    @staticmethod
    def create_variable_window_mask(size: int, window_size: int, dtype=torch.float32, device=torch.device("cpu")) -> torch.Tensor:
        """decoder.py:191-217 (materialised form, for API parity; the kernels take `window` as a parameter)."""
        i = torch.arange(size, device=device).unsqueeze(1)
        j = torch.arange(size, device=device).unsqueeze(0)
        vis = j <= i
        if window_size < size:
            vis = vis & (j >= i - window_size)
        return torch.full((size, size), float("-inf"), dtype=dtype, device=device).masked_fill(vis, 0.0)

    def get_tgt_masks(self, tgt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        T = tgt.shape[1]
        if self.attn_window > 0:
            tgt_mask = self.create_variable_window_mask(T, self.attn_window, device=tgt.device)
        else:
            tgt_mask = self.create_variable_window_mask(T, T, device=tgt.device)
        return tgt_mask, (tgt == 0).to(torch.float32)

    @staticmethod
    def _as_key_bias(mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if mask is None:
            return None
        if mask.dtype == torch.bool:
            return torch.zeros(mask.shape, dtype=torch.float32, device=mask.device).masked_fill(mask, float("-inf"))
        return mask.contiguous()

    def _cross_kv_pack(self, dt):
        """Row-group views over the packed cross-attention in_proj parameters of all layers, valid when the flat buffer holds
        them back to back (params._placement_order; checked here on every call, so any other layout just takes the per-layer path)."""
        layers = self.transformer_decoder.layers
        ws = [l.multihead_attn.in_proj_weight for l in layers]
        bs = [l.multihead_attn.in_proj_bias for l in layers]
        if len(layers) < 2 or getattr(ws[0], "omr_phys", None) is None:
            return None
        d = ws[0].shape[1]
        if (2 * d) % 128:
            return None
        L = len(layers)

        def block(tensors, rows):
            t0 = tensors[0]
            step = t0.numel() * t0.element_size()
            if any(t.data_ptr() != t0.data_ptr() + i * step or not t.is_contiguous() for i, t in enumerate(tensors)):
                return None
            return torch.as_strided(t0, (L * rows,) + tuple(t0.shape[1:]), t0.stride())

        w = block([Fn.wt(p, dt) for p in ws], 3 * d)
        gw = block([p.omr_grad for p in ws], 3 * d)
        b = block([p.omr_phys for p in bs], 3 * d)
        gb = block([p.omr_grad for p in bs], 3 * d)
        if w is None or gw is None or b is None or gb is None:
            return None
        return dict(L=L, d=d, w=w, gw=gw, b=b, gb=gb)

    # ---- KV-cached greedy decoding.  The reference re-runs the whole prefix every step
    #      (model.py:184-193, O(T^3)); here each step projects ONE token, appends its self-attention K|V to a cache and
    #      reads the cross-attention K|V that were projected once.  Same kernels, same per-row arithmetic order.
    @torch.no_grad()
    def init_decode(self, memory: torch.Tensor) -> dict:
        emb_w = self.embedding.weight
        dt = torch.bfloat16 if getattr(emb_w, "omr_lowp", None) is not None else torch.float32
        if memory.dtype != dt:
            memory = K.cast(memory.contiguous(), dt)
        memory = memory.contiguous()
        layers = self.transformer_decoder.layers
        d = emb_w.shape[1]
        B = memory.shape[0]                   # the reference decodes bs = 1; the cache is batched
        max_len = self.pos_1d.pe.shape[1]
        return dict(t=0, dtype=dt, d=d,
                    cross_kv=[l.multihead_attn.project_kv(memory) for l in layers],             # [B,S,2d] per layer, once
                    self_kv=torch.empty((len(layers), B, max_len, 2 * d), dtype=dt, device=memory.device))

    @torch.no_grad()
    def decode_step(self, token: torch.Tensor, st: dict) -> torch.Tensor:
        """token int64 [B,1] -> fp32 logits of the next position ([V] for B = 1, else [B,V]); advances the cache.  Every
        sample of the batch is at the same position t; rows are computed independently (same per-row arithmetic as bs = 1)."""
        t, d, dt = st["t"], st["d"], st["dtype"]
        if t >= self.pos_1d.pe.shape[1]:
            raise RuntimeError("decode_step beyond max_seq_len (positional-encoding table exhausted)")
        x = K.embed_pe(token.contiguous(), Fn.wt(self.embedding.weight, dt).view(-1, d), self.pos_1d.pe[0, t:t + 1])   # [1,1,d]
        lo = max(0, t - self.attn_window) if self.attn_window > 0 else 0     # banded causal mask = a key range (decoder.py:213-214)
        for li, layer in enumerate(self.transformer_decoder.layers):
            sa = layer.self_attn
            qkv = Fn.linear(x, sa.in_proj_weight, sa.in_proj_bias)                                # [B,1,3d]
            st["self_kv"][li, :, t].copy_(qkv[:, 0, d:])
            o = Fn.AttentionFn.apply(qkv[..., :d], st["self_kv"][li, :, lo:t + 1], sa.num_heads, False, -1, None, None, None, 0.0, 0)
            x = Fn.AddLayerNormFn.apply(Fn.linear(o, sa.out_proj.weight, sa.out_proj.bias), x, layer.norm1.weight, layer.norm1.bias)
            ca = layer.multihead_attn.cross_attention(x, st["cross_kv"][li], None, False)
            x = Fn.AddLayerNormFn.apply(ca, x, layer.norm2.weight, layer.norm2.bias)
            h = Fn.linear(x, layer.linear1.weight, layer.linear1.bias, relu=True)
            x = Fn.AddLayerNormFn.apply(Fn.linear(h, layer.linear2.weight, layer.linear2.bias), x, layer.norm3.weight, layer.norm3.bias)
        V = self.output_size
        logits = Fn.linear(x, self.out_layer.weight, self.out_layer.bias, out_ld=K.round_up(V, 8))[:, 0]         # [B,V]
        st["t"] = t + 1
        logits = logits if logits.dtype == torch.float32 else K.cast(logits.contiguous(), torch.float32)
        return logits[0] if logits.shape[0] == 1 else logits

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, memory_len: Optional[torch.Tensor]) -> torch.Tensor:
        emb_w = self.embedding.weight
        dt = torch.bfloat16 if getattr(emb_w, "omr_lowp", None) is not None else torch.float32
        if memory.dtype != dt:
            memory = K.cast(memory.contiguous(), dt)
        memory = memory.contiguous()
        B, T = tgt.shape
        x = Fn.EmbedPEFn.apply(tgt.contiguous(), emb_w, self.pos_1d.pe[0], self.padding_idx, dt)
        x = _dropout(x, self.pos_1d.dropout_p, self.training)
        mem_mask = self.get_memory_key_padding_mask(memory, memory_len)
        mem_bias = self._as_key_bias(mem_mask)
        # tgt_key_padding_mask = (tgt == 0).float() is ADDED (+1.0); dropped when there is no memory mask (decoder.py:131-132)
        self_bias = None if mem_mask is None else (tgt == 0).to(torch.float32).contiguous()
        window = self.attn_window if self.attn_window > 0 else -1
        layers = self.transformer_decoder.layers
        pack = self._cross_kv_pack(dt)
        kvs = [None] * len(layers)
        if pack is not None:
            sink = Fn.KVGradSink()
            kvs = Fn.FusedCrossKVFn.apply(memory, pack, sink)
            for li, kv in enumerate(kvs):
                kv.omr_grad_sink = (sink, li)
        for layer, kv in zip(layers, kvs):
            x = layer(x, memory, window, self_bias, mem_bias, kv)
        V = self.output_size
        logits = Fn.linear(x, self.out_layer.weight, self.out_layer.bias, out_ld=K.round_up(V, 8))  # [B,T,V], row pitch round_up(V,8)
        return logits.permute(0, 2, 1)  # [B, V, T] (decoder.py:145-146)
