"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.

A functional fp32 restatement (plain torch CPU ops over a flat state-dict) of the
encoder -> decoder hot path of mariaalfaroc/omr_a2s_multimodal_transformer.  It is the
checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package never does.

Parity status: PINNED.  The reference ships no tests/golden vectors of its own
(SURVEY.md section 4), so this file is pinned by ``tests/golden/*.npz`` -- outputs of the
reference's own modules imported in the build container by ``tests/golden/gen_golden.py``
(see ``tests/test_oracle_golden.py``).

Every function cites the reference lines it restates (paths relative to the reference
root).  Dropout layers are identity by default (eval-mode / p=0 parity); with a ``DropPlan``
the functions reproduce the reference's TRAIN-mode forward -- every nn.Dropout / nn.Dropout2d /
attention-probability dropout site in the reference's call order, with the Python ``random`` draws
that choose the dropout position and kind (encoder.py:102,160,219) -- taking each site's mask from the
plan instead of torch's RNG.  That form is pinned by tests/golden/f12_dropout.npz (the reference's own
modules run with the same injected masks) and is what checks the HIP path's counter-based masks.
"""
from __future__ import annotations

import math
import random

import numpy as np
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

HEIGHT_REDUCTION = 16  # src/transformer/encoder.py:8
WIDTH_REDUCTION = 8  # src/transformer/encoder.py:9


@dataclass
class OracleCfg:
    """Hyper-parameters the reference hard-codes (decoder.py:61-68, model.py:95-96)."""

    d_model: int = 256
    nhead: int = 4
    ff_dim: int = 256
    num_layers: int = 8
    attn_window: int = -1
    pad_idx: int = 0
    dropout: float = 0.1  # decoder.py:65, model.py:29,289 (used only with a DropPlan)
    encoder_dropout: float = 0.5  # encoder.py:251


class DropPlan:
    """Train-mode dropout with injected masks.  ``mask_fn(site, kind, p, shape, channel)`` returns the multiplicative mask
    (keep / (1 - p), broadcastable to ``shape``) of the site-th dropout call of the forward, in the reference's call order.
    kind: "nhwc" = a [B,C,H,W] feature map (MixDropout, PositionalEncoding2D; channel=True is nn.Dropout2d: one decision
    per (b, c)), "rows" = a [B,T,d] token tensor, "attn" = attention probabilities [B,nhead,T,S].  Sites with p = 0 are
    identities and do not count.

    ``relu_fn(site, shape)`` (optional) additionally fixes the ReLU masks: the site-th ReLU of the forward becomes
    ``x * mask`` (0/1, same shape), i.e. the network is evaluated on the piecewise-linear region the mask names.  With the
    masks of another correct fp32 run the forward values are unchanged up to the few pre-activations within rounding noise
    of zero, but the GRADIENT no longer depends on which side of zero those fall -- the only way two correct fp32
    implementations of this network can differ by more than rounding noise (tests/test_dropout_parity_gpu.py)."""

    def __init__(self, mask_fn, relu_fn=None):
        self.mask_fn = mask_fn
        self.relu_fn = relu_fn
        self.sites = 0
        self.relu_sites = 0

    def apply(self, x: Tensor, kind: str, p: float, channel: bool = False) -> Tensor:
        if p <= 0.0:
            return x
        m = self.mask_fn(self.sites, kind, float(p), tuple(x.shape), channel)
        self.sites += 1
        return x * m

    def relu(self, x: Tensor) -> Tensor:
        if self.relu_fn is None:
            return F.relu(x)
        m = self.relu_fn(self.relu_sites, tuple(x.shape))
        self.relu_sites += 1
        return x * m


def _drop(drop: Optional[DropPlan], x: Tensor, kind: str, p: float) -> Tensor:
    return x if drop is None else drop.apply(x, kind, p)


def _relu(drop: Optional[DropPlan], x: Tensor) -> Tensor:
    """nn.ReLU (encoder.py:127,202; TransformerDecoderLayer activation): plain, or with an injected mask (DropPlan.relu)."""
    return F.relu(x) if drop is None else drop.relu(x)


def mix_dropout(drop: Optional[DropPlan], x: Tensor, p: float) -> Tensor:
    """MixDropout.forward, encoder.py:101-104: random.random() < 0.5 -> nn.Dropout(p) else nn.Dropout2d(p / 2)
    (encoder.py:157,216 build it as MixDropout(dropout, dropout / 2))."""
    if drop is None:
        return x
    if random.random() < 0.5:
        return drop.apply(x, "nhwc", p, channel=False)
    return drop.apply(x, "nhwc", p / 2, channel=True)


# --------------------------------------------------------------------------------------
# Encoder  (src/transformer/encoder.py)
# --------------------------------------------------------------------------------------

CONV_STRIDES = ((1, 1), (2, 2), (2, 2), (2, 2), (2, 1))  # encoder.py:255-259


def instance_norm(x: Tensor, eps: float = 1e-3) -> Tensor:
    """nn.InstanceNorm2d(eps=0.001, affine=False, track_running_stats=False), encoder.py:151-156.
    Biased variance over (H, W) per (b, c); padded pixels participate."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def conv_block(sd: SD, p: str, x: Tensor, stride: Tuple[int, int], drop: Optional[DropPlan] = None, dp: float = 0.5) -> Tensor:
    """ConvBlock.forward, encoder.py:159-181.  With a DropPlan: pos = random.randint(1, 3) picks the conv after whose
    ReLU the MixDropout is applied (encoder.py:160-179)."""
    pos = random.randint(1, 3) if drop is not None else 0
    x = _relu(drop, F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1))
    if pos == 1:
        x = mix_dropout(drop, x, dp)
    x = _relu(drop, F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1))
    if pos == 2:
        x = mix_dropout(drop, x, dp)
    x = instance_norm(x)
    x = _relu(drop, F.conv2d(x, sd[p + "conv3.weight"], sd[p + "conv3.bias"], padding=1, stride=stride))
    if pos == 3:
        x = mix_dropout(drop, x, dp)
    return x


def depth_sep_conv(sd: SD, p: str, x: Tensor) -> Tensor:
    """DepthSepConv2D.forward, encoder.py:73-84: depthwise 3x3 (groups=C, pad 1) then 1x1."""
    c = x.shape[1]
    x = F.conv2d(x, sd[p + "depth_conv.weight"], sd[p + "depth_conv.bias"], padding=1, groups=c)
    return F.conv2d(x, sd[p + "point_conv.weight"], sd[p + "point_conv.bias"])


def dsc_block(sd: SD, p: str, x: Tensor, drop: Optional[DropPlan] = None, dp: float = 0.5) -> Tensor:
    """DSCBlock.forward, encoder.py:218-238 (no ReLU after conv3; all strides (1,1) :264-267)."""
    pos = random.randint(1, 3) if drop is not None else 0
    x = _relu(drop, depth_sep_conv(sd, p + "conv1.", x))
    if pos == 1:
        x = mix_dropout(drop, x, dp)
    x = _relu(drop, depth_sep_conv(sd, p + "conv2.", x))
    if pos == 2:
        x = mix_dropout(drop, x, dp)
    x = instance_norm(x)
    x = depth_sep_conv(sd, p + "conv3.", x)
    if pos == 3:
        x = mix_dropout(drop, x, dp)
    return x


def encoder(sd: SD, p: str, x: Tensor, drop: Optional[DropPlan] = None, dp: float = 0.5) -> Tensor:
    """Encoder.forward, encoder.py:271-291. x [B,1,H,W] -> [B,C,ceil(H/16),ceil(W/8)]."""
    for i, s in enumerate(CONV_STRIDES):
        x = conv_block(sd, f"{p}conv_blocks.{i}.", x, s, drop, dp)
    for i in range(4):
        xt = dsc_block(sd, f"{p}dscblocks.{i}.", x, drop, dp)
        x = x + xt if x.shape == xt.shape else xt  # encoder.py:289
    return x


# --------------------------------------------------------------------------------------
# Positional encodings
# --------------------------------------------------------------------------------------


def pe2d_table(num_channels: int, max_h: int, max_w: int) -> Tensor:
    """PositionalEncoding2D buffer, model.py:33-43. Returns [1,C,max_h,max_w]."""
    half = num_channels // 2
    den = torch.pow(10000, torch.arange(0, half, 2) / num_channels)
    pos_h = torch.arange(max_h).unsqueeze(1)
    pos_w = torch.arange(max_w).unsqueeze(1)
    pe = torch.zeros(max_h, max_w, num_channels)
    pe[:, :, 0:half:2] = torch.sin(pos_w / den).unsqueeze(0)
    pe[:, :, 1:half:2] = torch.cos(pos_w / den).unsqueeze(0)
    pe[:, :, half::2] = torch.sin(pos_h / den).unsqueeze(1)
    pe[:, :, half + 1 :: 2] = torch.cos(pos_h / den).unsqueeze(1)
    return pe.permute(2, 0, 1).unsqueeze(0).contiguous()


def pe1d_table(max_len: int, emb_dim: int) -> Tensor:
    """PositionalEncoding1D buffer, decoder.py:21-27. Returns [1,max_len,emb_dim]."""
    pos = torch.arange(max_len).unsqueeze(1)
    den = torch.pow(10000, torch.arange(0, emb_dim, 2) / emb_dim)
    pe = torch.zeros(1, max_len, emb_dim)
    pe[0, :, 0::2] = torch.sin(pos / den)
    pe[0, :, 1::2] = torch.cos(pos / den)
    return pe


def encode_to_memory(sd: SD, enc_prefix: str, pe: Tensor, x: Tensor, drop: Optional[DropPlan] = None,
                     cfg: Optional["OracleCfg"] = None) -> Tensor:
    """encoder -> +PE2D (-> Dropout(0.1), model.py:31,48) -> flatten(2).permute(0,2,1), model.py:143-147. -> [B,S,C]."""
    cfg = cfg if cfg is not None else OracleCfg()
    f = encoder(sd, enc_prefix, x, drop, cfg.encoder_dropout)
    f = f + pe[:, :, : f.shape[2], : f.shape[3]]
    f = _drop(drop, f, "nhwc", cfg.dropout)
    return f.flatten(2).permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------------------
# Multi-head attention, written out (torch nn/functional.py multi_head_attention_forward)
# --------------------------------------------------------------------------------------


def mha(
    q_in: Tensor,
    kv_in: Tensor,
    in_w: Tensor,
    in_b: Tensor,
    out_w: Tensor,
    out_b: Tensor,
    nhead: int,
    score_bias: Optional[Tensor] = None,
    drop: Optional[DropPlan] = None,
    dropout_p: float = 0.0,
) -> Tensor:
    """softmax(Q K^T / sqrt(hd) + bias) V with packed in_proj (rows [0:d]=Wq,[d:2d]=Wk,[2d:3d]=Wv),
    heads = contiguous hd-wide channel slices, then out_proj (SURVEY.md Appendix A "MHA math").
    ``score_bias`` broadcasts to [B, nhead, T, S]; float values are ADDED (so a 0/1 float
    padding mask adds +1.0 -- reference quirk 1), -inf entries mask.  Train mode: dropout on the
    probabilities (nn.MultiheadAttention dropout, decoder.py:91, model.py:292-297)."""
    b, t, d = q_in.shape
    s = kv_in.shape[1]
    hd = d // nhead
    q = F.linear(q_in, in_w[:d], in_b[:d]).view(b, t, nhead, hd).transpose(1, 2)
    k = F.linear(kv_in, in_w[d : 2 * d], in_b[d : 2 * d]).view(b, s, nhead, hd).transpose(1, 2)
    v = F.linear(kv_in, in_w[2 * d :], in_b[2 * d :]).view(b, s, nhead, hd).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    if score_bias is not None:
        scores = scores + score_bias
    p = torch.softmax(scores, dim=-1)
    p = _drop(drop, p, "attn", dropout_p)
    o = torch.matmul(p, v).transpose(1, 2).reshape(b, t, d)
    return F.linear(o, out_w, out_b)


# --------------------------------------------------------------------------------------
# Decoder  (src/transformer/decoder.py)
# --------------------------------------------------------------------------------------


def tgt_attn_mask(t: int, window: int) -> Tensor:
    """get_tgt_masks / create_variable_window_mask, decoder.py:191-254: additive [T,T] mask,
    0 on visible keys, -inf elsewhere.  window>0 and window<T: row i sees [max(0,i-window), i]
    (window+1 keys); otherwise plain causal."""
    i = torch.arange(t).unsqueeze(1)
    j = torch.arange(t).unsqueeze(0)
    visible = j <= i
    if window > 0 and window < t:
        visible = visible & (j >= i - window)
    m = torch.full((t, t), float("-inf"))
    m[visible] = 0.0
    return m


def memory_key_bias(memory: Tensor, memory_len) -> Optional[Tensor]:
    """get_memory_key_padding_mask, decoder.py:150-189, expressed as an additive [B,S] bias:
    None -> None; bool [B,S] -> 0 / -inf (torch _canonical_mask); int lengths -> float 0/1
    that torch ADDS to the scores (+1.0 on padded keys, quirk 1)."""
    if memory_len is None:
        return None
    b, s = memory.shape[:2]
    if memory_len.dtype == torch.bool:
        bias = torch.zeros(b, s)
        bias[memory_len] = float("-inf")
        return bias
    bias = torch.zeros(b, s)
    for i, l in enumerate(memory_len.tolist()):
        bias[i, int(l) :] = 1.0
    return bias


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def decoder_layer(sd: SD, p: str, x: Tensor, memory: Tensor, self_bias: Tensor, mem_bias: Optional[Tensor], nhead: int,
                  drop: Optional[DropPlan] = None, dp: float = 0.0) -> Tensor:
    """Post-norm nn.TransformerDecoderLayer (relu, batch_first) as configured at decoder.py:86-95;
    math per torch nn/modules/transformer.py:1129-1199.  Train-mode dropout sites in call order: self-attention
    probabilities, dropout1, cross-attention probabilities, dropout2, the FFN's inner dropout, dropout3."""
    sa = mha(x, x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
             sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], nhead, self_bias, drop, dp)
    x = layer_norm(x + _drop(drop, sa, "rows", dp), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    ca = mha(x, memory, sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"],
             sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"], nhead, mem_bias, drop, dp)
    x = layer_norm(x + _drop(drop, ca, "rows", dp), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = _drop(drop, _relu(drop, F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), "rows", dp)
    ff = F.linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return layer_norm(x + _drop(drop, ff, "rows", dp), sd[p + "norm3.weight"], sd[p + "norm3.bias"])


def decoder(sd: SD, p: str, tgt: Tensor, memory: Tensor, memory_len, cfg: OracleCfg, drop: Optional[DropPlan] = None) -> Tensor:
    """Decoder.forward, decoder.py:104-148 -> logits [B, V, T]."""
    b, t = tgt.shape
    emb = sd[p + "embedding.weight"][tgt]  # row pad_idx is zero (nn.Embedding padding_idx)
    x = emb + pe1d_table(t, cfg.d_model)  # decoder.py:124 (no sqrt(d) scaling)
    x = _drop(drop, x, "rows", cfg.dropout)  # PositionalEncoding1D dropout, decoder.py:19,32
    mem_bias = memory_key_bias(memory, memory_len)
    self_bias = tgt_attn_mask(t, cfg.attn_window).view(1, 1, t, t)
    if mem_bias is not None:
        # tgt_key_padding_mask = (tgt == 0).float() is ADDED (+1.0); dropped at inference
        # (decoder.py:131-132, 253).
        self_bias = self_bias + (tgt == cfg.pad_idx).to(torch.float32).view(b, 1, 1, t)
        mem_bias = mem_bias.view(b, 1, 1, -1)
    for i in range(cfg.num_layers):
        x = decoder_layer(sd, f"{p}transformer_decoder.layers.{i}.", x, memory, self_bias, mem_bias, cfg.nhead, drop, cfg.dropout)
    w = sd[p + "out_layer.weight"][:, :, 0]  # Conv1d k=1 head, decoder.py:98-102,145-146
    return (F.linear(x, w, sd[p + "out_layer.bias"])).permute(0, 2, 1).contiguous()


def ce_loss(logits_bvt: Tensor, target: Tensor, pad_idx: int = 0) -> Tensor:
    """CrossEntropyLoss(ignore_index=pad) over class dim 1 of [B,V,T], model.py:109,166."""
    logp = torch.log_softmax(logits_bvt, dim=1)
    picked = -logp.gather(1, target.unsqueeze(1)).squeeze(1)
    keep = target != pad_idx
    return (picked * keep).sum() / keep.sum()


# --------------------------------------------------------------------------------------
# Unimodal Transformer  (src/transformer/model.py:54-262)
# --------------------------------------------------------------------------------------


def transformer_forward(sd: SD, x: Tensor, xl, y_in: Tensor, cfg: OracleCfg, max_h: int, max_w: int,
                        drop: Optional[DropPlan] = None) -> Tensor:
    """Transformer.forward, model.py:141-150."""
    pe = pe2d_table(cfg.d_model, math.ceil(max_h / HEIGHT_REDUCTION), math.ceil(max_w / WIDTH_REDUCTION))
    mem = encode_to_memory(sd, "encoder.", pe, x, drop, cfg)
    return decoder(sd, "decoder.", y_in, mem, xl, cfg, drop)


def greedy_decode(sd: SD, dec_prefix: str, memory: Tensor, sos: int, eos: int, max_len: int, cfg: OracleCfg,
                  return_logits: bool = False):
    """validation_step loop, model.py:182-193: bs=1, memory_len=None, full re-run per step,
    argmax of logits[0,:,-1]; stops after emitting eos.  With return_logits the per-step
    (top1, top2) logit values are returned too (get_pred_seq_and_pred_prob_seq, model.py:250-260)."""
    assert memory.shape[0] == 1
    y_in = torch.tensor([[sos]], dtype=torch.long)
    toks: List[int] = []
    tops: List[Tuple[float, float]] = []
    for _ in range(max_len):
        logits = decoder(sd, dec_prefix, y_in, memory, None, cfg)[0, :, -1]
        top2 = logits.topk(2)
        tok = int(logits.argmax())
        toks.append(tok)
        tops.append((float(top2.values[0]), float(top2.values[1])))
        if tok == eos:
            break
        y_in = torch.cat([y_in, torch.tensor([[tok]])], dim=1)
    return (toks, tops) if return_logits else toks


def weighted_decode(sd_img: SD, sd_aud: SD, mem_img: Tensor, mem_aud: Tensor, sos: int, eos: int, max_len: int, cfg: OracleCfg,
                    alpha: float = 0.5) -> List[int]:
    """weighted_prediction, src/multimodal/weighted_multimodal/test.py:21-70: two unimodal models decoded in lock-step from
    the same prefix; per step softmax of each model's last-position logits, alpha * p_img + (1 - alpha) * p_audio, argmax;
    stops after emitting eos.  mem_*: each model's encoder memory [1, S, d] (test.py:28-44)."""
    y_in = torch.tensor([[sos]], dtype=torch.long)
    toks: List[int] = []
    for _ in range(max_len):
        pi = decoder(sd_img, "decoder.", y_in, mem_img, None, cfg)[0, :, -1].softmax(dim=-1)
        pa = decoder(sd_aud, "decoder.", y_in, mem_aud, None, cfg)[0, :, -1].softmax(dim=-1)
        tok = int((alpha * pi + (1 - alpha) * pa).argmax(dim=-1))
        toks.append(tok)
        if tok == eos:
            break
        y_in = torch.cat([y_in, torch.tensor([[tok]])], dim=1)
    return toks


# --------------------------------------------------------------------------------------
# Multimodal  (src/transformer/model.py:268-726)
# --------------------------------------------------------------------------------------


def cross_attention(sd: SD, p: str, query: Tensor, len_query, key_value: Tensor, len_kv, nhead: int = 4,
                    drop: Optional[DropPlan] = None, dp: float = 0.0) -> Tensor:
    """CrossAttention.forward, model.py:299-355.  The bool mask [B,La,Lb] with block
    [lq:, lkv:] = True is tiled with .repeat(nhead,1,1) (model.py:354), so batched head index
    b*nhead+h receives the mask of sample (b*nhead+h) % B  (reference quirk 2)."""
    b, la, _ = query.shape
    lb = key_value.shape[1]
    bias = None
    if len_query is not None and len_kv is not None:
        m = torch.zeros(b, la, lb)
        for i, (lq, lkv) in enumerate(zip(len_query.tolist(), len_kv.tolist())):
            m[i, int(lq) :, int(lkv) :] = float("-inf")
        idx = (torch.arange(b * nhead)) % b
        bias = m[idx].view(b, nhead, la, lb)
    a = p + "attention."
    return mha(query, key_value, sd[a + "in_proj_weight"], sd[a + "in_proj_bias"],
               sd[a + "out_proj.weight"], sd[a + "out_proj.bias"], nhead, bias, drop, dp)


def _len_mask(n: int, lens) -> Tensor:
    m = torch.zeros(len(lens), n, dtype=torch.bool)
    for i, l in enumerate(lens.tolist()):
        m[i, int(l) :] = True
    return m


def mixer(sd: SD, kind: str, xi: Tensor, xa: Tensor, xli, xla, drop: Optional[DropPlan] = None, dp: float = 0.0):
    """mixer_concat / mixer_attn_img / mixer_attn_audio / mixer_attn_both, model.py:644-726."""
    have = xli is not None and xla is not None
    if kind == "concat":
        x = torch.cat([xi, xa], dim=1)
        xl = torch.cat([_len_mask(xi.shape[1], xli), _len_mask(xa.shape[1], xla)], dim=1) if have else None
        return x, xl  # bool mask -> true -inf masking (model.py:663-672)
    if kind == "attn_img":  # q = audio, kv = image
        return cross_attention(sd, "cross_attn.", xa, xla, xi, xli, drop=drop, dp=dp), (xla if have else None)
    if kind == "attn_audio":  # q = image, kv = audio
        return cross_attention(sd, "cross_attn.", xi, xli, xa, xla, drop=drop, dp=dp), (xli if have else None)
    if kind == "attn_both":
        # model.py:723-725: variable shadowing -- the second attention uses the ALREADY
        # ATTENDED audio as key/value (quirk 3); one shared cross_attn module.
        xa2, xla2 = mixer(sd, "attn_img", xi, xa, xli, xla, drop, dp)
        xi2, xli2 = mixer(sd, "attn_audio", xi, xa2, xli, xla2, drop, dp)
        return mixer(sd, "concat", xi2, xa2, xli2, xla2)
    raise ValueError(f"Invalid mixer type: {kind}")


def multimodal_forward(sd: SD, xi: Tensor, xli, xa: Tensor, xla, y_in: Tensor, cfg: OracleCfg, mixer_type: str,
                       max_img_hw: Tuple[int, int], max_audio_hw: Tuple[int, int], modality: str = "both",
                       drop: Optional[DropPlan] = None) -> Tensor:
    """MultimodalTransformer.forward / encoder_forward, model.py:485-543 with the modality
    choice of apply_teacher_forcing_modality (model.py:561-575) passed in explicitly."""
    pe_i = pe2d_table(cfg.d_model, math.ceil(max_img_hw[0] / 16), math.ceil(max_img_hw[1] / 8))
    pe_a = pe2d_table(cfg.d_model, math.ceil(max_audio_hw[0] / 16), math.ceil(max_audio_hw[1] / 8))
    mi = encode_to_memory(sd, "image_encoder.", pe_i, xi, drop, cfg)
    ma = encode_to_memory(sd, "audio_encoder.", pe_a, xa, drop, cfg)
    if modality == "image":
        mem, ml = mi, xli
    elif modality == "audio":
        mem, ml = ma, xla
    else:
        mem, ml = mixer(sd, mixer_type, mi, ma, xli, xla, drop, cfg.dropout)
    return decoder(sd, "decoder.", y_in, mem, ml, cfg, drop)


# --------------------------------------------------------------------------------------
# Optimiser and metrics
# --------------------------------------------------------------------------------------


def adam_step(params: Sequence[Tensor], grads: Sequence[Tensor], exp_avg: Sequence[Tensor], exp_avg_sq: Sequence[Tensor],
              step, lr: float = 1e-4, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor math (torch optim/adam.py:347), as configured at
    model.py:134-139 (no weight decay, no amsgrad).  ``step`` is 1-based.  In place.
    A parameter whose gradient is None is SKIPPED -- no moment decay, no update, no step count (torch optim/adam.py
    _init_group collects only p.grad is not None; Lightning zeroes with set_to_none): the modality-drop steps of
    MultimodalTransformer leave one encoder and cross_attn without gradients (model.py:510-519).  ``step`` may therefore be a
    list with one 1-based count per parameter (the caller increments the counts of the parameters that had a gradient)."""
    steps = step if isinstance(step, (list, tuple)) else [step] * len(params)
    for p, g, m, v, st in zip(params, grads, exp_avg, exp_avg_sq, steps):
        if g is None:
            continue
        bc1 = 1.0 - b1**st
        bc2 = 1.0 - b2**st
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def levenshtein(a: Sequence, b: Sequence) -> int:
    """metrics.py:56-74."""
    if len(a) > len(b):
        a, b = b, a
    prev = list(range(len(a) + 1))
    for i in range(1, len(b) + 1):
        cur = [i] + [0] * len(a)
        for j in range(1, len(a) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[j - 1] != b[i - 1]))
        prev = cur
    return prev[len(a)]


def compute_ed_metrics(y_true: List[List[str]], y_pred: List[List[str]]) -> Dict[str, float]:
    """compute_ed_metrics, metrics.py:52-88."""
    ed_acc = length_acc = label_acc = 0
    for t, h in zip(y_true, y_pred):
        ed = levenshtein(t, h)
        ed_acc += ed
        length_acc += len(t)
        label_acc += ed > 0
    return {"sym-er": 100.0 * ed_acc / length_acc, "seq-er": 100.0 * label_acc / len(y_pred)}


# --------------------------------------------------------------------------------------
# Collate contract (src/data/preprocessing.py:55-144)
# --------------------------------------------------------------------------------------


def collate_unimodal(batch, pad_value: float):
    """ar_batch_preparation_unimodal, preprocessing.py:85-103: right/bottom pad x with
    pad_value (image 1.0 :110, audio 0.0 :117), xl int32, y_in=y[:-1], y_out=y[1:] padded with 0 (int64)."""
    xs, xls, ys = zip(*batch)
    mh = max(t.shape[1] for t in xs)
    mw = max(t.shape[2] for t in xs)
    x = torch.stack([F.pad(t, (0, mw - t.shape[2], 0, mh - t.shape[1]), value=pad_value) for t in xs]).float()
    xl = torch.tensor(xls, dtype=torch.int32)
    ml = max(len(t) for t in ys) - 1
    y_in = torch.stack([F.pad(t[:-1], (0, ml - (len(t) - 1))) for t in ys]).long()
    y_out = torch.stack([F.pad(t[1:], (0, ml - (len(t) - 1))) for t in ys]).long()
    return x, xl, y_in, y_out


def log_stft(y, n_fft: int = 2048, hop: int = 512, sr: int = 22050, fmax: float = 2093.0):
    """Spectrogram front end of the audio branch (reference src/data/preprocessing.py:17-30) for a signal that is already at
    22 050 Hz: librosa.stft(hop_length=512, win_length=2048, window="hann") -- centred frames, zero ("constant") padding of
    n_fft//2 samples at both ends, periodic Hann window, 1 + len(y)//hop frames -- keep the bins with frequency
    k*sr/n_fft <= 2093 Hz (195 of them), amplitude_to_db(|S|, ref=max) = 20 log10(max(1e-5, |S|)) - 20 log10(max(1e-5, max|S|))
    clipped at 80 dB below its own maximum, then /80 + 1.  Returns float32 [195, frames].
    PARITY UNPINNED: librosa is not installed in the build container, so this restates librosa's documented algorithm
    (librosa 0.10 core/spectrum.py: stft, amplitude_to_db, power_to_db) without a fixture from the reference itself; the
    resampling step (librosa.resample) is not restated."""
    import numpy as np
    y = np.asarray(y, dtype=np.float64)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n_frames = 1 + len(y) // hop
    n = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                 # scipy.signal.get_window("hann", n_fft, fftbins=True)
    frames = np.stack([yp[i * hop:i * hop + n_fft] * win for i in range(n_frames)], axis=1)     # [n_fft, frames]
    spec = np.fft.rfft(frames, axis=0)
    keep = (np.arange(n_fft // 2 + 1) * sr / n_fft) <= fmax
    mag = np.abs(spec[keep])
    amin = 1e-5
    db = 20.0 * np.log10(np.maximum(amin, mag)) - 20.0 * np.log10(max(amin, mag.max()))
    db = np.maximum(db, db.max() - 80.0)
    return (db / 80.0 + 1.0).astype(np.float32)


def resample_poly(y, orig_sr: int, target_sr: int):
    """Polyphase sample-rate conversion, the first line of get_spectrogram_from_raw_audio (src/data/preprocessing.py:19:
    librosa.resample(raw_audio, orig_sr=sr, target_sr=22050)).  This restates librosa's res_type="polyphase" =
    scipy.signal.resample_poly(y, up, down) (scipy/signal/_signaltools.py): a Kaiser(beta = 5) windowed-sinc low-pass of
    2 * 10 * max(up, down) + 1 taps at cutoff 1 / max(up, down), scaled by `up`, applied to the zero-stuffed signal with zero
    ("constant") edges, output length ceil(n * up / down).  Pinned against scipy itself (tests/golden/f18_resample.npz).
    PARITY UNPINNED with respect to the reference's call: librosa's DEFAULT res_type is "soxr_hq" (the soxr library, absent
    here, algorithm not restated), so the reference's own samples differ from these at the filter-design level."""
    import math
    import numpy as np
    y = np.asarray(y, dtype=np.float64)
    g = math.gcd(int(orig_sr), int(target_sr))
    up, down = int(target_sr) // g, int(orig_sr) // g
    if up == down:
        return y.astype(np.float32)
    h, r0 = resample_filter(up, down)
    n_out = -(-len(y) * up // down)
    out = np.zeros(n_out)
    for n in range(n_out):                       # y[n] = sum_i x[i] h[(n + r0) * down - i * up]
        t = (n + r0) * down
        i_lo = max(0, -(-(t - len(h) + 1) // up))
        i_hi = min(len(y) - 1, t // up)
        if i_hi >= i_lo:
            i = np.arange(i_lo, i_hi + 1)
            out[n] = np.dot(y[i], h[t - i * up])
    return out.astype(np.float32)


def resample_filter(up: int, down: int):
    """(taps h with scipy's leading zero padding, samples n_pre_remove dropped from the filtered signal) of
    scipy.signal.resample_poly(window=("kaiser", 5.0)); firwin restated: h = c sinc(c m) kaiser(N, 5), unit DC gain, times up."""
    import numpy as np
    max_rate = max(up, down)
    half_len = 10 * max_rate
    ntaps = 2 * half_len + 1
    c = 1.0 / max_rate
    m = np.arange(ntaps) - half_len
    h = c * np.sinc(c * m) * np.kaiser(ntaps, 5.0)
    h = h / h.sum() * up
    n_pre_pad = down - half_len % down
    h = np.concatenate([np.zeros(n_pre_pad), h])
    return h, (half_len + n_pre_pad) // down


# --------------------------------------------------------------------------------------
# score-image front end (src/data/preprocessing.py:44-52)
# --------------------------------------------------------------------------------------
# PIL convert("L") -> Image.resize((int(H * w / h), H)) -> ToTensor.  The arithmetic lives in Pillow (third-party; the
# reference pins no version, the build container has 12.2.0): restated here from Pillow's documented 8-bit resampler and
# PINNED against Pillow itself -- tests/golden/f11_image.npz holds Pillow's outputs (tests/golden/gen_image_golden.py),
# and tests/test_oracle_golden.py also compares with the importable Pillow directly.  Integer work: bit-exact.

def pil_gray(rgb: np.ndarray) -> np.ndarray:
    """uint8 [h, w] / [h, w, 3|4] -> uint8 [h, w]: L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (alpha ignored)."""
    a = np.asarray(rgb)
    if a.ndim == 2 or a.shape[2] == 1:
        return a.reshape(a.shape[0], a.shape[1]).copy()
    r, g, b = (a[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def pil_bicubic_tables(in_size: int, out_size: int):
    """Per output pixel: first contributing input pixel, tap count, and the taps in 22-bit fixed point (Keys cubic a = -0.5,
    support 2 stretched by max(1, in/out), normalised in double precision)."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 2.0 * fs
    ksize = int(np.ceil(support)) * 2 + 1

    def cubic(x: float) -> float:
        a = -0.5
        x = abs(x)
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0

    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [cubic((x + xmin - center + 0.5) * (1.0 / fs)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, xmax)
        for x, v in enumerate(w):
            coefs[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
    return bounds, coefs


def _pil_pass(img: np.ndarray, bounds: np.ndarray, coefs: np.ndarray) -> np.ndarray:
    """One horizontal pass: uint8 [h, w] -> uint8 [h, out_w]."""
    out = np.empty((img.shape[0], bounds.shape[0]), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx, (xmin, n) in enumerate(bounds):
        ss = (src[:, xmin:xmin + n] * coefs[xx, :n].astype(np.int64)).sum(axis=1) + (1 << 21)
        out[:, xx] = np.clip(ss >> 22, 0, 255)
    return out


def pil_resize_L(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.resize((out_w, out_h)) of a mode-"L" image: horizontal pass, then vertical pass, each only if that size changes."""
    h, w = img.shape
    if out_w != w:
        img = _pil_pass(img, *pil_bicubic_tables(w, out_w))
    if out_h != h:
        img = _pil_pass(np.ascontiguousarray(img.T), *pil_bicubic_tables(h, out_h)).T
    return np.ascontiguousarray(img)


def preprocess_image(pixels: np.ndarray, img_height=None) -> np.ndarray:
    """preprocessing.py:44-52 on uint8 pixels -> float32 [1, H, W] in [0, 1]."""
    g = pil_gray(pixels)
    if img_height is not None:
        g = pil_resize_L(g, img_height, int(img_height * g.shape[1] / g.shape[0]))
    return (g.astype(np.float32) / np.float32(255.0))[None]


# --------------------------------------------------------------------------------------
# late fusion (src/multimodal/smith_waterman/test.py:136-150)
# --------------------------------------------------------------------------------------

def sw_align(ref, query, match: int = 2, mismatch: int = -1, gap_penalty: int = -1, gap_extension_penalty: int = -1):
    """Smith-Waterman local alignment as swalign 0.3.x's LocalAlignment.align computes it (restated from the package's
    published algorithm; the package is absent here: PARITY UNPINNED).  Plain Python, small inputs only.
    -> (ops, r_pos, q_pos, score); ops over {'m', 'i', 'd'}."""
    nq, nr = len(query), len(ref)
    val = [[0] * (nr + 1) for _ in range(nq + 1)]
    op = [[" "] * (nr + 1) for _ in range(nq + 1)]
    run = [[0] * (nr + 1) for _ in range(nq + 1)]
    for row in range(1, nq + 1):
        op[row][0] = "i"
    for col in range(1, nr + 1):
        op[0][col] = "d"
    best = (0, 0, 0)
    for row in range(1, nq + 1):
        for col in range(1, nr + 1):
            mm = val[row - 1][col - 1] + (match if query[row - 1] == ref[col - 1] else mismatch)
            ins_run = del_run = 0
            if op[row - 1][col] == "i":
                ins_run = run[row - 1][col]
                ins = 0 if val[row - 1][col] == 0 else val[row - 1][col] + gap_extension_penalty
            else:
                ins = val[row - 1][col] + gap_penalty
            if op[row][col - 1] == "d":
                del_run = run[row][col - 1]
                dele = 0 if val[row][col - 1] == 0 else val[row][col - 1] + gap_extension_penalty
            else:
                dele = val[row][col - 1] + gap_penalty
            cell = max(mm, dele, ins, 0)
            if del_run and cell == dele:
                o, rl = "d", del_run + 1
            elif ins_run and cell == ins:
                o, rl = "i", ins_run + 1
            elif cell == mm:
                o, rl = "m", 0
            elif cell == dele:
                o, rl = "d", 1
            elif cell == ins:
                o, rl = "i", 1
            else:
                o, rl, cell = "x", 0, 0
            val[row][col], op[row][col], run[row][col] = cell, o, rl
            if cell >= best[0]:
                best = (cell, row, col)
    score, row, col = best
    ops = []
    while val[row][col] > 0:
        o = op[row][col]
        ops.append(o)
        if o == "m":
            row, col = row - 1, col - 1
        elif o == "i":
            row -= 1
        elif o == "d":
            col -= 1
        else:
            break
    return "".join(reversed(ops)), col, row, score
