"""torch.autograd plumbing over the HIP kernels (kernels.py).

Every Function's forward AND backward run hand-written gfx950 kernels through the C ABI; torch only
provides tensors, views and the autograd graph.  Parameter gradients are accumulated by the kernels
straight into the model's flat fp32 gradient buffer (``param.omr_grad`` views, see params.py), so the
Functions return ``None`` for parameters: there is no per-parameter gradient copy, the optimizer and
the gradient all-reduce work on one flat buffer.

Gradient protocol inside the encoder blocks: a layer whose output goes through ReLU (+dropout) can
leave the activation-function backward to its consumer (``mask_input=True`` there), which applies
``(x > 0) * scale`` in the epilogue of its data-gradient kernel -- x being exactly the saved
post-activation tensor, no mask is ever stored or regenerated.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import kernels as K
from .runtime import WgradStream, note_relu

Tensor = torch.Tensor

# Conv3x3Fn.backward through a normalise-on-load conv: True = take the data gradient twice (sums only, then again with the
# InstanceNorm backward applied in its epilogue: omr_conv3x3_fwd stat_mode 4 / 5) instead of storing it for the stand-alone apply
# pass.  Measured on C2 (round 3): the recomputation costs what the apply pass saves (conv data gradients 9.24 -> 10.83 ms,
# InstanceNorm 1.99 -> 0.33 ms per step; step 26.7 -> 27.3 ms without the side stream) -- the data-gradient kernels are not
# purely HBM-bound, a second pass costs almost a full one.  Kept as an option (same gradients: tests/test_round3_gpu.py).
TWO_PASS_NORM_BWD = False

# One-pass backward of the <= 32-channel stride-1 convs (csrc/conv_bwd_fused.hip; bf16): bit 1 = data + weight + bias gradient in one
# kernel (the operands of the weight gradient are the tiles the data gradient already holds), bit 2 = the InstanceNorm backward of the
# layer above applied while that kernel loads its tile (ConvBlock conv2 <- norm <- conv3: conv3's backward hands its un-applied
# gradient on, conv2's backward consumes it; no omr_instnorm_bwd_apply launch, its output is never written), bit 4 = the one-pass form
# of the 16-channel normalise-on-load conv (ConvBlock 0's conv3), bit 8 = of the 32-channel stride-(2,2) one (ConvBlock 1's conv3).
FUSED_BWD = int(os.environ.get("OMR_FUSED_BWD", "15"))
FUSED_NORM_CHANNELS = tuple(int(c) for c in os.environ.get("OMR_FUSED_NORM_CHANNELS", "16,32").split(",") if c)   # conv2 widths that take the hand-on
FUSED_MIN_COUT = int(os.environ.get("OMR_FUSED_MIN_COUT", "16"))
_PENDING_NORM = {}        # data_ptr of a handed-on gradient -> (y, mean, rstd, ws, slots, relu_mask, relu_scale); consumed by the producer conv's backward


def clear_pending_norm() -> None:
    """Drop hand-ons a failed backward left behind (FlatParams.zero_grad)."""
    _PENDING_NORM.clear()


def wt(p: Tensor, dtype: torch.dtype) -> Tensor:
    """Physical compute view of a parameter in the compute dtype (params.py attaches them)."""
    if dtype == torch.float32:
        return p.omr_phys
    lp = p.omr_lowp
    if lp is None:
        raise RuntimeError("bf16 compute copy missing: call model.flatten_parameters(compute_dtype=torch.bfloat16)")
    return lp


def split_k_for(m_red: int, out_rows: int = 128, out_cols: int = 128) -> int:
    """Split-K factor of a weight-gradient GEMM reducing over m_red rows into an [out_rows, out_cols] matrix: about
    384 blocks in flight (128x128 tiles x splits), at least 256 reduction rows per split, power of two.  Fewer splits
    starve the CUs (the main loop is latency-bound), more splits drown in fp32 atomics (measured: tools/gemm_splitk_sweep.py)."""
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    want = max(8, 384 // tiles)
    sk = 1 << (want.bit_length() - 1)
    return max(1, min(sk, (m_red + 255) // 256))


# ------------------------------------------------------------------------------------------------ encoder

class Conv3x3Fn(Function):
    """[InstanceNorm-apply ->] Conv2d(3x3, pad 1, stride) -> +bias [-> ReLU] [-> MixDropout] on NHWC, one kernel
    (ConvBlock, encoder.py:159-181).  in_stats = (mean, rstd) of the input (fused normalisation on load).
    drop = (p, seed, channel_mode) fuses the dropout into the epilogue; want_stats additionally returns the
    InstanceNorm statistics of the (dropped) output, reduced inside the same kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, relu, in_stats, mask_own, mask_input, in_scale, drop, want_stats, norm_bwd=0):
        """norm_bwd: 1 = this conv normalises on load and its producer's backward applies the InstanceNorm backward (this one hands its
        un-applied gradient on); 2 = this conv's output feeds such a consumer (its backward expects the hand-on)."""
        dt = x.dtype
        B = x.shape[0]
        cout = weight.shape[0]
        if drop is not None and x.shape[-1] == 1:
            raise RuntimeError("fused dropout is not available on the 1-channel first-layer kernel")
        Ho, Wo = K.conv_out_hw(x.shape[1], x.shape[2], stride)
        ws, slots = K.conv_stat_ws(B, Ho, Wo, cout, x.device) if want_stats else (None, 0)
        y = K.conv3x3(x, wt(weight, dt), bias.omr_phys, stride=stride, relu=relu, in_stats=in_stats, drop=drop,
                      stat_mode=1 if want_stats else 0, stat_ws=ws, stat_slots=slots)
        if relu:
            note_relu(y)
        own_scale = 1.0 / (1.0 - drop[0]) if drop is not None else 1.0
        ctx.cfg = (stride, relu, mask_own, mask_input, in_scale, own_scale)
        ctx.norm_bwd = norm_bwd
        ctx.weight, ctx.bias, ctx.stats = weight, bias, in_stats
        ctx.save_for_backward(x, y if (relu and mask_own) else None)
        if want_stats:
            mean, rstd = K.instnorm_finalize(ws, slots, B, cout, Ho * Wo)
            ctx.mark_non_differentiable(mean, rstd)
            return y, mean, rstd
        return y

    @staticmethod
    def backward(ctx, gy, *unused):
        stride, relu, mask_own, mask_input, in_scale, own_scale = ctx.cfg
        x, y = ctx.saved_tensors
        weight, bias, stats = ctx.weight, ctx.bias, ctx.stats
        g = gy.contiguous()
        pend = _PENDING_NORM.pop(g.data_ptr(), None) if ctx.norm_bwd == 2 else None
        if ctx.norm_bwd == 2 and pend is None:
            raise RuntimeError("Conv3x3Fn: the consumer's backward did not hand on its InstanceNorm gradient")
        one_pass = stats is None and ctx.needs_input_grad[0] and K.conv3x3_bwd_fused_ok(x, g, stride)
        if pend is not None and not (one_pass and pend[5]):
            # nobody to apply it on load: the stand-alone pass
            g = K.instnorm_bwd_apply(g, pend[0], pend[1], pend[2], pend[3], pend[4], relu_mask=pend[5], relu_scale=pend[6])
            pend = None
        if relu and mask_own:
            g = K.relu_bwd(g, y, own_scale)        # y is the stored (dropped) output: (y > 0) * 1/(1-p) is ReLU + dropout backward
        flat = getattr(weight, "omr_flat", None)        # flipped copies are re-laid once per optimizer step (params.FlatParams.refresh_flips)
        if one_pass and (pend is not None or (FUSED_BWD & 1 and g.shape[3] >= FUSED_MIN_COUT)):
            # one pass: data gradient (masked by the ReLU / dropout of the layer below), weight gradient, bias gradient
            wd = flat.flipped(weight, x.dtype) if flat is not None else K.conv3x3_weight_flip(wt(weight, x.dtype))
            if pend is not None:
                K.instnorm_reduce_sums(pend[3], pend[4], x.shape[0], g.shape[3])
            dx = K.conv3x3_bwd_fused(g, x, wd, weight.omr_grad, bias.omr_grad, mask_input, in_scale, norm=pend)
            return (dx,) + (None,) * 11
        # the normalise-on-load conv of the 16-channel block (stride 1): its one-pass form also normalises x in LDS and reduces the
        # InstanceNorm-backward sums
        one_pass_norm = (bool(FUSED_BWD & 4) and stats is not None and ctx.needs_input_grad[0] and tuple(stride) == (1, 1)
                         and x.shape[-1] == 16 and g.shape[-1] == 16 and K.conv3x3_bwd_fused_ok(x, g, stride))
        # ... and of the 32-channel one with stride (2, 2) (ConvBlock 1's conv3)
        one_pass_s2 = (bool(FUSED_BWD & 8) and stats is not None and ctx.needs_input_grad[0] and tuple(stride) == (2, 2) and x.dtype == torch.bfloat16
                       and x.shape[-1] == 32 and g.shape[-1] == 32)
        one_pass_norm = one_pass_norm or one_pass_s2
        # one workgroup per CU with a full LDS ring: little can run beside it, but its ramp-up / drain overlaps the data gradient's
        # (28.37 -> 28.17 ms per C2 step; a shallower ring that leaves LDS for the neighbour loses more than it gains: 28.4)
        if not one_pass_norm:
            WgradStream.run("conv", lambda: K.conv3x3_wgrad(x, g, weight.omr_grad, stride=stride, in_stats=stats, db=bias.omr_grad), x, g, *(stats or ()))
        dx = None
        if ctx.needs_input_grad[0]:
            wd = flat.flipped(weight, x.dtype) if flat is not None else K.conv3x3_weight_flip(wt(weight, x.dtype))
            H, W = x.shape[1], x.shape[2]
            if stats is None:
                dx = K.conv3x3(g, wd, None, stride=(1, 1), dil=stride, out_hw=(H, W), out_mask=x if mask_input else None, mask_scale=in_scale)
            else:
                # data gradient w.r.t. the normalised input with the InstanceNorm-backward sums reduced in its epilogue, then the
                # apply pass (or, TWO_PASS_NORM_BWD, the recomputing form: see the switch's comment)
                ws, slots = K.conv_stat_ws(x.shape[0], H, W, x.shape[3], x.device)
                kw = dict(stride=(1, 1), dil=stride, out_hw=(H, W), stat_ws=ws, stat_slots=slots, stat_x=x, stat_stats=stats)
                if ctx.norm_bwd == 1 or one_pass_norm:
                    if one_pass_s2:
                        dxh = K.conv3x3_bwd_fused_s2(g, x, wd, weight.omr_grad, bias.omr_grad, stats[0], stats[1], ws, slots)
                    elif one_pass_norm:
                        dxh = K.conv3x3_bwd_fused(g, x, wd, weight.omr_grad, bias.omr_grad, False, 1.0, xnorm=(stats[0], stats[1], ws, slots))
                    else:
                        dxh = K.conv3x3(g, wd, None, stat_mode=2, **kw)
                    if ctx.norm_bwd == 1:
                        # the producer conv's one-pass backward applies it on load: hand on dL/dxhat with what the apply needs
                        dx = dxh
                        _PENDING_NORM[dx.data_ptr()] = (x, stats[0], stats[1], ws, slots, mask_input, in_scale)
                    else:
                        dx = K.instnorm_bwd_apply(dxh, x, stats[0], stats[1], ws, slots, relu_mask=mask_input, relu_scale=in_scale)
                elif TWO_PASS_NORM_BWD:
                    K.conv3x3(g, wd, None, stat_mode=4, **kw)
                    K.instnorm_reduce_sums(ws, slots, x.shape[0], x.shape[3])
                    dx = K.conv3x3(g, wd, None, stat_mode=5, relu=mask_input, mask_scale=in_scale, **kw)
                else:
                    dxh = K.conv3x3(g, wd, None, stat_mode=2, **kw)
                    dx = K.instnorm_bwd_apply(dxh, x, stats[0], stats[1], ws, slots, relu_mask=mask_input, relu_scale=in_scale)
        return (dx,) + (None,) * 11


class DwConv3x3Fn(Function):
    """[InstanceNorm ->] depthwise 3x3 + bias on NHWC (DepthSepConv2D.depth_conv, encoder.py:56-64,74)."""

    @staticmethod
    def forward(ctx, x, weight, bias, use_norm, mask_input, in_scale):
        stats = K.instnorm_stats(x) if use_norm else None
        y = K.dwconv3x3(x, wt(weight, x.dtype), bias.omr_phys, in_stats=stats)
        ctx.cfg = (mask_input, in_scale)
        ctx.weight, ctx.bias, ctx.stats = weight, bias, stats
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        mask_input, in_scale = ctx.cfg
        (x,) = ctx.saved_tensors
        weight, bias, stats = ctx.weight, ctx.bias, ctx.stats
        g = gy.contiguous()
        WgradStream.run("depthwise", lambda: K.dwconv3x3_wgrad(x, g, weight.omr_grad, bias.omr_grad, in_stats=stats), x, g, *(stats or ()))
        dx = None
        if ctx.needs_input_grad[0]:
            w = wt(weight, x.dtype)
            if stats is None:
                dx = K.dwconv3x3(g, w, None, flip=True, out_mask=x if mask_input else None, mask_scale=in_scale)
            else:
                dxh = K.dwconv3x3(g, w, None, flip=True)
                dx = K.instnorm_bwd(dxh, x, stats[0], stats[1], relu_mask=mask_input, relu_scale=in_scale)
        return dx, None, None, None, None, None


class LinearFn(Function):
    """y = x W^T + b [-> ReLU] over the last dim (nn.Linear, 1x1 point_conv on NHWC, Conv1d(k=1) head).
    ``rows=(a, b)`` uses rows [a, b) of the weight/bias (packed MHA in_proj: q rows / k|v rows).
    ``out_ld`` > N pads the output row stride (head: vocabulary rounded up to 8 for aligned rows)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, mask_own, out_ld, rows, drop=None):
        w = wt(weight, x.dtype)
        w = w.view(w.shape[0], -1)           # [N_all, K]
        b = None if bias is None else bias.omr_phys
        if rows is not None:
            w = w[rows[0]:rows[1]]
            b = None if b is None else b[rows[0]:rows[1]]
        N, Kd = w.shape
        x2 = x.reshape(-1, Kd)
        M = x2.shape[0]
        if out_ld and out_ld != N:
            buf = torch.empty((M, out_ld), dtype=x.dtype, device=x.device)
            y2 = K.gemm(x2, w, bias=b, relu=relu, out=buf[:, :N])
        else:
            y2 = K.gemm(x2, w, bias=b, relu=relu, drop=drop)
        if drop is not None:
            assert relu and mask_own and not (out_ld and out_ld != N), "fused dropout: ReLU layers whose own backward applies (y > 0) / (1 - p)"
        ctx.cfg = (relu, mask_own, tuple(x.shape), rows, 1.0 / (1.0 - drop[0]) if drop is not None else 1.0)
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x2, y2 if (relu and mask_own) else None)
        if relu:
            note_relu(y2.view(*x.shape[:-1], N))
        return y2.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        relu, mask_own, xshape, rows, own_scale = ctx.cfg
        x2, y2 = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        w = wt(weight, x2.dtype)
        w = w.view(w.shape[0], -1)
        gw = weight.omr_grad.view(w.shape)
        gb = None if bias is None else bias.omr_grad
        if rows is not None:
            w, gw = w[rows[0]:rows[1]], gw[rows[0]:rows[1]]
            gb = None if gb is None else gb[rows[0]:rows[1]]
        N, Kd = w.shape
        g2 = gy.reshape(-1, N)
        if g2.stride(1) != 1 or g2.stride(0) % 8:
            g2 = g2.contiguous()
        if relu and mask_own:
            g2 = K.relu_bwd(g2.contiguous(), y2, own_scale)     # y2 is the stored (dropped) output: ReLU + dropout backward in one mask
        # dW and db in one pass (nothing in backward consumes them): collected and launched together with the other linear
        # layers' at the next flush point, on the side stream (runtime.WgradStream.defer_linear)
        WgradStream.defer_linear(g2, x2, gw, gb)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.gemm(g2, w, trans_b=True).view(xshape)
        return dx, None, None, None, None, None, None, None


def linear(x, weight, bias, relu=False, mask_own=True, out_ld=0, rows=None, drop=None):
    return LinearFn.apply(x, weight, bias, relu, mask_own, out_ld, rows, drop)


class AddFn(Function):
    """Residual add (encoder.py:289)."""

    @staticmethod
    def forward(ctx, a, b):
        return K.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class AddPE2DFn(Function):
    """x + pe[:, :, :h, :w] on NHWC (PositionalEncoding2D.forward, model.py:45-47)."""

    @staticmethod
    def forward(ctx, x, pe_hwc):
        return K.add_pe2d(x, pe_hwc)

    @staticmethod
    def backward(ctx, g):
        return g, None


class DropoutFn(Function):
    """nn.Dropout / nn.Dropout2d with a counter-based mask that is regenerated (not stored) in backward.
    bwd_identity: the consumer applies (x > 0) * 1/(1-p) itself (see module docstring)."""

    @staticmethod
    def forward(ctx, x, p, seed, channel_mode, bwd_identity):
        ctx.cfg = (p, seed, channel_mode, bwd_identity)
        return K.dropout(x.contiguous(), p, seed, channel_mode)

    @staticmethod
    def backward(ctx, g):
        p, seed, channel_mode, bwd_identity = ctx.cfg
        if bwd_identity:
            return g, None, None, None, None
        return K.dropout(g.contiguous(), p, seed, channel_mode), None, None, None, None


# ------------------------------------------------------------------------------------------------ decoder

class EmbedPEFn(Function):
    """embedding(tgt) + pe[:, :T] (decoder.py:124)."""

    @staticmethod
    def forward(ctx, tokens, table, pe, pad_idx, dtype):
        ctx.table, ctx.pad_idx = table, pad_idx
        ctx.save_for_backward(tokens)
        return K.embed_pe(tokens, wt(table, dtype), pe)

    @staticmethod
    def backward(ctx, g):
        (tokens,) = ctx.saved_tensors
        K.embed_bwd(tokens, g.contiguous(), ctx.table.omr_grad, ctx.pad_idx)
        return None, None, None, None, None


class AddLayerNormFn(Function):
    """LayerNorm(dropout(x) + res): the post-norm residual with its sublayer dropout in one kernel
    (torch nn/modules/transformer.py:1146-1154).  drop = (p, seed) or None."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, drop=None):
        p, seed = drop if drop is not None else (0.0, 0)
        out, mean, rstd = K.add_layernorm_fwd(x.contiguous(), res.contiguous(), gamma.omr_phys, beta.omr_phys, drop_p=p, drop_seed=seed)
        ctx.gamma, ctx.beta, ctx.drop = gamma, beta, (p, seed)
        ctx.save_for_backward(x, res, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, res, mean, rstd = ctx.saved_tensors
        p, seed = ctx.drop
        r = K.add_layernorm_bwd(gy.contiguous(), x.contiguous(), res.contiguous(), ctx.gamma.omr_phys, mean, rstd, ctx.gamma.omr_grad, ctx.beta.omr_grad,
                                drop_p=p, drop_seed=seed)
        if p > 0.0:
            ds, dx = r
            return dx, ds, None, None, None
        return r, r, None, None, None


class AttentionFn(Function):
    """Fused attention core over packed projections.
    self mode : qkv [B,T,3d] -> o [B,T,d]           (backward returns d(qkv) packed)
    cross mode: q [B,T,d], kv [B,S,2d] -> o [B,T,d] (backward returns dq and d(kv) packed)"""

    @staticmethod
    def forward(ctx, q_or_qkv, kv, nhead, causal, window, key_bias, blk_lq, blk_lkv, dropout_p, seed):
        if kv is None:
            d = q_or_qkv.shape[-1] // 3
            q, k, v = q_or_qkv[..., :d], q_or_qkv[..., d:2 * d], q_or_qkv[..., 2 * d:]
        else:
            d = q_or_qkv.shape[-1]
            q, k, v = q_or_qkv, kv[..., :d], kv[..., d:]
        # attention-probability dropout: the keep bits of this (layer, step) are generated once, in the word layout all three
        # kernels read (K.attn_dropout_words), and kept for the backward pass
        words = None
        if dropout_p > 0.0:
            B_, T_, S_, dev_ = q.shape[0], q.shape[1], k.shape[1], q.device
            words = WgradStream.prepare(lambda: K.attn_dropout_words(B_, nhead, T_, S_, dropout_p, seed, dev_), dev_)
        o, lse = K.attn_fwd(q, k, v, nhead, causal=causal, window=window, key_bias=key_bias, blk_lq=blk_lq, blk_lkv=blk_lkv,
                            dropout_p=dropout_p, seed=seed, drop_words=words)
        ctx.cfg = (nhead, causal, window, dropout_p, seed, kv is None, d)
        ctx.drop_words = words
        ctx.kv_sink = getattr(kv, "omr_grad_sink", None) if kv is not None else None     # (KVGradSink, layer): see FusedCrossKVFn
        ctx.save_for_backward(q_or_qkv, kv, o, lse, key_bias, blk_lq, blk_lkv)
        return o

    @staticmethod
    def backward(ctx, go):
        nhead, causal, window, dropout_p, seed, packed, d = ctx.cfg
        q_or_qkv, kv, o, lse, key_bias, blk_lq, blk_lkv = ctx.saved_tensors
        go = go.contiguous()
        if packed:
            dqkv = torch.empty_like(q_or_qkv)
            q, k, v = q_or_qkv[..., :d], q_or_qkv[..., d:2 * d], q_or_qkv[..., 2 * d:]
            dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
            dkv = None
        else:
            dqkv = torch.empty_like(q_or_qkv)
            dkv = torch.empty_like(kv) if ctx.kv_sink is None else ctx.kv_sink[0].slot(kv, ctx.kv_sink[1])
            q, k, v = q_or_qkv, kv[..., :d], kv[..., d:]
            dq, dk, dv = dqkv, dkv[..., :d], dkv[..., d:]
        K.attn_bwd(q, k, v, o, go, lse, dq, dk, dv, nhead, causal=causal, window=window, key_bias=key_bias, blk_lq=blk_lq, blk_lkv=blk_lkv,
                   dropout_p=dropout_p, seed=seed, drop_words=ctx.drop_words)
        return dqkv, dkv, None, None, None, None, None, None, None, None


class DecoderLayerFn(Function):
    """One post-norm decoder layer (torch nn/modules/transformer.py:1129-1199 as configured at decoder.py:86-95) as ONE autograd
    node: the eleven kernels of the forward pass and the ~20 of the backward pass are the same launches, in the same order and
    with the same operands, as the per-operation nodes (LinearFn, AttentionFn, AddLayerNormFn) issue -- results are identical to
    the bit -- but the host pays for one node instead of eleven: per-node autograd bookkeeping is most of the host time of the
    small configurations (C1 / C3 / C4 are host-issue bound).  `seeds` = (attn1, drop1, attn2, drop2, ffn, drop3) drawn by the
    caller in the per-operation order (runtime.next_seed), p = the layer's dropout probability (0 in eval mode)."""

    @staticmethod
    def forward(ctx, x, kv, layer, window, self_bias, mem_bias, p, seeds):
        sa, ca = layer.self_attn, layer.multihead_attn
        dt = x.dtype
        B, T, d = x.shape
        S, H, dev = kv.shape[1], sa.num_heads, x.device
        x2 = x.reshape(-1, d)

        def lin(inp2, weight, bias, rows=None, relu=False, drop=None):
            w = wt(weight, dt)
            w = w.view(w.shape[0], -1)
            b = bias.omr_phys
            if rows is not None:
                w, b = w[rows[0]:rows[1]], b[rows[0]:rows[1]]
            return K.gemm(inp2, w, bias=b, relu=relu, drop=drop)

        def words(Tq, Sk, seed):
            return WgradStream.prepare(lambda: K.attn_dropout_words(B, H, Tq, Sk, p, seed, dev), dev) if p > 0.0 else None

        drop = (lambda i: (p, seeds[i])) if p > 0.0 else (lambda i: None)
        # self-attention block
        qkv = lin(x2, sa.in_proj_weight, sa.in_proj_bias).view(B, T, 3 * d)
        w1 = words(T, T, seeds[0])
        o1, lse1 = K.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, causal=True, window=window, key_bias=self_bias,
                              dropout_p=p, seed=seeds[0], drop_words=w1)
        a1 = lin(o1.view(-1, d), sa.out_proj.weight, sa.out_proj.bias).view(B, T, d)
        x1, mean1, rstd1 = K.add_layernorm_fwd(a1, x, layer.norm1.weight.omr_phys, layer.norm1.bias.omr_phys, drop_p=p, drop_seed=seeds[1])
        # cross-attention block
        q2 = lin(x1.view(-1, d), ca.in_proj_weight, ca.in_proj_bias, rows=(0, d)).view(B, T, d)
        w2 = words(T, S, seeds[2])
        o2, lse2 = K.attn_fwd(q2, kv[..., :d], kv[..., d:], H, causal=False, window=-1, key_bias=mem_bias, dropout_p=p, seed=seeds[2], drop_words=w2)
        a2 = lin(o2.view(-1, d), ca.out_proj.weight, ca.out_proj.bias).view(B, T, d)
        x2n, mean2, rstd2 = K.add_layernorm_fwd(a2, x1, layer.norm2.weight.omr_phys, layer.norm2.bias.omr_phys, drop_p=p, drop_seed=seeds[3])
        # feed-forward block (ReLU + dropout in the first GEMM's epilogue)
        h = lin(x2n.view(-1, d), layer.linear1.weight, layer.linear1.bias, relu=True, drop=drop(4))
        note_relu(h.view(B, T, -1))
        f = lin(h, layer.linear2.weight, layer.linear2.bias).view(B, T, d)
        y, mean3, rstd3 = K.add_layernorm_fwd(f, x2n, layer.norm3.weight.omr_phys, layer.norm3.bias.omr_phys, drop_p=p, drop_seed=seeds[5])
        ctx.layer, ctx.cfg = layer, (window, p, seeds, H, d)
        ctx.kv_sink = getattr(kv, "omr_grad_sink", None)
        ctx.words = (w1, w2)
        ctx.save_for_backward(x, kv, self_bias, mem_bias, qkv, o1, lse1, a1, x1, mean1, rstd1, q2, o2, lse2, a2, x2n, mean2, rstd2, h, f, mean3, rstd3)
        return y

    @staticmethod
    def backward(ctx, gy):
        layer = ctx.layer
        window, p, seeds, H, d = ctx.cfg
        x, kv, self_bias, mem_bias, qkv, o1, lse1, a1, x1, mean1, rstd1, q2, o2, lse2, a2, x2n, mean2, rstd2, h, f, mean3, rstd3 = ctx.saved_tensors
        sa, ca = layer.self_attn, layer.multihead_attn
        dt = x.dtype
        B, T, _ = x.shape

        def ln_bwd(g, branch, res, norm, mean, rstd, seed):
            r = K.add_layernorm_bwd(g.contiguous(), branch, res, norm.weight.omr_phys, mean, rstd, norm.weight.omr_grad, norm.bias.omr_grad,
                                    drop_p=p, drop_seed=seed)
            if p > 0.0:
                ds, dbranch = r                        # K.add_layernorm_bwd: (gradient of the residual input, gradient of the dropped branch)
                return dbranch, ds
            return r, r                                # (gradient of the sub-layer output, gradient of the residual input)

        def lin_bwd(g2, inp2, weight, bias, rows=None):
            w = wt(weight, dt)
            w = w.view(w.shape[0], -1)
            gw, gb = weight.omr_grad.view(w.shape), bias.omr_grad
            if rows is not None:
                w, gw, gb = w[rows[0]:rows[1]], gw[rows[0]:rows[1]], gb[rows[0]:rows[1]]
            WgradStream.defer_linear(g2, inp2, gw, gb)
            return K.gemm(g2, w, trans_b=True)

        # feed-forward block
        df, dres = ln_bwd(gy, f, x2n, layer.norm3, mean3, rstd3, seeds[5])
        dh = lin_bwd(df.view(-1, d), h, layer.linear2.weight, layer.linear2.bias)
        g1 = K.relu_bwd(dh, h, 1.0 / (1.0 - p) if p > 0.0 else 1.0)
        dx2 = dres + lin_bwd(g1, x2n.view(-1, d), layer.linear1.weight, layer.linear1.bias).view(B, T, d)
        # cross-attention block
        da2, dres = ln_bwd(dx2, a2, x1, layer.norm2, mean2, rstd2, seeds[3])
        do2 = lin_bwd(da2.view(-1, d), o2.view(-1, d), ca.out_proj.weight, ca.out_proj.bias).view(B, T, d)
        dq2 = torch.empty_like(q2)
        dkv = torch.empty_like(kv) if ctx.kv_sink is None else ctx.kv_sink[0].slot(kv, ctx.kv_sink[1])
        K.attn_bwd(q2, kv[..., :d], kv[..., d:], o2, do2, lse2, dq2, dkv[..., :d], dkv[..., d:], H, causal=False, window=-1, key_bias=mem_bias,
                   dropout_p=p, seed=seeds[2], drop_words=ctx.words[1])
        dx1 = dres + lin_bwd(dq2.view(-1, d), x1.view(-1, d), ca.in_proj_weight, ca.in_proj_bias, rows=(0, d)).view(B, T, d)
        # self-attention block
        da1, dres = ln_bwd(dx1, a1, x, layer.norm1, mean1, rstd1, seeds[1])
        do1 = lin_bwd(da1.view(-1, d), o1.view(-1, d), sa.out_proj.weight, sa.out_proj.bias).view(B, T, d)
        dqkv = torch.empty_like(qkv)
        K.attn_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o1, do1, lse1, dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:], H,
                   causal=True, window=window, key_bias=self_bias, dropout_p=p, seed=seeds[0], drop_words=ctx.words[0])
        dx = dres + lin_bwd(dqkv.view(-1, 3 * d), x.reshape(-1, d), sa.in_proj_weight, sa.in_proj_bias).view(B, T, d)
        return dx, dkv, None, None, None, None, None, None


class KVGradSink:
    """Shared landing buffer for the K|V gradients of all decoder layers: each layer's attention backward writes its
    [B,S,2d] gradient straight into its column block of one [B,S,L*2d] tensor, so FusedCrossKVFn.backward can run ONE
    weight-gradient GEMM and ONE data-gradient GEMM over it -- no per-layer gradient tensors, no gradient adds."""

    def __init__(self):
        self.buf = None

    def slot(self, kv: Tensor, layer: int) -> Tensor:
        width = kv.shape[-1]
        if self.buf is None:
            self.buf = torch.empty((kv.shape[0], kv.shape[1], kv.stride(1)), dtype=kv.dtype, device=kv.device)
        return self.buf[..., layer * width:(layer + 1) * width]

    def take(self):
        buf, self.buf = self.buf, None
        return buf


class FusedCrossKVFn(Function):
    """K|V projections of the encoder memory for ALL decoder layers in one GEMM (nn.MultiheadAttention in_proj rows
    [d, 3d) of every layer, torch nn/functional.py multi_head_attention_forward / decoder.py:86-95): the memory is the same
    tensor in every layer, so it is read once instead of L times, and its gradient is one GEMM over the concatenated K|V
    gradients instead of L GEMMs plus L-1 gradient adds.  `pack` holds row-group views of the layers' packed in_proj
    parameters where they lie back to back in the flat buffer (params._placement_order).  Returns L views [B,S,2d] of one
    [B,S,L*2d] buffer."""

    @staticmethod
    def forward(ctx, memory, pack, sink):
        L, d = pack["L"], pack["d"]
        mem2 = memory.reshape(-1, d)
        Rm = mem2.shape[0]
        out = torch.empty((Rm, L * 2 * d), dtype=memory.dtype, device=memory.device)
        K.gemm_row_groups(mem2, pack["w"], out, Rm, L * 2 * d, d, bias=pack["b"], group=(2 * d, 3 * d, d, 1))
        ctx.pack, ctx.sink, ctx.mshape = pack, sink, tuple(memory.shape)
        ctx.save_for_backward(mem2)
        kv_all = out.view(memory.shape[0], memory.shape[1], L * 2 * d)
        return tuple(kv_all[..., l * 2 * d:(l + 1) * 2 * d] for l in range(L))

    @staticmethod
    def backward(ctx, *grads):
        pack, (mem2,) = ctx.pack, ctx.saved_tensors
        L, d = pack["L"], pack["d"]
        Rm = mem2.shape[0]
        buf = ctx.sink.take()
        if buf is None:
            buf = torch.empty((ctx.mshape[0], ctx.mshape[1], L * 2 * d), dtype=mem2.dtype, device=mem2.device)
        for l, g in enumerate(grads):           # anything that did not land in the sink (or a missing gradient) is copied in
            dst = buf[..., l * 2 * d:(l + 1) * 2 * d]
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)
        g2 = buf.view(Rm, L * 2 * d)
        WgradStream.defer_linear(g2, mem2, pack["gw"], pack["gb"], group=(2 * d, 3 * d, d))
        dmem = None
        if ctx.needs_input_grad[0]:
            dmem = torch.empty((Rm, d), dtype=mem2.dtype, device=mem2.device)
            K.gemm_row_groups(g2, pack["w"], dmem, Rm, d, L * 2 * d, trans_b=True, group=(2 * d, 3 * d, d, 2))
            dmem = dmem.view(ctx.mshape)
        return dmem, None, None


class CrossEntropyFn(Function):
    """CrossEntropyLoss(ignore_index) on row-major logits [M, V] (model.py:109,166)."""

    @staticmethod
    def forward(ctx, logits2d, target, V, pad_idx):
        loss, lse, acc2 = K.ce_fwd(logits2d, target, V, pad_idx)
        ctx.cfg = (V, pad_idx)
        ctx.save_for_backward(logits2d, target, lse, acc2)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        V, pad_idx = ctx.cfg
        logits2d, target, lse, acc2 = ctx.saved_tensors
        dl = K.ce_bwd(logits2d, target, lse, acc2, V, pad_idx, grad_out=g)
        return dl, None, None, None


def cross_entropy(logits_bvt: Tensor, target_bt: Tensor, pad_idx: int) -> Tensor:
    """Loss on logits in the reference's [B, V, T] layout.  The decoder produces them as a stride
    permutation of row-major [B*T, V], which is what the kernel consumes (no copy)."""
    B, V, T = logits_bvt.shape
    rows = logits_bvt.permute(0, 2, 1)
    if rows.stride(2) != 1 or rows.stride(1) % 8 or rows.stride(0) != T * rows.stride(1):
        rows = rows.contiguous()
    l2 = rows.reshape(B * T, V)   # a view: rows of one batch follow each other at the same pitch
    return CrossEntropyFn.apply(l2, target_bt.reshape(-1).contiguous(), V, pad_idx)
