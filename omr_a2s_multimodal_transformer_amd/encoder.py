"""Convolutional encoder on HIP kernels -- same classes, constructor signatures, attribute and state-dict
names as the reference's src/transformer/encoder.py; activations are NHWC internally and every conv /
norm / dropout is a hand-written gfx950 kernel (csrc/conv.hip, norm.hip, gemm.hip, elementwise.hip).
"""
from __future__ import annotations

import math
import random
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .runtime import next_seed

HEIGHT_REDUCTION = 16  # encoder.py:8
WIDTH_REDUCTION = 8    # encoder.py:9


class Conv2d(nn.Module):
    """Parameter holder with nn.Conv2d's names/shapes and default init (kaiming_uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)))."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Tuple[int, int], stride=(1, 1), padding=(0, 0), groups: int = 1):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, tuple(kernel_size)
        self.stride, self.padding, self.groups = tuple(stride), tuple(padding), groups
        fan_in = (in_channels // groups) * kernel_size[0] * kernel_size[1]
        bound = 1.0 / math.sqrt(fan_in)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *kernel_size).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))


def _pair(v) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else tuple(v)


class MixDropout(nn.Module):
    """encoder.py:87-104: with probability 0.5 nn.Dropout(dropout_prob) else nn.Dropout2d(dropout_2d_prob).
    The choice consumes one Python `random.random()` draw per call, exactly like the reference (also in
    eval mode, where it is the identity)."""

    def __init__(self, dropout_prob: float = 0.4, dropout_2d_prob: float = 0.2):
        super().__init__()
        self.dropout_prob, self.dropout_2d_prob = dropout_prob, dropout_2d_prob

    def pick(self) -> Tuple[float, bool]:
        return (self.dropout_prob, False) if random.random() < 0.5 else (self.dropout_2d_prob, True)

    def apply_nhwc(self, x: torch.Tensor, fused_bwd: bool) -> Tuple[torch.Tensor, float]:
        """Returns (y, 1/(1-p)).  fused_bwd: the consumer's data-gradient epilogue applies (y>0)*scale."""
        p, channel_mode = self.pick()
        if not self.training or p <= 0.0:
            return x, 1.0
        return Fn.DropoutFn.apply(x, p, next_seed("nhwc", p, channel_mode), channel_mode, fused_bwd), 1.0 / (1.0 - p)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # logical NCHW in, NCHW out (API parity)
        y, _ = self.apply_nhwc(x.permute(0, 2, 3, 1).contiguous(), False)
        return y.permute(0, 3, 1, 2)


class DepthSepConv2D(nn.Module):
    """encoder.py:12-84 restricted to what the reference instantiates: odd kernel (3,3), padding True or
    (1,1), stride (1,1), dilation 1, no activation."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Tuple[int, int], activation: Optional[nn.Module] = None,
                 padding: Union[bool, Tuple[int, int]] = True, stride: Union[int, Tuple[int, int]] = (1, 1),
                 dilation: Union[int, Tuple[int, int]] = (1, 1)):
        super().__init__()
        if tuple(kernel_size) != (3, 3) or _pair(stride) != (1, 1) or _pair(dilation) != (1, 1) or activation is not None or not padding:
            raise NotImplementedError("HIP DepthSepConv2D covers the reference's configuration: 3x3, stride 1, pad 1, no activation")
        self.padding = None
        self.activation = activation
        self.depth_conv = Conv2d(in_channels, in_channels, (3, 3), padding=(1, 1), groups=in_channels)
        self.point_conv = Conv2d(in_channels, out_channels, (1, 1))

    def nhwc(self, x: torch.Tensor, use_norm: bool, mask_input: bool, in_scale: float, relu: bool, mask_own: bool) -> torch.Tensor:
        t = Fn.DwConv3x3Fn.apply(x, self.depth_conv.weight, self.depth_conv.bias, use_norm, mask_input, in_scale)
        return Fn.linear(t, self.point_conv.weight, self.point_conv.bias, relu=relu, mask_own=mask_own)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.nhwc(_to_nhwc(x), False, False, 1.0, False, True).permute(0, 3, 1, 2)


class ConvBlock(nn.Module):
    """encoder.py:107-181: conv1-ReLU-[drop]-conv2-ReLU-[drop]-InstanceNorm-conv3(stride)-ReLU-[drop]."""

    def __init__(self, in_c: int, out_c: int, stride: Union[int, Tuple[int, int]] = (1, 1), kernel: int = 3,
                 activation: Optional[nn.Module] = None, dropout: float = 0.5):
        super().__init__()
        if kernel != 3:
            raise NotImplementedError("HIP ConvBlock implements the reference's kernel=3")
        self.stride = _pair(stride)
        self.conv1 = Conv2d(in_c, out_c, (3, 3), padding=(1, 1))
        self.conv2 = Conv2d(out_c, out_c, (3, 3), padding=(1, 1))
        self.conv3 = Conv2d(out_c, out_c, (3, 3), padding=(1, 1), stride=self.stride)
        self.dropout = MixDropout(dropout_prob=dropout, dropout_2d_prob=dropout / 2)

    def _drop(self):
        """MixDropout choice for the conv kernel's fused epilogue: (p, seed, channel_mode) or None (eval / p = 0).
        Consumes one `random.random()` like MixDropout.forward (encoder.py:102), in eval mode too."""
        p, channel_mode = self.dropout.pick()
        if not self.training or p <= 0.0:
            return None
        return (p, next_seed("nhwc", p, channel_mode), channel_mode)

    def nhwc(self, x: torch.Tensor, in_mask: bool, in_scale: float, defer_out: bool) -> Tuple[torch.Tensor, float]:
        """x NHWC.  in_mask/in_scale: x is a ReLU(+dropout) output whose activation backward this block must
        apply.  defer_out: the consumer of the returned tensor applies OUR final ReLU(+dropout) backward;
        returns (y, scale) with scale = 1/(1-p) of a dropout applied after conv3 (else 1).
        Every conv is ONE kernel: bias, ReLU, the MixDropout of this position, the InstanceNorm statistics (conv2) and
        the InstanceNorm apply (conv3's loads) are fused into it."""
        pos = random.randint(1, 3)  # encoder.py:160 (drawn in eval mode too)
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        d1 = self._drop() if pos == 1 else None
        if d1 is not None and x.shape[-1] == 1:      # 1-channel first layer: direct kernel without the fused dropout
            x = Fn.Conv3x3Fn.apply(x, c1.weight, c1.bias, (1, 1), True, None, False, in_mask, in_scale, None, False)
            x = Fn.DropoutFn.apply(x, d1[0], d1[1], d1[2], True)
        else:
            x = Fn.Conv3x3Fn.apply(x, c1.weight, c1.bias, (1, 1), True, None, False, in_mask, in_scale, d1, False)
        s1 = 1.0 / (1.0 - d1[0]) if d1 is not None else 1.0
        d2 = self._drop() if pos == 2 else None
        # conv2 <- InstanceNorm <- conv3 in backward: where the one-pass conv backward exists (bf16, 16 / 32 channels), conv3 hands its
        # un-applied gradient on and conv2's backward applies the InstanceNorm backward while it loads (functional.FUSED_BWD)
        hand_on = bool(Fn.FUSED_BWD & 2) and torch.is_grad_enabled() and x.requires_grad and x.dtype == torch.bfloat16 and x.shape[-1] in Fn.FUSED_NORM_CHANNELS
        x, mean, rstd = Fn.Conv3x3Fn.apply(x, c2.weight, c2.bias, (1, 1), True, None, False, True, s1, d2, True, 2 if hand_on else 0)
        s2 = 1.0 / (1.0 - d2[0]) if d2 is not None else 1.0
        d3 = self._drop() if pos == 3 else None
        x = Fn.Conv3x3Fn.apply(x, c3.weight, c3.bias, self.stride, True, (mean, rstd), not defer_out, True, s2, d3, False, 1 if hand_on else 0)
        s3 = 1.0 / (1.0 - d3[0]) if d3 is not None else 1.0
        return x, s3

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y, _ = self.nhwc(_to_nhwc(x), False, 1.0, False)
        return y.permute(0, 3, 1, 2)


class DSCBlock(nn.Module):
    """encoder.py:184-238: dsc1-ReLU-[drop]-dsc2-ReLU-[drop]-InstanceNorm-dsc3-[drop] (no ReLU after dsc3)."""

    def __init__(self, in_c: int, out_c: int, stride: Union[int, Tuple[int, int]] = (2, 1), activation: Optional[nn.Module] = None,
                 dropout: float = 0.5):
        super().__init__()
        if _pair(stride) != (1, 1):
            raise NotImplementedError("HIP DSCBlock implements stride (1,1), the only value the reference Encoder uses (encoder.py:264-267)")
        self.conv1 = DepthSepConv2D(in_c, out_c, kernel_size=(3, 3))
        self.conv2 = DepthSepConv2D(out_c, out_c, kernel_size=(3, 3))
        self.conv3 = DepthSepConv2D(out_c, out_c, kernel_size=(3, 3), padding=(1, 1), stride=(1, 1))
        self.dropout = MixDropout(dropout_prob=dropout, dropout_2d_prob=dropout / 2)

    def nhwc(self, x: torch.Tensor) -> torch.Tensor:
        pos = random.randint(1, 3)  # encoder.py:219
        x = self.conv1.nhwc(x, False, False, 1.0, True, False)
        s = 1.0
        if pos == 1:
            x, s = self.dropout.apply_nhwc(x, True)
        x = self.conv2.nhwc(x, False, True, s, True, False)
        s = 1.0
        if pos == 2:
            x, s = self.dropout.apply_nhwc(x, True)
        x = self.conv3.nhwc(x, True, True, s, False, True)
        if pos == 3:
            x, _ = self.dropout.apply_nhwc(x, False)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.nhwc(_to_nhwc(x)).permute(0, 3, 1, 2)


def _to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW -> NHWC tensor (free for C == 1 and for channels_last-strided inputs)."""
    if x.shape[1] == 1:
        return x.contiguous().view(x.shape[0], x.shape[2], x.shape[3], 1)
    return x.permute(0, 2, 3, 1).contiguous()


class Encoder(nn.Module):
    """encoder.py:241-291.  forward(x [B,in_channels,H,W] fp32) -> [B, out_channels, ceil(H/16), ceil(W/8)] as a
    channels_last-strided tensor (its NHWC memory is the decoder's [B, S, C] memory, model.py:147).
    `out_channels` (reference: 256, encoder.py:267) and `compute_dtype` are this build's extra knobs."""

    def __init__(self, in_channels: int, dropout: float = 0.5, out_channels: int = 256):
        super().__init__()
        self.conv_blocks = nn.ModuleList([
            ConvBlock(in_c=in_channels, out_c=16, stride=(1, 1), dropout=dropout),
            ConvBlock(in_c=16, out_c=32, stride=(2, 2), dropout=dropout),
            ConvBlock(in_c=32, out_c=64, stride=(2, 2), dropout=dropout),
            ConvBlock(in_c=64, out_c=128, stride=(2, 2), dropout=dropout),
            ConvBlock(in_c=128, out_c=128, stride=(2, 1), dropout=dropout),
        ])
        self.dscblocks = nn.ModuleList([
            DSCBlock(in_c=128, out_c=128, stride=(1, 1), dropout=dropout),
            DSCBlock(in_c=128, out_c=128, stride=(1, 1), dropout=dropout),
            DSCBlock(in_c=128, out_c=128, stride=(1, 1), dropout=dropout),
            DSCBlock(in_c=128, out_c=out_channels, stride=(1, 1), dropout=dropout),
        ])

    def forward_nhwc(self, x: torch.Tensor, compute_dtype: torch.dtype) -> torch.Tensor:
        x = _to_nhwc(x)
        if x.dtype != compute_dtype:
            x = K.cast(x, compute_dtype)
        n = len(self.conv_blocks)
        mask, scale = False, 1.0
        for i, blk in enumerate(self.conv_blocks):
            x, scale = blk.nhwc(x, mask, scale, defer_out=(i < n - 1))
            mask = True
        if torch.is_grad_enabled() and x.requires_grad:
            # backward reaches this point when the DSC blocks are done: their collected 1x1-conv weight gradients start now, on the side
            # stream under the ConvBlocks' backward (at the end of the pass they were 0.24 ms of exposed tail in front of Adam)
            from .ddp import GradBoundary
            x = GradBoundary.apply(None, (), x)
        for blk in self.dscblocks:
            xt = blk.nhwc(x)
            x = Fn.AddFn.apply(x, xt) if x.shape == xt.shape else xt  # encoder.py:289
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = torch.bfloat16 if getattr(self.conv_blocks[0].conv1.weight, "omr_lowp", None) is not None else torch.float32
        if not hasattr(self.conv_blocks[0].conv1.weight, "omr_phys"):
            raise RuntimeError("parameters are not on the GPU flat buffers yet: call flatten_parameters() on the owning model")
        return self.forward_nhwc(x, dt).permute(0, 3, 1, 2)
