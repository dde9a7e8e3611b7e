"""Parameter shape tables, seeded initialisation and synthetic batches (host side, pure torch).

Shape tables follow the reference's state-dict key contract (SURVEY.md section 5;
split_multimodal_ckpt.py:45-59; model.py:94-100,409-438).  Synthetic batches follow the collate
contract of src/data/preprocessing.py:85-144 and SURVEY.md section 8(d).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

Shapes = "OrderedDict[str, Tuple[int, ...]]"

CONV_PLAN = ((None, 16, (1, 1)), (16, 32, (2, 2)), (32, 64, (2, 2)), (64, 128, (2, 2)), (128, 128, (2, 1)))  # encoder.py:255-259
HEIGHT_REDUCTION = 16  # encoder.py:8
WIDTH_REDUCTION = 8    # encoder.py:9

PAD_TOKEN, SOS_TOKEN, EOS_TOKEN = "<PAD>", "<sos>", "<eos>"  # ar_dataset.py:21-23
GRANDSTAFF_VOCAB = 6997  # grandstaff/vocabs/ar_w2i_kern.json
GRANDSTAFF_SOS, GRANDSTAFF_EOS = 6836, 6835


def encoder_shapes(prefix: str, in_channels: int = 1, out_channels: int = 256) -> "Shapes":
    """Parameter names/shapes of Encoder (encoder.py:241-269), registration order."""
    s: "Shapes" = OrderedDict()
    for i, (cin, cout, _) in enumerate(CONV_PLAN):
        cin = in_channels if cin is None else cin
        for j, c in enumerate((cin, cout, cout), start=1):
            s[f"{prefix}conv_blocks.{i}.conv{j}.weight"] = (cout, c, 3, 3)
            s[f"{prefix}conv_blocks.{i}.conv{j}.bias"] = (cout,)
    for i in range(4):
        cout = 128 if i < 3 else out_channels  # encoder.py:264-267
        for j, c in enumerate((128, cout, cout), start=1):
            p = f"{prefix}dscblocks.{i}.conv{j}."
            s[p + "depth_conv.weight"] = (c, 1, 3, 3)
            s[p + "depth_conv.bias"] = (c,)
            s[p + "point_conv.weight"] = (cout, c, 1, 1)
            s[p + "point_conv.bias"] = (cout,)
    return s


def mha_shapes(prefix: str, d: int) -> "Shapes":
    return OrderedDict([
        (prefix + "in_proj_weight", (3 * d, d)), (prefix + "in_proj_bias", (3 * d,)),
        (prefix + "out_proj.weight", (d, d)), (prefix + "out_proj.bias", (d,)),
    ])


def decoder_shapes(prefix: str, vocab: int, d: int = 256, ff: int = 256, layers: int = 8, out_size: Optional[int] = None) -> "Shapes":
    """Parameter names/shapes of Decoder (decoder.py:72-102), registration order."""
    out_size = vocab if out_size is None else out_size
    s: "Shapes" = OrderedDict()
    s[prefix + "embedding.weight"] = (vocab, d)
    for i in range(layers):
        p = f"{prefix}transformer_decoder.layers.{i}."
        s.update(mha_shapes(p + "self_attn.", d))
        s.update(mha_shapes(p + "multihead_attn.", d))
        s[p + "linear1.weight"] = (ff, d)
        s[p + "linear1.bias"] = (ff,)
        s[p + "linear2.weight"] = (d, ff)
        s[p + "linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3"):
            s[p + n + ".weight"] = (d,)
            s[p + n + ".bias"] = (d,)
    s[prefix + "out_layer.weight"] = (out_size, d, 1)
    s[prefix + "out_layer.bias"] = (out_size,)
    return s


def transformer_shapes(vocab: int, d: int = 256, ff: int = 256, layers: int = 8) -> "Shapes":
    s = encoder_shapes("encoder.", 1, d)
    s.update(decoder_shapes("decoder.", vocab, d, ff, layers))
    return s


def multimodal_shapes(vocab: int, mixer_type: str, d: int = 256, ff: int = 256, layers: int = 8) -> "Shapes":
    s = encoder_shapes("image_encoder.", 1, d)
    s.update(encoder_shapes("audio_encoder.", 1, d))
    s.update(decoder_shapes("decoder.", vocab, d, ff, layers))
    if mixer_type != "concat":
        s.update(mha_shapes("cross_attn.attention.", d))
    return s


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    return g


def seeded_state_dict(shapes: "Shapes", seed: int, mode: str = "test") -> Dict[str, torch.Tensor]:
    """Deterministic fp32 weights, independent of construction order (each tensor has its own
    generator keyed by (seed, name)).

    mode="torch_default": the distributions torch's default init gives the reference modules
      (SURVEY.md Appendix A "Initialisation"): Conv/Linear weight and bias U(+-1/sqrt(fan_in)),
      MHA in_proj_weight Xavier-uniform with zero in_proj_bias/out_proj.bias, LayerNorm 1/0,
      Embedding N(0,1) with row 0 (PAD) zero.
    mode="test": same weight distributions, but every bias is non-zero and LayerNorm
      gains/offsets are perturbed so parity tests exercise those terms.
    """
    sd: Dict[str, torch.Tensor] = {}
    fan_in_of: Dict[str, int] = {}
    for name, shape in shapes.items():
        g = _gen(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("embedding.weight"):
            t = torch.randn(shape, generator=g)
            t[0].zero_()
        elif "norm" in name.rsplit(".", 2)[-2] and len(shape) == 1:
            if mode == "test":
                u = torch.rand(shape, generator=g) * 2 - 1
                t = (1.0 + 0.2 * u) if leaf == "weight" else 0.2 * u
            else:
                t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif leaf == "in_proj_weight":
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif len(shape) >= 2:
            fan_in = int(math.prod(shape[1:]))
            fan_in_of[name.rsplit(".", 1)[0]] = fan_in
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:  # biases
            owner = name.rsplit(".", 1)[0]
            is_mha_bias = leaf == "in_proj_bias" or name.endswith("out_proj.bias")
            if is_mha_bias and mode == "torch_default":
                t = torch.zeros(shape)
            else:
                fan_in = fan_in_of.get(owner, shape[0])
                t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        sd[name] = t.contiguous()
    return sd


def make_vocab(size: int) -> Tuple[Dict[str, int], Dict[int, str]]:
    """A w2i/i2w pair with the reference's id layout: PAD=0, the rest sorted lexicographically
    with <eos> before <sos> (ar_dataset.py:323-332)."""
    words = sorted([f"tok{i:05d}" for i in range(size - 3)] + [SOS_TOKEN, EOS_TOKEN])
    w2i = {PAD_TOKEN: 0}
    for i, w in enumerate(words, start=1):
        w2i[w] = i
    return w2i, {i: w for w, i in w2i.items()}


def synthetic_unimodal_batch(batch: int, height: int, width: int, seq_len: int, vocab: int, sos: int, eos: int,
                             seed: int, pad_value: float = 1.0, ragged: bool = True):
    """A batch with the collate contract of ar_batch_preparation_unimodal (preprocessing.py:85-103)
    built as SURVEY.md section 8(d) prescribes: x ~ U[0,1) f32 [B,1,H,W] right-padded with pad_value
    beyond each sample's true width, xl = ceil(H/16)*ceil(w_i/8) (ar_dataset.py:439-442) int32,
    y = [sos] + toks + [eos] with y_in=y[:-1], y_out=y[1:], 0-padded to seq_len (int64)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.rand((batch, 1, height, width), generator=g)
    xl = torch.empty(batch, dtype=torch.int32)
    y_in = torch.zeros((batch, seq_len), dtype=torch.int64)
    y_out = torch.zeros((batch, seq_len), dtype=torch.int64)
    hp = math.ceil(height / HEIGHT_REDUCTION)
    for i in range(batch):
        wi = int(torch.randint(width // 2, width + 1, (1,), generator=g)) if ragged else width
        if i == 0:
            wi = width  # the collate pads to the widest sample, so one sample is full width
        x[i, :, :, wi:] = pad_value
        xl[i] = hp * math.ceil(wi / WIDTH_REDUCTION)
        ti = int(torch.randint(seq_len // 2, seq_len + 1, (1,), generator=g)) if ragged else seq_len
        if i == 0:
            ti = seq_len
        toks = torch.randint(1, vocab, (ti - 1,), generator=g)
        toks[(toks == sos) | (toks == eos)] = 1
        y = torch.cat([torch.tensor([sos]), toks, torch.tensor([eos])])
        y_in[i, :ti] = y[:-1]
        y_out[i, :ti] = y[1:]
    return x, xl, y_in, y_out


def seeded_dropout_mask(seed: int, site: int, p: float, shape: Tuple[int, ...], channel: bool = False) -> torch.Tensor:
    """Multiplicative dropout mask keep / (1 - p) for the site-th dropout call of a forward pass, a pure function of
    (seed, site): the golden generator injects it into the reference's nn.Dropout / nn.Dropout2d / attention dropout
    (tests/golden/gen_golden_r2.py) and the oracle tests inject the same mask into oracle.ref_cpu.DropPlan.
    channel=True: one decision per (b, c) of a [B, C, ...] tensor (nn.Dropout2d).  The random numbers are drawn over the
    flat element count, so [B*H, T, S] and [B, H, T, S] views of attention probabilities get the same mask."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed * 7919 + site)
    if channel:
        shape = tuple(shape[:2]) + (1,) * (len(shape) - 2)
    n = int(math.prod(shape))
    keep = torch.rand(n, generator=g, dtype=torch.float32) >= p
    return keep.to(torch.float32).view(shape) / (1.0 - p)
