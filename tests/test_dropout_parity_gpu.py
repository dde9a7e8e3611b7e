"""TRAIN-MODE (dropout ON) parity of the HIP path against the CPU oracle, by mask injection.

The HIP path never stores a dropout mask: every site's mask is a pure function of (seed, element index) that the forward
kernel applies and the backward kernels regenerate (fused conv epilogues, add+LayerNorm, FFN GEMM epilogue, attention
probabilities, stand-alone omr_dropout).  runtime.trace_dropout() records the (kind, p, seed, mode) of every site of a
forward pass; each mask is then MATERIALISED through an independent entry point (omr_dropout on a tensor of ones,
omr_attn_dropout_mask) and injected into oracle.ref_cpu.DropPlan, whose train-mode semantics are pinned against the
reference itself (tests/golden/f12_dropout.npz, tests/test_oracle_golden_r2.py).  Forward, loss and EVERY parameter
gradient are compared in fp32 at the north-star tolerance, 1e-3 relative (L2 per tensor).

ReLU masks.  Between two correct fp32 implementations the gradient of this network is reproducible only to ~1e-3..1e-2:
a ReLU pre-activation within rounding noise of zero lands on different sides, and ONE such element moves its layer's
gradient -- and every tensor upstream -- by ~1/sqrt(N) of its norm (6e-3 for one element of a 3x128x4x20 map).  Measured
between the reference and the oracle themselves, both torch CPU: 2.5e-3 (tools/relu_flip_probe.py), while the HIP forward
pass is as close to the fp64 result as the CPU fp32 one (tools/fwd_error_probe.py: 3e-6 after nine blocks).  The gradient
check therefore fixes the piecewise-linear region: runtime.trace_relu() records the HIP run's ReLU masks (stored output > 0)
and the oracle takes its gradient with those masks (DropPlan(relu_fn=...)) in fp64.  That this is the same function is
checked too: the plain oracle (its own ReLUs) reproduces the HIP logits to 1e-3, and the injected masks differ from the
oracle's own on a vanishing fraction of elements, all of them with |pre-activation| at rounding level.
"""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import synthetic as syn  # noqa: E402
from omr_a2s_multimodal_transformer_amd.config import ModelConfig  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda:0"


def materialise(trace, relu_trace):
    """The HIP run's dropout masks (from (kind, p, seed, mode), through omr_dropout / omr_attn_dropout_mask) and ReLU masks
    as CPU tensors in the oracle's layouts; dropout masks are built lazily because their shapes come from the oracle."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    cache = {}

    def drop_mask(site, kind, p, shape, channel):
        tkind, tp, seed, tch = trace[site]
        assert tkind == kind and abs(tp - p) < 1e-12 and tch == channel, (site, trace[site], kind, p, channel)
        if site not in cache:
            if kind == "attn":
                B, H, T, S = shape
                m = K.attn_dropout_mask(B, H, T, S, p, seed, DEV).float() / (1.0 - p)
            elif kind == "nhwc":
                B, C, Hh, W = shape
                m = K.dropout(torch.ones((B, Hh, W, C), device=DEV, dtype=torch.float32), p, seed, channel).permute(0, 3, 1, 2)
            else:
                m = K.dropout(torch.ones(shape, device=DEV, dtype=torch.float32), p, seed, False)
            cache[site] = m.cpu().contiguous()
        return cache[site]

    relu_masks = [(m.permute(0, 3, 1, 2) if m.dim() == 4 else m).cpu().contiguous() for m in relu_trace]      # NHWC -> NCHW
    return drop_mask, relu_masks


class CheckingPlan(R.DropPlan):
    """The plain oracle (its own ReLUs, injected dropout masks) that also compares its ReLU masks with the HIP run's."""

    def __init__(self, drop_mask, relu_masks):
        super().__init__(drop_mask)
        self.masks, self.mismatch, self.elements, self.worst = relu_masks, 0, 0, 0.0

    def relu(self, x):
        m = self.masks[self.relu_sites]
        assert tuple(m.shape) == tuple(x.shape), (self.relu_sites, tuple(m.shape), tuple(x.shape))
        self.relu_sites += 1
        y = F.relu(x)
        # a fused dropout behind the ReLU zeroes the recorded mask at dropped elements too: compare where the mask is set,
        # and where the oracle's own activation is positive but the mask is not, only count elements that survive no later check
        diff = m & (x.detach() <= 0)
        self.mismatch += int(diff.sum())
        self.elements += m.numel()
        if diff.any():
            self.worst = max(self.worst, float(x.detach()[diff].abs().max()))
        return y


def oracle_grads(fwd, sd32, dtype, drop_mask, relu_masks):
    sd = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd32.items()}
    plan = R.DropPlan(lambda *a: drop_mask(*a).to(dtype), relu_fn=lambda site, shape: relu_masks[site].to(dtype))
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)          # the oracle's sinusoid tables follow the default dtype
    try:
        logits, loss = fwd(sd, plan)
        loss.backward()
    finally:
        torch.set_default_dtype(prev)
    assert plan.relu_sites == len(relu_masks)
    return {k: v.grad for k, v in sd.items()}


def check_against_oracle(model, logits, loss, fwd, sd, trace, relu_trace, rseed):
    drop_mask, relu_masks = materialise(trace, relu_trace)
    # 1. forward + loss against the plain oracle
    random.seed(rseed)
    plain = CheckingPlan(drop_mask, relu_masks)
    with torch.no_grad():
        lo32, loss32 = fwd(sd, plain)
    assert plain.sites == len(trace) and plain.relu_sites == len(relu_masks)
    got = logits.detach().float().cpu()
    assert torch.isfinite(got).all()
    rel = ((got - lo32).norm() / lo32.norm()).item()
    assert rel < 1e-3 and (got - lo32).abs().max().item() < 1e-3 * max(1.0, lo32.abs().max().item()), f"logits rel {rel}"
    assert abs(float(loss) - float(loss32)) / abs(float(loss32)) < 1e-4
    # the HIP run's ReLU masks are the oracle's own except at rounding-level pre-activations
    assert plain.mismatch <= max(3, 2e-5 * plain.elements) and plain.worst < 1e-4, (plain.mismatch, plain.elements, plain.worst)
    # 2. every parameter gradient against the fp64 oracle on the same piecewise-linear region
    random.seed(rseed)
    g64 = oracle_grads(fwd, sd, torch.float64, drop_mask, relu_masks)
    worst = (0.0, "")
    for n, p in model.named_parameters():
        ref = g64[n]
        if ref is None:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        err = ((p.grad.detach().double().cpu() - ref).norm() / ref.norm()).item()
        worst = max(worst, (err, n))
    assert worst[0] < 1e-3, worst
    return worst


def load(module, shapes, seed):
    sd = syn.seeded_state_dict(shapes, seed)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith("pe") or m.endswith("pe_hwc") for m in missing)
    return sd


@pytest.mark.parametrize("rseed", [2, 6, 9])      # together: MixDropout at position 1 / 2 / 3, nn.Dropout and nn.Dropout2d, in ConvBlocks and DSCBlocks
@pytest.mark.parametrize("window", [-1, 5])
def test_unimodal_train_mode_matches_oracle_with_injected_masks(rseed, window):
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout, trace_dropout, trace_relu
    V, H, W, T, L = 50, 64, 160, 12, 2
    w2i, i2w = syn.make_vocab(V)
    cfg = ModelConfig(num_layers=L)               # reference dropout rates: 0.1 decoder / PE, 0.5 | 0.25 MixDropout
    m = Transformer(H, W, 16, w2i, i2w, attn_window=window, config=cfg)
    sd = load(m, syn.transformer_shapes(V, layers=L), 41)
    m.flatten_parameters()
    m.train()
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, H, W, T, V, w2i["<sos>"], w2i["<eos>"], seed=9)
    seed_dropout(100 + rseed)
    random.seed(rseed)
    m.zero_grad()
    with trace_dropout() as trace, trace_relu() as relus:
        logits = m(x.to(DEV), xl.to(DEV), y_in.to(DEV))
        loss = m.compute_loss(logits, y_out.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    kinds = [t[0] for t in trace]
    assert len(trace) == 9 + 1 + 1 + 6 * L and kinds.count("attn") == 2 * L and kinds.count("nhwc") == 10
    assert len(relus) == 5 * 3 + 4 * 2 + L
    ocfg = R.OracleCfg(num_layers=L, attn_window=window)

    def fwd(sdx, plan):
        lo = R.transformer_forward(sdx, x.to(next(iter(sdx.values())).dtype), xl, y_in, ocfg, H, W, drop=plan)
        return lo, R.ce_loss(lo, y_out)

    check_against_oracle(m, logits, loss, fwd, sd, trace, relus, rseed)
    from omr_a2s_multimodal_transformer_amd import kernels as K
    for kind, p, seed, ch in trace:                # the masks drop what they should
        if kind == "attn":
            assert abs(K.attn_dropout_mask(2, 4, T, 80, p, seed, DEV).float().mean().item() - (1 - p)) < 0.02


@pytest.mark.parametrize("mt,modality,rseed", [("attn_both", "both", 6), ("concat", "both", 2), ("attn_img", "both", 9), ("attn_both", "audio", 2)])
def test_multimodal_train_mode_matches_oracle_with_injected_masks(mt, modality, rseed):
    """Both encoders (their MixDropouts draw from the same Python stream, in the reference's order), the CrossAttention
    mixer's probability dropout with the quirk-2 mask, bool / additive key masks in the decoder."""
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout, trace_dropout, trace_relu
    V, L = 40, 2
    IMG, AUD = (64, 160), (195, 96)
    w2i, i2w = syn.make_vocab(V)
    m = MultimodalTransformer(IMG[0], IMG[1], AUD[0], AUD[1], 12, w2i, i2w, mixer_type=mt, config=ModelConfig(num_layers=L))
    sd = load(m, syn.multimodal_shapes(V, mt, layers=L), 51)
    m.flatten_parameters()
    m.train()
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, IMG[0], IMG[1], 11, V, w2i["<sos>"], w2i["<eos>"], seed=16)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, AUD[0], AUD[1], 11, V, w2i["<sos>"], w2i["<eos>"], seed=17, pad_value=0.0)
    m.apply_teacher_forcing_modality = lambda: modality
    seed_dropout(200 + rseed)
    random.seed(rseed)
    m.zero_grad()
    with trace_dropout() as trace, trace_relu() as relus:
        logits = m(xi.to(DEV), xli, xa.to(DEV), xla, y_in, apply_teacher_forcing_modality=True)
        loss = m.compute_loss(logits, y_out.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    n_mixer = {"concat": 0, "attn_img": 1, "attn_audio": 1, "attn_both": 2}[mt] if modality == "both" else 0
    assert len(trace) == 2 * 10 + n_mixer + 1 + 6 * L and len(relus) == 2 * (5 * 3 + 4 * 2) + L
    ocfg = R.OracleCfg(num_layers=L)

    def fwd(sdx, plan):
        dt = next(iter(sdx.values())).dtype
        lo = R.multimodal_forward(sdx, xi.to(dt), xli, xa.to(dt), xla, y_in, ocfg, mt, IMG, AUD, modality, drop=plan)
        return lo, R.ce_loss(lo, y_out)

    check_against_oracle(m, logits, loss, fwd, sd, trace, relus, rseed)


def test_bf16_train_mode_tracks_the_fp32_oracle_under_the_same_masks():
    """bf16 compute path (the benchmark's): same sites, same counter-based masks; loss within 2e-2 of the fp32 oracle fed
    with the masks of the bf16 run (the mask of a site does not depend on the compute dtype)."""
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout, trace_dropout
    V, H, W, T, L = 50, 64, 160, 12, 2
    w2i, i2w = syn.make_vocab(V)
    m = Transformer(H, W, 16, w2i, i2w, config=ModelConfig(num_layers=L, compute_dtype="bf16"))
    sd = load(m, syn.transformer_shapes(V, layers=L), 41)
    m.flatten_parameters()
    m.train()
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, H, W, T, V, w2i["<sos>"], w2i["<eos>"], seed=9)
    seed_dropout(7)
    random.seed(6)
    m.zero_grad()
    with trace_dropout() as trace:
        loss = m.compute_loss(m(x.to(DEV), xl, y_in), y_out.to(DEV))
    loss.backward()
    random.seed(6)
    drop_mask, _ = materialise(trace, [])
    plan = R.DropPlan(drop_mask)
    ref = R.ce_loss(R.transformer_forward(sd, x, xl, y_in, R.OracleCfg(num_layers=L), H, W, drop=plan), y_out)
    assert plan.sites == len(trace)
    assert abs(float(loss) - float(ref)) / float(ref) < 2e-2
    assert torch.isfinite(m._flat.grad).all() and float(m._flat.grad.abs().max()) > 0
