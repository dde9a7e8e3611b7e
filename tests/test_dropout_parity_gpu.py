"""TRAIN-MODE (dropout ON) parity of the HIP path against the CPU oracle, by mask injection.

The HIP path never stores a dropout mask: every site's mask is a pure function of (seed, element index) that the forward
kernel applies and the backward kernels regenerate (fused conv epilogues, add+LayerNorm, FFN GEMM epilogue, attention
probabilities, stand-alone omr_dropout).  runtime.trace_dropout() records the (kind, p, seed, mode) of every site of a
forward pass; each mask is then MATERIALISED through an independent entry point (omr_dropout on a tensor of ones,
omr_attn_dropout_mask) and injected into oracle.ref_cpu.DropPlan, whose train-mode semantics are pinned against the
reference itself (tests/golden/f12_dropout.npz, tests/test_oracle_golden_r2.py).  Forward, loss and every parameter
gradient are compared, fp32.

Gradient criterion.  fp32 gradients of this network are only reproducible to ~1e-3..1e-2 between ANY two correct fp32
implementations: one ReLU pre-activation within rounding noise of zero flips its mask and moves that layer's gradient (and
everything upstream) by ~1/sqrt(N) of its norm -- 6e-3 for one element of a 3x128x4x20 map (tools/relu_flip_probe.py: the
reference and the oracle, both torch CPU, differ by 2.5e-3 on such tensors; each is ~1.2e-3 from the fp64 result).  So the
arbiter is the oracle run in fp64 with the same masks, and the yardstick is the CPU fp32 arithmetic itself: the fp32 oracle
is run on the exact input AND on inputs jittered by one part in 5e6 (a few ulp: other near-zero elements flip), and the
distance of those runs from the fp64 gradient measures how reproducible this configuration is.  The HIP gradients must be
as close to the fp64 gradient as that: RMS over tensors within 3x of the worst CPU run (floor 1e-3), worst tensor within
3x of the worst CPU tensor (floor 1e-2).  A wrong mask, a missing 1/(1-p) or a mask/element misalignment in any backward
kernel is an O(1) error in the tensors downstream of it.
"""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import synthetic as syn  # noqa: E402
from omr_a2s_multimodal_transformer_amd.config import ModelConfig  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda:0"


def plan_from_trace(trace, dtype=torch.float32):
    """DropPlan whose masks are the HIP path's own, materialised site by site from the recorded (kind, p, seed, mode)."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    cache = {}

    def fn(site, kind, p, shape, channel):
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
        return cache[site].to(dtype)

    return R.DropPlan(fn)


def oracle_run(fwd, sd32, dtype, trace):
    """fwd(sd, plan) -> (logits, loss); returns logits, loss, {name: grad} in `dtype` arithmetic."""
    sd = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd32.items()}
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)          # the oracle's sinusoid tables follow the default dtype
    try:
        logits, loss = fwd(sd, plan_from_trace(trace, dtype))
        loss.backward()
    finally:
        torch.set_default_dtype(prev)
    return logits.detach(), float(loss), {k: v.grad for k, v in sd.items()}


def check_against_oracle(model, logits, loss, fwd, sd, trace, rseed):
    random.seed(rseed)
    lo32, loss32, g32 = oracle_run(fwd, sd, torch.float32, trace)
    random.seed(rseed)
    _, _, g64 = oracle_run(fwd, sd, torch.float64, trace)
    jittered = []
    for k in range(2):                                 # the CPU fp32 arithmetic on inputs a few ulp away: its own flip statistics
        random.seed(rseed)
        jittered.append(oracle_run(lambda sdx, plan: fwd(sdx, plan, jitter=k + 1), sd, torch.float32, trace)[2])
    got = logits.detach().float().cpu()
    assert torch.isfinite(got).all()
    rel = ((got - lo32).norm() / lo32.norm()).item()
    assert rel < 1e-3 and (got - lo32).abs().max().item() < 1e-3 * max(1.0, lo32.abs().max().item()), f"logits rel {rel}"
    assert abs(float(loss) - loss32) / abs(loss32) < 1e-4
    e_hip, e_cpu, names = [], [], []
    for n, p in model.named_parameters():
        ref = g64[n]
        if ref is None:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        e_hip.append(((p.grad.detach().double().cpu() - ref).norm() / ref.norm()).item())
        e_cpu.append([((g[n].double() - ref).norm() / ref.norm()).item() for g in [g32] + jittered])
        names.append(n)
    e_hip, e_cpu = np.array(e_hip), np.array(e_cpu)          # [tensors], [tensors, 3 CPU runs]
    worst = int(np.argmax(e_hip))
    assert e_hip.max() <= max(1e-2, 3.0 * e_cpu.max()), (names[worst], e_hip[worst], e_cpu[worst], e_cpu.max())
    rms_hip, rms_cpu = float(np.sqrt((e_hip ** 2).mean())), float(np.sqrt((e_cpu ** 2).mean(axis=0)).max())
    assert rms_hip <= max(1e-3, 3.0 * rms_cpu), (rms_hip, rms_cpu)
    return rms_hip, rms_cpu


def jittered(x, k):
    """x * (1 + 2e-7 * noise): a few ulp, seeded by k (k = 0: x itself)."""
    if not k:
        return x
    g = torch.Generator().manual_seed(1000 + k)
    return x * (1.0 + 2e-7 * torch.randn(x.shape, generator=g, dtype=torch.float32).to(x.dtype))


def load(module, shapes, seed):
    sd = syn.seeded_state_dict(shapes, seed)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith("pe") or m.endswith("pe_hwc") for m in missing)
    return sd


@pytest.mark.parametrize("rseed", [2, 6, 9])      # together: MixDropout at position 1 / 2 / 3, nn.Dropout and nn.Dropout2d, in ConvBlocks and DSCBlocks
@pytest.mark.parametrize("window", [-1, 5])
def test_unimodal_train_mode_matches_oracle_with_injected_masks(rseed, window):
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout, trace_dropout
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
    with trace_dropout() as trace:
        logits = m(x.to(DEV), xl.to(DEV), y_in.to(DEV))
        loss = m.compute_loss(logits, y_out.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    kinds = [t[0] for t in trace]
    assert len(trace) == 9 + 1 + 1 + 6 * L and kinds.count("attn") == 2 * L and kinds.count("nhwc") == 10
    ocfg = R.OracleCfg(num_layers=L, attn_window=window)

    def fwd(sdx, plan, jitter=0):
        lo = R.transformer_forward(sdx, jittered(x, jitter).to(next(iter(sdx.values())).dtype), xl, y_in, ocfg, H, W, drop=plan)
        assert plan.sites == len(trace)
        return lo, R.ce_loss(lo, y_out)

    check_against_oracle(m, logits, loss, fwd, sd, trace, rseed)
    # the masks drop what they should: keep rates of the materialised masks
    from omr_a2s_multimodal_transformer_amd import kernels as K
    for kind, p, seed, ch in trace:
        if kind == "attn":
            keep = K.attn_dropout_mask(2, 4, T, 80, p, seed, DEV).float().mean().item()
            assert abs(keep - (1 - p)) < 0.02


@pytest.mark.parametrize("mt,modality,rseed", [("attn_both", "both", 6), ("concat", "both", 2), ("attn_img", "both", 9), ("attn_both", "audio", 2)])
def test_multimodal_train_mode_matches_oracle_with_injected_masks(mt, modality, rseed):
    """Both encoders (their MixDropouts draw from the same Python stream, in the reference's order), the CrossAttention
    mixer's probability dropout with the quirk-2 mask, bool / additive key masks in the decoder."""
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout, trace_dropout
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
    with trace_dropout() as trace:
        logits = m(xi.to(DEV), xli, xa.to(DEV), xla, y_in, apply_teacher_forcing_modality=True)
        loss = m.compute_loss(logits, y_out.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    n_mixer = {"concat": 0, "attn_img": 1, "attn_audio": 1, "attn_both": 2}[mt] if modality == "both" else 0
    assert len(trace) == 2 * 10 + n_mixer + 1 + 6 * L
    ocfg = R.OracleCfg(num_layers=L)

    def fwd(sdx, plan, jitter=0):
        dt = next(iter(sdx.values())).dtype
        lo = R.multimodal_forward(sdx, jittered(xi, jitter).to(dt), xli, jittered(xa, 10 * jitter).to(dt), xla, y_in, ocfg, mt, IMG, AUD, modality, drop=plan)
        assert plan.sites == len(trace)
        return lo, R.ce_loss(lo, y_out)

    check_against_oracle(m, logits, loss, fwd, sd, trace, rseed)


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
    plan = plan_from_trace(trace)
    ref = R.ce_loss(R.transformer_forward(sd, x, xl, y_in, R.OracleCfg(num_layers=L), H, W, drop=plan), y_out)
    assert plan.sites == len(trace)
    assert abs(float(loss) - float(ref)) / float(ref) < 2e-2
    assert torch.isfinite(m._flat.grad).all() and float(m._flat.grad.abs().max()) > 0
