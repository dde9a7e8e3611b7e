"""Round-2 GPU parity: the HIP modules against the reference's round-2 golden vectors (tests/golden/gen_golden_r2.py):
non-degenerate shapes (deepest feature maps 4x20 / 13x12), weighted late-fusion decode, Adam with skipped sub-modules."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import synthetic as syn  # noqa: E402
from omr_a2s_multimodal_transformer_amd.config import ModelConfig  # noqa: E402

DEV = "cuda:0"
NO_DROP = dict(dropout=0.0, encoder_dropout=0.0)
IMG_HW, AUD_HW = (64, 160), (195, 96)


def rnd(shape, seed):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed))


def load(module, shapes, seed):
    sd = syn.seeded_state_dict(shapes, seed)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith("pe") or m.endswith("pe_hwc") for m in missing)
    return sd


def check_grad_heads(model, names, ref_norms, ref_heads):
    """The first 8 elements of every gradient tensor (logical layout) against the reference's own (F13 `*_grad_heads`), with the
    criterion the oracle is held to (tests/test_oracle_golden_r2.py check_grads): single elements move more than norms."""
    ps = dict(model.named_parameters())
    heads = np.stack([np.pad(ps[n].grad.detach().float().flatten()[:8].cpu().numpy(), (0, max(0, 8 - ps[n].numel()))) for n in names])
    ref_norms = np.where(np.asarray(ref_norms) < 0, 0.0, ref_norms)
    rms = np.array([ref_norms[i] / np.sqrt(ps[n].numel()) for i, n in enumerate(names)])[:, None]
    bad = np.abs(heads - ref_heads) > 5e-2 * np.abs(ref_heads) + 0.1 * rms + 1e-12
    assert not bad.any(), [names[i] for i in np.flatnonzero(bad.any(axis=1))]


def check_grad_norms(model, names, ref):
    """Norms against the reference's: hard 1e-2 on every tensor, 1e-3 on the typical one (single ReLU-mask flips between two
    fp32 implementations move whole tensors by ~1/sqrt(N): tests/test_dropout_parity_gpu.py has the measurement)."""
    ps = dict(model.named_parameters())
    assert names == [n for n, _ in model.named_parameters()]
    got = np.array([float(ps[n].grad.detach().double().norm()) for n in names])
    ref = np.where(ref < 0, 0.0, ref)         # reference: unused parameters have grad None; here their slice stays zero
    assert np.array_equal(got == 0, ref == 0)
    live = ref > 0
    rel = np.abs(got[live] - ref[live]) / ref[live]
    assert rel.max() < 1e-2 and np.median(rel) < 1e-3, (rel.max(), np.median(rel), names[int(np.argmax(np.where(live, np.abs(got - ref) / np.maximum(ref, 1e-30), 0)))])


def test_unimodal_forward_backward_matches_reference_golden_nondegenerate(golden):
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    g = golden("f13_nondegenerate")
    V = 50
    w2i, i2w = syn.make_vocab(V)
    m = Transformer(IMG_HW[0], IMG_HW[1], 16, w2i, i2w, config=ModelConfig(**NO_DROP))
    load(m, syn.transformer_shapes(V), 41)
    m.flatten_parameters()
    m.train()
    random.seed(0)
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, IMG_HW[0], IMG_HW[1], 12, V, w2i["<sos>"], w2i["<eos>"], seed=9)
    m.zero_grad()
    logits = m(x.to(DEV), xl, y_in)
    np.testing.assert_allclose(logits.detach().float().cpu().numpy(), g["uni_logits"], rtol=1e-3, atol=2e-4)
    loss = m.compute_loss(logits, y_out.to(DEV))
    np.testing.assert_allclose(float(loss), float(g["uni_loss"]), rtol=1e-4)
    loss.backward()
    check_grad_norms(m, [str(n) for n in g["uni_grad_names"]], g["uni_grad_norms"])
    check_grad_heads(m, [str(n) for n in g["uni_grad_names"]], g["uni_grad_norms"], g["uni_grad_heads"])


@pytest.mark.parametrize("mt,modality", [("concat", "both"), ("attn_img", "both"), ("attn_audio", "both"), ("attn_both", "both"),
                                         ("attn_both", "image"), ("attn_both", "audio")])
def test_multimodal_matches_reference_golden_nondegenerate(golden, mt, modality):
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer
    g = golden("f13_nondegenerate")
    V = 40
    w2i, i2w = syn.make_vocab(V)
    m = MultimodalTransformer(IMG_HW[0], IMG_HW[1], AUD_HW[0], AUD_HW[1], 12, w2i, i2w, mixer_type=mt, config=ModelConfig(**NO_DROP))
    load(m, syn.multimodal_shapes(V, mt), 51)
    m.flatten_parameters()
    m.train()
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, IMG_HW[0], IMG_HW[1], 11, V, w2i["<sos>"], w2i["<eos>"], seed=16)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, AUD_HW[0], AUD_HW[1], 11, V, w2i["<sos>"], w2i["<eos>"], seed=17, pad_value=0.0)
    m.apply_teacher_forcing_modality = lambda: modality
    m.zero_grad()
    logits = m(xi.to(DEV), xli, xa.to(DEV), xla, y_in, apply_teacher_forcing_modality=True)
    np.testing.assert_allclose(logits.detach().float().cpu().numpy(), g[f"{mt}_{modality}_logits"], rtol=1e-3, atol=2e-4)
    loss = m.compute_loss(logits, y_out.to(DEV))
    np.testing.assert_allclose(float(loss), float(g[f"{mt}_{modality}_loss"]), rtol=1e-4)
    loss.backward()
    check_grad_norms(m, [str(n) for n in g[f"{mt}_{modality}_grad_names"]], g[f"{mt}_{modality}_grad_norms"])
    check_grad_heads(m, [str(n) for n in g[f"{mt}_{modality}_grad_names"]], g[f"{mt}_{modality}_grad_norms"], g[f"{mt}_{modality}_grad_heads"])
    assert m._touched == {"both": None, "image": ("image_encoder", "decoder"), "audio": ("audio_encoder", "decoder")}[modality]


@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.5, 1.0])
def test_weighted_late_fusion_decode_matches_reference_tokens(golden, alpha):
    """weighted_prediction (src/multimodal/weighted_multimodal/test.py:21-70): bit-exact token ids for every alpha."""
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    from omr_a2s_multimodal_transformer_amd.weighted_fusion import weighted_prediction
    g = golden("f15_weighted")
    V = 30
    w2i, i2w = syn.make_vocab(V)
    img = Transformer(64, 128, 14, w2i, i2w).eval()
    load(img, syn.transformer_shapes(V), 81)
    img.flatten_parameters()
    aud = Transformer(195, 64, 14, w2i, i2w).eval()
    load(aud, syn.transformer_shapes(V), 82)
    aud.flatten_parameters()
    xi, xa = rnd((1, 1, 64, 128), 801).to(DEV), rnd((1, 1, 195, 64), 802).to(DEV)
    words = weighted_prediction(xi, xa, img, aud, alpha=alpha)
    np.testing.assert_array_equal(np.array([w2i[w] for w in words]), g[f"a{alpha}_tokens"])
    assert float(np.min(g[f"a{alpha}_margin"])) > 1e-6
    if alpha == 1.0:                        # alpha = 1: the image model's own greedy decode
        assert words == img._greedy(img.encode(xi))[0]
    if alpha == 0.0:
        assert words == aud._greedy(aud.encode(xa))[0]


def test_weighted_argmax_kernel_matches_torch():
    from omr_a2s_multimodal_transformer_amd import kernels as K
    for n, alpha in ((6997, 0.3), (30, 0.5), (257, 1.0), (1000, 0.0)):
        la, lb = (rnd((n,), 5) * 8 - 4).to(DEV), (rnd((n,), 6) * 8 - 4).to(DEV)
        idx, prob = K.weighted_argmax(la, lb, alpha)
        mix = alpha * la.cpu().softmax(-1) + (1 - alpha) * lb.cpu().softmax(-1)
        assert int(idx) == int(mix.argmax()), (n, alpha)
        torch.testing.assert_close(prob.cpu()[0], mix.max(), rtol=1e-5, atol=1e-8)
    la = torch.zeros(64, device=DEV)
    assert int(K.weighted_argmax(la, la.clone(), 0.5)[0]) == 0          # ties: first index


def test_adam_skips_submodules_without_gradient_like_the_reference(golden):
    """Five Adam steps of the multimodal model with modality drops (both, image, image, audio, both): parameters of the
    sub-modules that took no part in a step keep their values, moments and step counts (torch.optim.Adam with grad None)."""
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer
    g = golden("f17_adam_multimodal")
    V = 40
    w2i, i2w = syn.make_vocab(V)
    m = MultimodalTransformer(32, 48, 35, 40, 12, w2i, i2w, mixer_type="attn_both", config=ModelConfig(**NO_DROP))
    sd0 = load(m, syn.multimodal_shapes(V, "attn_both"), 51)
    m.flatten_parameters()
    m.train()
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, 32, 48, 9, V, w2i["<sos>"], w2i["<eos>"], seed=6)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, 35, 40, 9, V, w2i["<sos>"], w2i["<eos>"], seed=7, pad_value=0.0)
    xi, xa, y_out = xi.to(DEV), xa.to(DEV), y_out.to(DEV)
    opt = m.configure_optimizers()
    losses = []
    frozen = None
    for modality in [str(s) for s in g["seq"]]:
        m.apply_teacher_forcing_modality = lambda mod=modality: mod
        opt.zero_grad()
        before = m._flat.master.clone()
        loss = m.compute_loss(m(xi, xli, xa, xla, y_in, apply_teacher_forcing_modality=True), y_out)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        rng = opt.ranges
        if modality == "image":             # the audio encoder and the mixer did not move at all
            for name in ("audio_encoder", "cross_attn"):
                b, e = rng[name]
                assert torch.equal(before[b:e], m._flat.master[b:e]), name
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3)
    assert opt.steps == {"image_encoder": 4, "audio_encoder": 3, "decoder": 5, "cross_attn": 2}
    ps = dict(m.named_parameters())
    for k, da, h in zip(g["sel"], g["delta_abs"], g["heads"]):
        d = (ps[str(k)].detach().cpu() - sd0[str(k)]).double()
        np.testing.assert_allclose(float(d.abs().sum()), da, rtol=2e-2, err_msg=str(k))       # Adam's sign-like update amplifies tiny gradient differences
        np.testing.assert_allclose(d.flatten()[:8].numpy(), h, rtol=0.2, atol=2e-5, err_msg=str(k))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_linear_weight_gradients_match_torch(dtype):
    """omr_linear_wgrad_grouped: many (dY, X) pairs of different shapes in one launch, accumulated into fp32 buffers that
    already hold values; ragged sizes, a strided dY (logit rows with a padded pitch), a row-group view of dW / db, and more
    problems than one kernel-argument table holds."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 256, 256), (77, 768, 256), (4096, 128, 128), (513, 50, 256), (2048, 256, 128)] * 6      # 30 problems (> 24)
    probs, refs = [], []
    for i, (rows, n_out, n_in) in enumerate(shapes):
        ld = (n_out + 7) // 8 * 8 + (8 if i % 2 else 0)
        dybuf = (torch.randn((rows, ld), generator=g) * 0.5).to(dtype)
        x = torch.randn((rows, n_in), generator=g).to(dtype)
        dw0, db0 = torch.randn((n_out, n_in), generator=g), torch.randn(n_out, generator=g)
        dy = dybuf[:, :n_out]
        refs.append((dw0.double() + dy.double().t() @ x.double(), db0.double() + dy.double().sum(0)))
        probs.append((dybuf.to(DEV)[:, :n_out], x.to(DEV), dw0.to(DEV), db0.to(DEV) if i % 3 else None, None))
    # row-group view: two groups of 128 rows inside a [2 * 384, 64] block (rows [128, 256) of each 384-row group)
    dy, x = torch.randn((640, 256), generator=g).to(dtype), torch.randn((640, 64), generator=g).to(dtype)
    blk, bb = torch.randn((768, 64), generator=g), torch.randn(768, generator=g)
    want_w, want_b = blk.double().clone(), bb.double().clone()
    full_w, full_b = dy.double().t() @ x.double(), dy.double().sum(0)
    for grp in range(2):
        want_w[grp * 384 + 128: grp * 384 + 256] += full_w[grp * 128:(grp + 1) * 128]
        want_b[grp * 384 + 128: grp * 384 + 256] += full_b[grp * 128:(grp + 1) * 128]
    probs.append((dy.to(DEV), x.to(DEV), blk.to(DEV), bb.to(DEV), (128, 384, 128)))
    K.linear_wgrad_grouped(probs)
    tol = 1e-4 if dtype == torch.float32 else 2e-2        # bf16 inputs were rounded before the fp64 reference was formed: only accumulation order differs
    tol = 1e-4 if dtype == torch.float32 else 1e-3
    for (dy_, x_, dw, db, _), (rw, rb) in zip(probs[:-1], refs):
        assert ((dw.double().cpu() - rw).norm() / rw.norm()).item() < tol
        if db is not None:
            assert ((db.double().cpu() - rb).norm() / rb.norm()).item() < tol
    assert ((probs[-1][2].double().cpu() - want_w).norm() / want_w.norm()).item() < tol
    assert ((probs[-1][3].double().cpu() - want_b).norm() / want_b.norm()).item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fp8_row_quantisation_and_fp8_mfma_gemm(dtype):
    """fp8 path (BASELINE config 5; the reference has no fp8, so the checker is torch's own float8_e4m3fn): the row
    quantiser reproduces scale = absmax / 448 and torch's round-to-nearest e4m3 codes; the fp8-MFMA GEMM equals the fp64
    product of the DE-QUANTISED operands (only the accumulation order differs) and is within fp8 rounding of the unquantised
    product."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    for M, N, Kd in ((5, 256, 256), (130, 6997, 256), (64, 768, 128), (1, 256, 256)):
        x = (torch.randn((M, Kd), generator=g) * 1.7).to(dtype)
        w = (torch.randn((N, Kd), generator=g) * 0.06).to(dtype)
        bias = torch.randn(N, generator=g) * 0.1
        x8, sx = K.quantize_rows_fp8(x.to(DEV))
        w8, sw = K.quantize_rows_fp8(w.to(DEV))
        for t, q, s in ((x, x8, sx), (w, w8, sw)):
            ref_s = t.float().abs().amax(dim=1) / 448.0
            torch.testing.assert_close(s.cpu(), ref_s, rtol=1e-6, atol=0)
            ref_q = (t.float() / ref_s[:, None]).to(torch.float8_e4m3fn)
            same = (q.cpu().view(torch.float8_e4m3fn).float() == ref_q.float()).float().mean().item()
            assert same > 0.999, same                       # ties of the division's last bit may round the other way
        deq = lambda q, s: q.cpu().view(torch.float8_e4m3fn).double() * s.cpu().double()[:, None]
        ref = deq(x8, sx) @ deq(w8, sw).t() + bias.double()
        for relu in (False, True):
            out = K.gemm_fp8(x8, sx, w8, sw, bias=bias.to(DEV), relu=relu, out_dtype=torch.float32)
            want = ref.clamp_min(0) if relu else ref
            assert ((out.double().cpu() - want).norm() / want.norm()).item() < 1e-4      # fp32 accumulation order + fp32 scale products
        out16 = K.gemm_fp8(x8, sx, w8, sw, bias=bias.to(DEV), out_dtype=torch.bfloat16)
        assert ((out16.double().cpu() - ref).norm() / ref.norm()).item() < 5e-3
        full = x.double() @ w.double().t() + bias.double()
        assert ((ref - full).norm() / full.norm()).item() < 6e-2              # what e4m3 costs (3 mantissa bits on both operands)


def test_fp8_decode_tracks_the_bf16_decode():
    """Greedy decoding with fp8 MFMA weights (ModelConfig.fp8_decode): the per-step logits stay close to the bf16 decode fed
    with the same tokens, and a run of tokens from one host call equals the single steps of the fp8 mode."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    V = 40
    w2i, i2w = syn.make_vocab(V)
    models = {}
    for fp8 in (False, True):
        m = Transformer(32, 96, 20, w2i, i2w, config=ModelConfig(num_layers=3, compute_dtype="bf16", fp8_decode=fp8)).eval()
        load(m, syn.transformer_shapes(V, layers=3), 61)
        m.flatten_parameters()
        models[fp8] = m
    x = rnd((2, 1, 32, 96), 702).to(DEV)
    mem = models[False].encode(x)
    st16, st8 = models[False].decoder.init_decode(mem), models[True].decoder.init_decode(mem)
    assert st8.fp8 and not st16.fp8
    tok = torch.full((2, 1), w2i["<sos>"], dtype=torch.int64, device=DEV)
    for _ in range(6):
        l16 = models[False].decoder.decode_step(tok, st16).clone()
        l8 = models[True].decoder.decode_step(tok, st8).clone()
        assert torch.isfinite(l8).all()
        assert ((l8 - l16).norm() / l16.norm()).item() < 0.15
        tok = K.argmax(l16.contiguous())[0].view(2, 1)
    st_a, st_b = models[True].decoder.init_decode(mem), models[True].decoder.init_decode(mem)
    tok0 = torch.full((2, 1), w2i["<sos>"], dtype=torch.int64, device=DEV)
    toks, _ = models[True].decoder.decode_tokens(tok0, st_a, 5)
    t = tok0
    for i in range(5):
        idx, _ = K.argmax(models[True].decoder.decode_step(t, st_b).contiguous())
        assert torch.equal(idx, toks[i])
        t = idx.view(2, 1)
    words, _ = models[True]._greedy(mem[:1].contiguous())
    assert len(words) >= 1


@pytest.mark.parametrize("dtype,win,B", [("fp32", -1, 1), ("fp32", 3, 3), ("bf16", -1, 2)])
def test_native_decode_runs_of_tokens_equal_single_steps(dtype, win, B):
    """omr_decode_steps: n positions from ONE host call (token chained on the device) give exactly the tokens, top-1 logits
    and cache contents of n single-position calls with the token picked by omr_argmax in between."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    V = 30
    w2i, i2w = syn.make_vocab(V)
    m = Transformer(32, 96, 20, w2i, i2w, attn_window=win, config=ModelConfig(num_layers=3, compute_dtype=dtype)).eval()
    load(m, syn.transformer_shapes(V, layers=3), 61)
    m.flatten_parameters()
    mem = m.encode(rnd((B, 1, 32, 96), 702).to(DEV))
    tok0 = torch.full((B, 1), w2i["<sos>"], dtype=torch.int64, device=DEV)
    st_run = m.decoder.init_decode(mem)
    toks, top1 = m.decoder.decode_tokens(tok0, st_run, 9)
    toks2, top2 = m.decoder.decode_tokens(toks[-1].view(B, 1), st_run, 4)           # a second run continues the same cache
    st_one = m.decoder.init_decode(mem)
    tok = tok0
    for i in range(13):
        logits = m.decoder.decode_step(tok, st_one)
        idx, val = K.argmax(logits.view(B, -1).contiguous())
        want_t, want_v = (toks[i], top1[i]) if i < 9 else (toks2[i - 9], top2[i - 9])
        assert torch.equal(idx, want_t) and torch.equal(val, want_v), i
        tok = idx.view(B, 1)
    assert st_run.t == st_one.t == 13
    assert torch.equal(st_run.self_kv[:, :, :13], st_one.self_kv[:, :, :13])
    with pytest.raises(RuntimeError):
        m.decoder.decode_tokens(tok, st_run, 8)                                      # 13 + 8 > max_seq_len 20: positional table exhausted


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 5, 19])
def test_decode_row_linear_prologues_equal_the_separate_kernels(dtype, M):
    """omr_decode_linear: each prologue builds exactly the rows the stand-alone kernel would have written (add + LayerNorm,
    embedding + positional row, key-split merge), the linear itself matches the MFMA GEMM on those rows to fp32 round-off, and
    a row's result does not depend on how many rows share the launch."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(31 + M)
    d, ff, V, H = 128, 384, 77, 4
    dev = DEV
    r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev).to(dtype)
    w1, b1 = r(ff, d), torch.randn(ff, generator=g).to(dev)
    w2, b2 = r(3 * d, ff), torch.randn(3 * d, generator=g).to(dev)
    y, res = r(M, d), r(M, d)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(dev), (0.1 * torch.randn(d, generator=g)).to(dev)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    # 1: LayerNorm(y + res) prologue, ReLU epilogue
    xn_ref, _, _ = K.add_layernorm_fwd(y, res, gamma, beta, 1e-5)
    h, _, built, _ = K.decode_linear(w1, b1, x=y, res=res, ln=(gamma, beta, 1e-5), relu=True)
    assert torch.equal(built, xn_ref)
    torch.testing.assert_close(h.float(), torch.relu(xn_ref.float() @ w1.float().t() + b1), **tol)
    # 0: plain rows, two output segments, fp32 copy of the rounded values
    o0, o1, _, o32 = K.decode_linear(w2, b2, x=h, n0=d, want32=True)
    ref = h.float() @ w2.float().t() + b2
    torch.testing.assert_close(torch.cat([o0, o1], 1).float(), ref, **tol)
    assert torch.equal(o32, torch.cat([o0, o1], 1).float())
    # the greedy pick = first index of the row maximum of the rounded outputs
    _, _, _, o32b, (idx, val) = K.decode_linear(w2, b2, x=h, n0=d, want32=True, want_argmax=True)
    assert torch.equal(idx, o32b.argmax(dim=1)) and torch.equal(val, o32b.max(dim=1).values)
    # row independence: the same rows alone
    for i in (0, M - 1):
        a0, a1, _, _ = K.decode_linear(w2, b2, x=h[i:i + 1].contiguous(), n0=d)
        assert torch.equal(a0[0], o0[i]) and torch.equal(a1[0], o1[i])
    # 2: embedding + positional row
    emb, pe = r(V, d), torch.randn(d, generator=g).to(dev)
    tok = torch.randint(0, V, (M,), generator=g).to(dev)
    q, _, built, _ = K.decode_linear(w1, b1, tokens=tok, emb=emb, pe_row=pe)
    ref_rows = (emb[tok].float() + pe).to(dtype)
    assert torch.equal(built, ref_rows)
    torch.testing.assert_close(q.float(), ref_rows.float() @ w1.float().t() + b1, **tol)
    # 3: merge of key-split partials = the attention's own merge kernel
    S, hd = 1100, d // H
    qq, kv = r(M, 1, d), r(M, S, 2 * d)
    o_ref, _ = K.attn_fwd(qq, kv[..., :d], kv[..., d:], H)
    part, ns = K.attn_fwd_split_partials(qq, kv[..., :d], kv[..., d:], H)
    assert ns > 1
    wo, bo = r(d, d), torch.randn(d, generator=g).to(dev)
    o, _, _, _ = K.decode_linear(wo, bo, part=part, heads=H)
    torch.testing.assert_close(o.float(), o_ref.view(M, d).float() @ wo.float().t() + bo, **tol)
