"""Pins oracle/ref_cpu.py against golden vectors produced by the imported reference
(tests/golden/gen_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

from omr_a2s_multimodal_transformer_amd import synthetic as syn
from oracle import ref_cpu as R

TOL = dict(rtol=2e-5, atol=2e-5)


def rnd(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    np.testing.assert_allclose(a.detach().numpy() if isinstance(a, torch.Tensor) else a, b, **tol)


def conv_block_shapes(p, cin, cout):
    s = {}
    for j, c in enumerate((cin, cout, cout), 1):
        s[f"{p}conv{j}.weight"] = (cout, c, 3, 3)
        s[f"{p}conv{j}.bias"] = (cout,)
    return s


def dsc_shapes(p, cin, cout):
    s = {}
    for j, c in enumerate((cin, cout, cout), 1):
        s[f"{p}conv{j}.depth_conv.weight"] = (c, 1, 3, 3)
        s[f"{p}conv{j}.depth_conv.bias"] = (c,)
        s[f"{p}conv{j}.point_conv.weight"] = (cout, c, 1, 1)
        s[f"{p}conv{j}.point_conv.bias"] = (cout,)
    return s


def test_f1_encoder(golden):
    g = golden("f1_encoder")
    sd = syn.seeded_state_dict(conv_block_shapes("cb.", 16, 32), 11)
    close(R.conv_block(sd, "cb.", rnd((2, 16, 13, 19), 101, -1, 1), (2, 2)), g["cb_out"])
    sd = syn.seeded_state_dict(dsc_shapes("db.", 128, 128), 12)
    close(R.dsc_block(sd, "db.", rnd((2, 128, 5, 7), 102, -1, 1)), g["db_out"])
    sd = syn.seeded_state_dict(syn.encoder_shapes("encoder."), 13)
    close(R.encoder(sd, "encoder.", rnd((2, 1, 48, 80), 103)), g["enc_a"], atol=1e-4)
    close(R.encoder(sd, "encoder.", rnd((1, 1, 195, 64), 104)), g["enc_b"], atol=1e-4)


def test_f2_pe(golden):
    g = golden("f2_pe")
    close(R.pe2d_table(256, 4, 6), g["pe2d"], atol=1e-6)
    close(R.pe2d_table(128, 3, 5), g["pe2d_128"], atol=1e-6)
    close(R.pe1d_table(32, 256), g["pe1d"], atol=1e-6)
    close(R.pe1d_table(16, 128), g["pe1d_128"], atol=1e-6)


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("win", [-1, 3, 100])
def test_f3_decoder(golden, L, win):
    g = golden("f3_decoder")
    sd = syn.seeded_state_dict(syn.decoder_shapes("decoder.", 64, layers=L), 20 + L)
    cfg = R.OracleCfg(num_layers=L, attn_window=win)
    tgt = torch.from_numpy(g["tgt"])
    mem = rnd((3, 20, 256), 301, -1, 1)
    close(R.decoder(sd, "decoder.", tgt, mem, torch.from_numpy(g["lens"]), cfg), g[f"L{L}_w{win}_len"], atol=1e-4)
    close(R.decoder(sd, "decoder.", tgt, mem, torch.from_numpy(g["bmask"]), cfg), g[f"L{L}_w{win}_bool"], atol=1e-4)
    close(R.decoder(sd, "decoder.", tgt, mem, None, cfg), g[f"L{L}_w{win}_none"], atol=1e-4)


def test_f3_additive_mask_is_not_minus_inf(golden):
    """Quirk 1: int lengths give a +1.0 additive bias, which differs visibly from -inf masking."""
    g = golden("f3_decoder")
    a, b = g["L2_w-1_len"], g["L2_w-1_bool"]
    assert np.abs(a - b).max() > 1e-2


def test_f4_cross_attention_and_mixers(golden):
    g = golden("f4_cross_attention")
    sd = syn.seeded_state_dict(syn.mha_shapes("cross_attn.attention.", 256), 31)
    q, kv = rnd((3, 9, 256), 401, -1, 1), rnd((3, 11, 256), 402, -1, 1)
    lq, lkv = torch.from_numpy(g["lq"]), torch.from_numpy(g["lkv"])
    close(R.cross_attention(sd, "cross_attn.", q, lq, kv, lkv), g["ca_masked"], atol=1e-5)
    close(R.cross_attention(sd, "cross_attn.", q, None, kv, None), g["ca_nomask"], atol=1e-5)
    xi, xa = rnd((3, 11, 256), 403, -1, 1), rnd((3, 9, 256), 404, -1, 1)
    xli, xla = torch.from_numpy(g["xli"]), torch.from_numpy(g["xla"])
    for mt in ("concat", "attn_img", "attn_audio", "attn_both"):
        x, xl = R.mixer(sd, mt, xi, xa, xli, xla)
        close(x, g[f"mix_{mt}_x"], atol=1e-5)
        np.testing.assert_array_equal(xl.numpy(), g[f"mix_{mt}_xl"])
        x2, xl2 = R.mixer(sd, mt, xi, xa, None, None)
        close(x2, g[f"mix_{mt}_x_nolen"], atol=1e-5)
        assert xl2 is None


def grad_norms(sd, names):
    return np.array([float(sd[n].grad.double().norm()) if sd[n].grad is not None else -1.0 for n in names])


def test_f5_transformer_forward_backward(golden):
    g = golden("f5_transformer")
    V = 50
    sd = syn.seeded_state_dict(syn.transformer_shapes(V), 41)
    for v in sd.values():
        v.requires_grad_(True)
    w2i, _ = syn.make_vocab(V)
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5)
    np.testing.assert_array_equal(y_in.numpy(), g["y_in"])
    logits = R.transformer_forward(sd, x, xl, y_in, R.OracleCfg(), 32, 64)
    close(logits, g["logits"], atol=1e-4)
    loss = R.ce_loss(logits, y_out)
    close(loss, g["loss"], atol=1e-5)
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    assert names == list(sd.keys())  # registration-order contract
    np.testing.assert_allclose(grad_norms(sd, names), g["grad_norms"], rtol=2e-3, atol=1e-6)


def test_f6_adam(golden):
    g = golden("f5_transformer")
    V = 50
    sd = syn.seeded_state_dict(syn.transformer_shapes(V), 41)
    for v in sd.values():
        v.requires_grad_(True)
    w2i, _ = syn.make_vocab(V)
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5)
    ps = list(sd.values())
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    losses = []
    for step in range(1, 4):
        for p in ps:
            p.grad = None
        loss = R.ce_loss(R.transformer_forward(sd, x, xl, y_in, R.OracleCfg(), 32, 64), y_out)
        loss.backward()
        losses.append(float(loss.detach()))
        with torch.no_grad():
            R.adam_step(ps, [p.grad for p in ps], m, v, step)
    np.testing.assert_allclose(losses, g["adam_losses"], rtol=1e-4)
    for k, s, h in zip(g["adam_sel"], g["adam_sums"], g["adam_heads"]):
        t = sd[str(k)].detach()
        np.testing.assert_allclose(float(t.double().sum()), s, rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(t.flatten()[:8].numpy(), h, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("mt,modality", [("concat", "both"), ("attn_img", "both"), ("attn_audio", "both"),
                                         ("attn_both", "both"), ("attn_both", "image"), ("attn_both", "audio")])
def test_f5_multimodal(golden, mt, modality):
    g = golden("f5_multimodal")
    V = 40
    sd = syn.seeded_state_dict(syn.multimodal_shapes(V, mt), 51)
    for v in sd.values():
        v.requires_grad_(True)
    w2i, _ = syn.make_vocab(V)
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, 32, 48, 9, V, w2i["<sos>"], w2i["<eos>"], seed=6)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, 35, 40, 9, V, w2i["<sos>"], w2i["<eos>"], seed=7, pad_value=0.0)
    logits = R.multimodal_forward(sd, xi, xli, xa, xla, y_in, R.OracleCfg(), mt, (32, 48), (35, 40), modality)
    close(logits, g[f"{mt}_{modality}_logits"], atol=1e-4)
    loss = R.ce_loss(logits, y_out)
    close(loss, g[f"{mt}_{modality}_loss"], atol=1e-5)
    loss.backward()
    names = [str(n) for n in g[f"{mt}_{modality}_grad_names"]]
    assert names == list(sd.keys())
    np.testing.assert_allclose(grad_norms(sd, names), g[f"{mt}_{modality}_grad_norms"], rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("win", [-1, 4])
def test_f7_greedy_decode(golden, win):
    g = golden("f7_decode")
    V = 30
    w2i, _ = syn.make_vocab(V)
    sd = syn.seeded_state_dict(syn.transformer_shapes(V), 61)
    cfg = R.OracleCfg(attn_window=win)
    with torch.no_grad():
        pe = R.pe2d_table(256, 2, 12)
        mem = R.encode_to_memory(sd, "encoder.", pe, rnd((1, 1, 32, 96), 701))
        toks, tops = R.greedy_decode(sd, "decoder.", mem, w2i["<sos>"], w2i["<eos>"], 14, cfg, return_logits=True)
    np.testing.assert_array_equal(np.array(toks), g[f"w{win}_tokens"])
    np.testing.assert_allclose(np.array([t[0] for t in tops]), g[f"w{win}_top1"], rtol=1e-4, atol=1e-4)


def test_f8_metrics(golden):
    g = golden("f8_metrics")
    cases = [([["a", "b", "c"]], [["a", "c"]]),
             ([["a", "b"], ["c", "d", "e"]], [["a", "b"], ["c", "x", "e", "f"]]),
             ([["x"] * 5, ["y"]], [[], ["y"]])]
    for i, (t, p) in enumerate(cases):
        m = R.compute_ed_metrics(t, p)
        assert math.isclose(m["sym-er"], g["sym"][i]) and math.isclose(m["seq-er"], g["seq"][i])


def test_f9_c1_variant(golden):
    g = golden("f9_c1")
    V = 45
    shapes = syn.encoder_shapes("encoder.", 1, 128)
    shapes.update(syn.decoder_shapes("decoder.", V, 128, 128, 2))
    sd = syn.seeded_state_dict(shapes, 71)
    for v in sd.values():
        v.requires_grad_(True)
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, 44, 43, seed=8)
    cfg = R.OracleCfg(d_model=128, ff_dim=128, num_layers=2)
    logits = R.transformer_forward(sd, x, xl, y_in, cfg, 32, 64)
    close(logits, g["logits"], atol=1e-4)
    loss = R.ce_loss(logits, y_out)
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    assert names == list(sd.keys())
    np.testing.assert_allclose(grad_norms(sd, names), g["grad_norms"], rtol=2e-3, atol=1e-6)


def test_f10_collate(golden):
    g = golden("f10_collate")
    gen = torch.Generator().manual_seed(3)
    items = []
    for h, w, n in ((10, 17, 5), (12, 9, 7), (7, 20, 3)):
        items.append((torch.rand((1, h, w), generator=gen), w, torch.randint(1, 9, (n,), generator=gen)))
    xi, xli, yi, yo = R.collate_unimodal(items, 1.0)
    xa, xla, _, _ = R.collate_unimodal(items, 0.0)
    np.testing.assert_array_equal(xi.numpy(), g["xi"])
    np.testing.assert_array_equal(xa.numpy(), g["xa"])
    np.testing.assert_array_equal(xli.numpy(), g["xli"])
    np.testing.assert_array_equal(yi.numpy(), g["yi"])
    np.testing.assert_array_equal(yo.numpy(), g["yo"])
    assert xli.dtype == torch.int32 and yi.dtype == torch.int64 and xi.dtype == torch.float32
