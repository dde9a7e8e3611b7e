"""Model-level parity on the GPU: the HIP modules vs (a) the committed golden vectors produced by the imported
reference and (b) the CPU oracle on the same seeded inputs.  fp32 tolerance: 1e-3 relative (north star)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import synthetic as syn  # noqa: E402
from omr_a2s_multimodal_transformer_amd.config import ModelConfig  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda:0"
NO_DROP = dict(dropout=0.0, encoder_dropout=0.0)


def rnd(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def close(got, ref, rtol=1e-3, atol=2e-4, what=""):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert np.isfinite(got).all(), f"{what}: non-finite"
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=what)


def flat_module(module, dtype=torch.float32):
    """Give a stand-alone sub-module (Encoder, Decoder, ...) flat GPU parameter storage."""
    from omr_a2s_multimodal_transformer_amd.params import FlatParams
    module._test_flat = FlatParams(list(module.named_parameters()), torch.device(DEV), dtype)
    for mod in module.modules():
        for name, buf in list(mod._buffers.items()):
            if buf is not None:
                mod._buffers[name] = buf.to(DEV)
    return module


def load_seeded(module, shapes, seed, prefix=""):
    sd = syn.seeded_state_dict(shapes, seed)
    own = {k[len(prefix):]: v for k, v in sd.items()}
    missing, unexpected = module.load_state_dict(own, strict=False)
    assert not unexpected and all(m.endswith("pe") or m.endswith("pe_hwc") for m in missing), (missing, unexpected)
    return sd


def test_encoder_matches_reference_golden(golden):
    from omr_a2s_multimodal_transformer_amd.encoder import Encoder
    g = golden("f1_encoder")
    enc = Encoder(1).eval()
    load_seeded(enc, syn.encoder_shapes("encoder."), 13, "encoder.")
    flat_module(enc)
    for key, shape, seed in (("enc_a", (2, 1, 48, 80), 103), ("enc_b", (1, 1, 195, 64), 104)):
        out = enc(rnd(shape, seed).to(DEV))
        assert tuple(out.shape) == g[key].shape
        close(out, g[key], atol=5e-4, what=key)


@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("win", [-1, 3, 100])
def test_decoder_matches_reference_golden(golden, L, win):
    from omr_a2s_multimodal_transformer_amd.decoder import Decoder
    g = golden("f3_decoder")
    dec = Decoder(64, 32, 64, num_transformer_layers=L, attn_window=win).eval()
    load_seeded(dec, syn.decoder_shapes("decoder.", 64, layers=L), 20 + L, "decoder.")
    flat_module(dec)
    tgt = torch.from_numpy(g["tgt"]).to(DEV)
    mem = rnd((3, 20, 256), 301, -1, 1).to(DEV)
    close(dec(tgt, mem, torch.from_numpy(g["lens"]).to(DEV)), g[f"L{L}_w{win}_len"], what="int lengths (+1.0 additive)")
    close(dec(tgt, mem, torch.from_numpy(g["bmask"]).to(DEV)), g[f"L{L}_w{win}_bool"], what="bool mask (-inf)")
    close(dec(tgt, mem, None), g[f"L{L}_w{win}_none"], what="no mask")


def test_cross_attention_and_mixers_match_reference_golden(golden):
    from omr_a2s_multimodal_transformer_amd.model import CrossAttention, MultimodalTransformer
    g = golden("f4_cross_attention")
    ca = CrossAttention(256).eval()
    load_seeded(ca, syn.mha_shapes("cross_attn.attention.", 256), 31, "cross_attn.")
    flat_module(ca)
    q, kv = rnd((3, 9, 256), 401, -1, 1).to(DEV), rnd((3, 11, 256), 402, -1, 1).to(DEV)
    lq, lkv = torch.from_numpy(g["lq"]), torch.from_numpy(g["lkv"])
    close(ca(q, lq, kv, lkv)[0], g["ca_masked"], what="quirk-2 mask tiling")
    close(ca(q, None, kv, None)[0], g["ca_nomask"], what="no mask")
    w2i, i2w = syn.make_vocab(40)
    xi, xa = rnd((3, 11, 256), 403, -1, 1).to(DEV), rnd((3, 9, 256), 404, -1, 1).to(DEV)
    xli, xla = torch.from_numpy(g["xli"]), torch.from_numpy(g["xla"])
    for mt in ("concat", "attn_img", "attn_audio", "attn_both"):
        m = MultimodalTransformer(32, 64, 32, 64, 16, w2i, i2w, mixer_type=mt).eval()
        if mt != "concat":
            load_seeded(m.cross_attn, syn.mha_shapes("cross_attn.attention.", 256), 31, "cross_attn.")
        m.flatten_parameters()
        x, xl = m.mixer(xi=xi, xa=xa, xli=xli, xla=xla)
        close(x, g[f"mix_{mt}_x"], what=f"mixer {mt}")
        np.testing.assert_array_equal(xl.cpu().numpy(), g[f"mix_{mt}_xl"])
        x2, xl2 = m.mixer(xi=xi, xa=xa, xli=None, xla=None)
        close(x2, g[f"mix_{mt}_x_nolen"], what=f"mixer {mt} (no lengths)")
        assert xl2 is None


def grad_norms(model, names):
    ps = dict(model.named_parameters())
    return np.array([float(ps[n].grad.detach().double().norm()) for n in names])


def make_transformer(V, cfg, seed, hw=(32, 64), max_seq=16, win=-1):
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    w2i, i2w = syn.make_vocab(V)
    m = Transformer(hw[0], hw[1], max_seq, w2i, i2w, attn_window=win, config=cfg)
    load_seeded(m, syn.transformer_shapes(V, cfg.d_model, cfg.ff_dim, cfg.num_layers), seed)
    m.flatten_parameters()
    return m, w2i


def test_transformer_forward_backward_matches_reference_golden(golden):
    g = golden("f5_transformer")
    V = 50
    m, w2i = make_transformer(V, ModelConfig(**NO_DROP), 41)
    m.train()
    random.seed(0)
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5)
    m.zero_grad()
    logits = m(x.to(DEV), xl.to(DEV), y_in.to(DEV))
    assert tuple(logits.shape) == (2, V, 10)
    close(logits, g["logits"], what="logits [B,V,T]")
    loss = m.compute_loss(logits, y_out.to(DEV))
    close(loss, g["loss"], rtol=1e-4, what="loss")
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    assert names == [n for n, _ in m.named_parameters()]
    got = grad_norms(m, names)
    np.testing.assert_allclose(got, g["grad_norms"], rtol=2e-3, atol=1e-6)
    ps = dict(m.named_parameters())
    for n, head in zip(names, g["grad_heads"]):
        k = min(8, ps[n].numel())
        np.testing.assert_allclose(ps[n].grad.detach().flatten()[:k].cpu().numpy(), head[:k], rtol=5e-3, atol=2e-6, err_msg=n)


def test_adam_steps_match_reference_golden(golden):
    g = golden("f5_transformer")
    V = 50
    m, w2i = make_transformer(V, ModelConfig(**NO_DROP), 41)
    m.train()
    x, xl, y_in, y_out = (t.to(DEV) for t in syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5))
    opt = m.configure_optimizers()
    losses = []
    for _ in range(3):
        random.seed(0)
        opt.zero_grad()
        loss = m.compute_loss(m(x, xl, y_in), y_out)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g["adam_losses"], rtol=1e-3)
    ps = dict(m.named_parameters())
    for k, s, h in zip(g["adam_sel"], g["adam_sums"], g["adam_heads"]):
        t = ps[str(k)].detach()
        np.testing.assert_allclose(float(t.double().sum()), s, rtol=1e-4, atol=1e-3, err_msg=str(k))
        np.testing.assert_allclose(t.flatten()[:8].cpu().numpy(), h, rtol=1e-3, atol=1e-5, err_msg=str(k))


@pytest.mark.parametrize("mt,modality", [("concat", "both"), ("attn_img", "both"), ("attn_audio", "both"),
                                         ("attn_both", "both"), ("attn_both", "image"), ("attn_both", "audio")])
def test_multimodal_matches_reference_golden(golden, mt, modality):
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer
    g = golden("f5_multimodal")
    V = 40
    w2i, i2w = syn.make_vocab(V)
    m = MultimodalTransformer(32, 48, 35, 40, 12, w2i, i2w, mixer_type=mt, config=ModelConfig(**NO_DROP))
    load_seeded(m, syn.multimodal_shapes(V, mt), 51)
    m.flatten_parameters()
    m.train()
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, 32, 48, 9, V, w2i["<sos>"], w2i["<eos>"], seed=6)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, 35, 40, 9, V, w2i["<sos>"], w2i["<eos>"], seed=7, pad_value=0.0)
    m.apply_teacher_forcing_modality = lambda: modality
    m.zero_grad()
    logits = m(xi.to(DEV), xli.to(DEV), xa.to(DEV), xla.to(DEV), y_in.to(DEV), apply_teacher_forcing_modality=True)
    close(logits, g[f"{mt}_{modality}_logits"], what="logits")
    loss = m.compute_loss(logits, y_out.to(DEV))
    close(loss, g[f"{mt}_{modality}_loss"], rtol=1e-4, what="loss")
    loss.backward()
    names = [str(n) for n in g[f"{mt}_{modality}_grad_names"]]
    assert names == [n for n, _ in m.named_parameters()]
    ref = g[f"{mt}_{modality}_grad_norms"]
    got = grad_norms(m, names)
    ref = np.where(ref < 0, 0.0, ref)  # reference: unused parameters have grad None; here their flat slice stays zero
    # Hard bound 1e-2 on every tensor, 1e-3 on the typical one.  A near-zero ReLU pre-activation that flips its mask between
    # two fp32 implementations moves a layer's gradient (and everything upstream) by ~1/sqrt(N) of its norm -- measured between
    # the reference and the oracle themselves, both torch CPU (tools/relu_flip_probe.py; tests/test_dropout_parity_gpu.py uses
    # an fp64 run as the arbiter).  The non-degenerate shapes are in tests/test_round2_gpu.py.
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-6)
    live = ref > 0
    assert np.median(np.abs(got[live] - ref[live]) / ref[live]) < 1e-3


@pytest.mark.parametrize("win", [-1, 4])
def test_greedy_decode_tokens_match_reference_golden(golden, win):
    g = golden("f7_decode")
    V = 30
    m, w2i = make_transformer(V, ModelConfig(), 61, hw=(32, 96), max_seq=14, win=win)
    m.eval()
    x = rnd((1, 1, 32, 96), 701)
    y = torch.tensor([[w2i["<sos>"], 5, 6, w2i["<eos>"]]])
    m.validation_step((x.to(DEV), y), 0)
    toks = np.array([w2i[w] for w in m.YHat[0]])
    np.testing.assert_array_equal(toks, g[f"w{win}_tokens"])          # bit-exact token ids
    assert float(np.min(g[f"w{win}_margin"])) > 1e-4                    # the fixture's top-1 margins make that meaningful
    seq, probs = m.get_pred_seq_and_pred_prob_seq(x.to(DEV))
    assert seq == m.YHat[0]
    np.testing.assert_allclose(np.array(probs), g[f"w{win}_top1"], rtol=1e-3, atol=1e-4)
    # KV-cached steps (default) and the reference-style full re-run give the same tokens and top-1 logits
    seq_nc, probs_nc = m._greedy(m.encode(x.to(DEV)), want_probs=True, use_cache=False)
    assert seq_nc == seq
    np.testing.assert_allclose(np.array(probs_nc), np.array(probs), rtol=1e-4, atol=1e-5)
    m.Y.append(["a"])  # metric hook keys
    m.YHat.append(["a"])
    assert set(m.on_validation_epoch_end().keys()) == {"sym-er", "seq-er"}


def test_c1_variant_matches_reference_golden(golden):
    g = golden("f9_c1")
    V = 45
    cfg = ModelConfig(d_model=128, ff_dim=128, num_layers=2, **NO_DROP)
    m, _ = make_transformer(V, cfg, 71)
    m.train()
    x, xl, y_in, y_out = (t.to(DEV) for t in syn.synthetic_unimodal_batch(2, 32, 64, 10, V, 44, 43, seed=8))
    m.zero_grad()
    logits = m(x, xl, y_in)
    close(logits, g["logits"], what="C1 logits")
    loss = m.compute_loss(logits, y_out)
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    np.testing.assert_allclose(grad_norms(m, names), g["grad_norms"], rtol=2e-3, atol=1e-6)


def test_bf16_step_tracks_fp32_oracle():
    """bf16 compute / fp32 master mode: loss within 3e-2 relative of the fp32 CPU oracle on the same batch."""
    V = 50
    cfg = ModelConfig(compute_dtype="bf16", **NO_DROP)
    m, w2i = make_transformer(V, cfg, 41)
    assert m.compute_dtype() == torch.bfloat16
    m.train()
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5)
    sd = syn.seeded_state_dict(syn.transformer_shapes(V), 41)
    ref = R.ce_loss(R.transformer_forward(sd, x, xl, y_in, R.OracleCfg(), 32, 64), y_out)
    opt = m.configure_optimizers()
    opt.zero_grad()
    loss = m.compute_loss(m(x.to(DEV), xl.to(DEV), y_in.to(DEV)), y_out.to(DEV))
    loss.backward()
    opt.step()
    assert abs(float(loss) - float(ref)) / float(ref) < 3e-2
    assert torch.isfinite(m._flat.master).all() and torch.isfinite(m._flat.grad).all()


def test_training_mode_with_dropout_runs_and_learns():
    """Dropout ON (reference defaults 0.5 / 0.1): loss is finite and decreases over a few Adam steps on one batch."""
    V = 50
    m, w2i = make_transformer(V, ModelConfig(), 41)
    m.train()
    m.teacher_forcing_prob = 0.0
    batch = tuple(t.to(DEV) for t in syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5))
    opt = m.configure_optimizers()
    opt.param_groups[0]["lr"] = 1e-3
    losses = []
    for i in range(8):
        opt.zero_grad()
        loss = m.training_step(batch, i)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_checkpoint_roundtrip(tmp_path):
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    V = 30
    m, w2i = make_transformer(V, ModelConfig(d_model=128, ff_dim=128, num_layers=1), 3)
    p = str(tmp_path / "m.ckpt")
    m.save_checkpoint(p)
    m2 = Transformer.load_from_checkpoint(p)
    m2.flatten_parameters()
    for (n1, a), (n2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert n1 == n2 and torch.equal(a.cpu(), b.cpu())
    assert m2.config.d_model == 128


def test_kv_cached_decode_matches_full_rerun_bf16_long():
    """Longer sequence, attention window, bf16: cached and uncached greedy decoding agree token for token."""
    V = 60
    cfg = ModelConfig(d_model=128, ff_dim=128, num_layers=2, compute_dtype="bf16")
    m, w2i = make_transformer(V, cfg, 9, hw=(32, 128), max_seq=40, win=7)
    m.eval()
    mem = m.encode(rnd((1, 1, 32, 128), 702).to(DEV))
    a, _ = m._greedy(mem, use_cache=True)
    b, _ = m._greedy(mem, use_cache=False)
    assert a == b and len(a) >= 1


def test_decoder_bucket_is_final_when_backward_crosses_the_memory_boundary():
    """DDP overlap contract (ddp.py): the decoder bucket's all-reduce is enqueued when backward reaches the encoder->decoder
    memory hand-off.  Every decoder-side gradient -- including the embedding's and the fused cross-attention K|V
    projection's, whose nodes are not ancestors of the memory gradient -- must already be final at that moment."""
    from omr_a2s_multimodal_transformer_amd.ddp import GradReducer
    from omr_a2s_multimodal_transformer_amd.runtime import WgradStream

    class Probe(GradReducer):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.snap = {}

        def reduce_bucket(self, i):
            if i not in self.snap:
                WgradStream.join()              # what GradReducer.reduce_bucket does before it records its event
                b, e = self.buckets[i]
                self.snap[i] = self.flat.grad[b:e].clone()

    V = 50
    m, w2i = make_transformer(V, ModelConfig(num_layers=3, **NO_DROP), 41)
    m.train()
    m.teacher_forcing_prob = 0.0
    real = m.attach_reducer()
    probe = Probe(m._flat, None, real.buckets)
    m._reducer = probe
    x, xl, y_in, y_out = (t.to(DEV) for t in syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5))
    m.zero_grad()
    m.compute_loss(m(x, xl, y_in), y_out).backward()
    assert 1 in probe.snap and 0 not in probe.snap          # only the decoder bucket fired during backward
    b, e = probe.buckets[1]
    final = m._flat.grad[b:e]
    assert final.abs().max() > 0
    assert torch.equal(probe.snap[1], final), "a decoder gradient changed after the bucket was handed to the all-reduce"


@pytest.mark.parametrize("mt,modality", [("concat", "both"), ("attn_both", "both"), ("attn_img", "both"), ("attn_both", "image")])
def test_multimodal_decoder_bucket_is_final_at_the_two_memory_boundary(mt, modality):
    """MultimodalTransformer under DDP: the decoder-side bucket (decoder + mixer attention) is handed to the all-reduce
    when backward has the gradients of BOTH encoder memories (one GradBoundary node over the pair) -- also in the steps
    that drop a modality, where one memory has no gradient at all."""
    from omr_a2s_multimodal_transformer_amd.ddp import GradReducer
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer
    from omr_a2s_multimodal_transformer_amd.runtime import WgradStream

    class Probe(GradReducer):
        snap = None

        def reduce_bucket(self, i):
            if i == 1 and self.snap is None:
                WgradStream.join()
                b, e = self.buckets[1]
                self.snap = self.flat.grad[b:e].clone()

    V = 40
    w2i, i2w = syn.make_vocab(V)
    m = MultimodalTransformer(32, 48, 35, 40, 12, w2i, i2w, mixer_type=mt, config=ModelConfig(num_layers=2, **NO_DROP))
    load_seeded(m, syn.multimodal_shapes(V, mt, layers=2), 52)
    m.flatten_parameters()
    m.train()
    real = m.attach_reducer()
    probe = Probe(m._flat, None, real.buckets)
    m._reducer = probe
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, 32, 48, 9, V, w2i["<sos>"], w2i["<eos>"], seed=6)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, 35, 40, 9, V, w2i["<sos>"], w2i["<eos>"], seed=7, pad_value=0.0)
    m.apply_teacher_forcing_modality = lambda: modality
    m.zero_grad()
    logits = m(xi.to(DEV), xli.to(DEV), xa.to(DEV), xla.to(DEV), y_in.to(DEV), apply_teacher_forcing_modality=True)
    m.compute_loss(logits, y_out.to(DEV)).backward()
    assert probe.snap is not None, "backward never crossed the memory boundary"
    b, e = probe.buckets[1]
    final = m._flat.grad[b:e]
    assert final.abs().max() > 0
    assert torch.equal(probe.snap, final), "a decoder-side gradient changed after the bucket was handed to the all-reduce"
    eb, ee = probe.buckets[0]
    assert m._flat.grad[eb:ee].abs().max() > 0          # the encoders did get their gradients afterwards


def test_side_stream_weight_gradients_are_complete_and_equal_to_in_order_ones():
    """runtime.WgradStream: weight-gradient GEMMs / depthwise weight gradients issued on the side stream must all have
    landed in the flat gradient buffer when backward returns (engine callback joins the streams), and must equal the
    gradients of the same pass with everything on one stream (up to the order of the fp32 atomics)."""
    from omr_a2s_multimodal_transformer_amd.runtime import WgradStream
    V = 60
    m, w2i = make_transformer(V, ModelConfig(num_layers=2, **NO_DROP), 43, hw=(64, 128), max_seq=24)
    m.train()
    m.teacher_forcing_prob = 0.0
    x, xl, y_in, y_out = (t.to(DEV) for t in syn.synthetic_unimodal_batch(3, 64, 128, 20, V, w2i["<sos>"], w2i["<eos>"], seed=6))
    grads = {}
    try:
        for mode in (False, False, True, True):
            WgradStream.enabled = mode
            m.zero_grad()
            m.compute_loss(m(x, xl, y_in), y_out).backward()
            assert not WgradStream._pending, "backward returned with un-joined side-stream work"
            grads.setdefault(mode, []).append(m._flat.grad.clone())
    finally:
        WgradStream.enabled = True
    ref = grads[False][0]
    assert ref.abs().max() > 0
    # Every gradient must agree to fp32-atomics noise, encoder included: the InstanceNorm statistics are a fixed-order slot
    # reduction (no atomics), so the forward pass and the data-gradient chain are bit-identical from run to run and only the
    # weight-gradient sums (fp32 atomics, order varies) differ in their last bits.
    for g in grads[True] + grads[False][1:]:
        for n, (o, c) in m._flat.offsets.items():
            r = ref[o:o + c]
            if r.abs().max() == 0:
                continue
            assert g[o:o + c].abs().max() > 0, f"{n}: gradient missing"
            rel = ((g[o:o + c] - r).norm() / r.norm()).item()
            assert rel < 1e-5, (n, rel)


@pytest.mark.parametrize("dtype,win", [("fp32", -1), ("fp32", 4), ("bf16", -1)])
def test_batched_greedy_equals_per_sample_greedy(dtype, win):
    """KV-cached greedy decode batched over same-sized samples gives, for every sample, exactly
    the token sequence of the reference-style bs = 1 loop (rows of a batch never interact)."""
    V = 30
    m, w2i = make_transformer(V, ModelConfig(compute_dtype=dtype), 61, hw=(32, 96), max_seq=14, win=win)
    m.eval()
    xs = rnd((5, 1, 32, 96), 702).to(DEV)
    mem = m.encode(xs)
    batched = m.greedy_batch(mem, sync_every=3)
    assert len(batched) == 5
    for i in range(5):
        single, _ = m._greedy(m.encode(xs[i:i + 1]))
        assert batched[i] == single, (i, batched[i], single)
    # the rows differ (a random-init model may still pick the same tokens): the batched step reproduces each sample's own
    # next-token logits to the bit, so no row can have been mixed up with another
    sos = w2i["<sos>"]
    st_b = m.decoder.init_decode(mem)
    st_1 = [m.decoder.init_decode(mem[i:i + 1].contiguous()) for i in range(5)]
    tok = torch.full((5, 1), sos, dtype=torch.int64, device=DEV)
    for _ in range(3):
        lb = m.decoder.decode_step(tok, st_b)
        rows = [m.decoder.decode_step(tok[i:i + 1], st_1[i]) for i in range(5)]
        for i in range(5):
            assert torch.equal(lb[i], rows[i])
        assert not torch.equal(lb[0], lb[1])
        tok = lb.argmax(dim=1, keepdim=True)


@pytest.mark.parametrize("win", [-1, 4])
def test_beam_search_beam1_is_greedy_and_wider_beams_do_not_score_lower(win):
    """Beam search (BASELINE C5 extension): beam = 1 reproduces the greedy tokens; a wider beam never returns a lower score."""
    V = 30
    m, w2i = make_transformer(V, ModelConfig(), 61, hw=(32, 96), max_seq=14, win=win)
    m.eval()
    mem = m.encode(rnd((1, 1, 32, 96), 703).to(DEV))
    greedy, _ = m._greedy(mem)
    seq1, score1 = m.beam_search(mem, beam=1)
    assert seq1 == greedy
    seq4, score4 = m.beam_search(mem, beam=4)
    assert score4 >= score1 - 1e-5 and len(seq4) >= 1


def test_topk_logprob_matches_torch():
    from omr_a2s_multimodal_transformer_amd import kernels as K
    x = (rnd((5, 6997), 704) * 8 - 4).to(DEV)
    x[2, 100] = x[2, 7]                      # a tie: the smaller index must come first
    idx, val = K.topk_logprob(x, 6)
    ref = torch.log_softmax(x, dim=1)
    rv, ri = torch.sort(ref, dim=1, descending=True, stable=True)
    assert torch.equal(idx, ri[:, :6])
    torch.testing.assert_close(val, rv[:, :6], rtol=1e-5, atol=1e-5)
    assert torch.equal(K.argmax(x)[0], idx[:, 0])
