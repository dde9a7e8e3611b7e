"""CPU-only host logic: mask builders, positional tables, config surface, metrics, synthetic batches, state-dict contract."""
import math

import numpy as np
import pytest
import torch

from omr_a2s_multimodal_transformer_amd import synthetic as syn
from omr_a2s_multimodal_transformer_amd.config import C1_TINY, ModelConfig
from omr_a2s_multimodal_transformer_amd.decoder import Decoder, sinusoid_1d
from omr_a2s_multimodal_transformer_amd.metrics import compute_ed_metrics, compute_metrics
from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer, Transformer, sinusoid_2d
from oracle import ref_cpu as R


def test_positional_tables_match_reference_golden(golden):
    g = golden("f2_pe")
    np.testing.assert_allclose(sinusoid_2d(256, 4, 6).numpy(), g["pe2d"], atol=1e-6)
    np.testing.assert_allclose(sinusoid_2d(128, 3, 5).numpy(), g["pe2d_128"], atol=1e-6)
    np.testing.assert_allclose(sinusoid_1d(32, 256).numpy(), g["pe1d"], atol=1e-6)
    np.testing.assert_allclose(sinusoid_1d(16, 128).numpy(), g["pe1d_128"], atol=1e-6)


def test_mask_builders_follow_reference_semantics():
    dec = Decoder(20, 16, 20, attn_window=3)
    mem = torch.zeros(3, 6, 256)
    assert dec.get_memory_key_padding_mask(mem, None) is None
    m = dec.get_memory_key_padding_mask(mem, torch.tensor([6, 4, 1], dtype=torch.int32))
    assert m.dtype == torch.float32                       # float 0/1 -> ADDED (+1.0), quirk 1 (decoder.py:186-188)
    assert m.tolist() == [[0] * 6, [0, 0, 0, 0, 1, 1], [0, 1, 1, 1, 1, 1]]
    b = torch.zeros(3, 6, dtype=torch.bool)
    b[1, 2:] = True
    mb = dec.get_memory_key_padding_mask(mem, b)
    assert mb.dtype == torch.bool and torch.equal(mb, b) and mb.data_ptr() != b.data_ptr()
    bias = Decoder._as_key_bias(mb)
    assert bias[1, 2].item() == float("-inf") and bias[1, 1].item() == 0.0
    with pytest.raises(AssertionError):
        dec.get_memory_key_padding_mask(mem, torch.zeros(2, 6, dtype=torch.bool))
    for T, w in ((7, 3), (5, 100), (6, 5)):
        np.testing.assert_array_equal(Decoder.create_variable_window_mask(T, w).numpy(), R.tgt_attn_mask(T, w).numpy())
    tm, pad = dec.get_tgt_masks(torch.tensor([[5, 6, 0, 0]]))
    assert pad.tolist() == [[0.0, 0.0, 1.0, 1.0]] and tm[2, 0].item() == 0.0 and tm[0, 1].item() == float("-inf")


def test_metrics_match_reference_golden(golden):
    g = golden("f8_metrics")
    cases = [([["a", "b", "c"]], [["a", "c"]]),
             ([["a", "b"], ["c", "d", "e"]], [["a", "b"], ["c", "x", "e", "f"]]),
             ([["x"] * 5, ["y"]], [[], ["y"]])]
    for i, (t, p) in enumerate(cases):
        m = compute_metrics(t, p)
        assert math.isclose(m["sym-er"], g["sym"][i]) and math.isclose(m["seq-er"], g["seq"][i])
    assert compute_ed_metrics([["a"]], [["a"]]) == {"sym-er": 0.0, "seq-er": 0.0}
    with pytest.raises(NotImplementedError):
        compute_metrics([["a"]], [["a"]], compute_mv2h=True)


def test_config_surface_roundtrip(tmp_path):
    c = ModelConfig(d_model=128, num_layers=2, compute_dtype="bf16")
    p = str(tmp_path / "config.json")
    c.save_pretrained(p)
    assert ModelConfig.from_pretrained(p) == c and ModelConfig.from_dict({**c.to_dict(), "unknown": 1}) == c
    assert ModelConfig() == ModelConfig(256, 4, 256, 8, 0.1, 0.5, "fp32")   # the reference's hard-coded values
    assert C1_TINY.d_model == 128 and C1_TINY.num_layers == 2


def test_state_dict_contract_and_param_counts():
    """Key names/shapes/order of the reference (SURVEY.md section 5, Appendix A param counts)."""
    w2i, i2w = syn.make_vocab(syn.GRANDSTAFF_VOCAB)
    m = Transformer(64, 128, 20, w2i, i2w)
    sd = m.state_dict()
    keys = [k for k in sd if not k.endswith(".pe")]
    assert keys == list(syn.transformer_shapes(syn.GRANDSTAFF_VOCAB).keys())
    assert tuple(sd["decoder.out_layer.weight"].shape) == (6997, 256, 1)
    assert tuple(sd["pos_2d.pe"].shape) == (1, 256, 4, 16) and tuple(sd["decoder.pos_1d.pe"].shape) == (1, 20, 256)
    n_enc = sum(p.numel() for p in m.encoder.parameters())
    n_dec = sum(p.numel() for p in m.decoder.parameters())
    assert (n_enc, n_dec) == (1263200, 8865109)
    assert float(sd["decoder.embedding.weight"][0].abs().sum()) == 0.0      # PAD row zero
    l0, l7 = "decoder.transformer_decoder.layers.0.", "decoder.transformer_decoder.layers.7."
    assert torch.equal(sd[l0 + "linear1.weight"], sd[l7 + "linear1.weight"])  # nn.TransformerDecoder deep-copies one layer
    mm = MultimodalTransformer(32, 48, 35, 40, 12, *syn.make_vocab(40), mixer_type="attn_img")
    assert [k for k in mm.state_dict() if not k.endswith(".pe")] == list(syn.multimodal_shapes(40, "attn_img").keys())
    with pytest.raises(ValueError):
        MultimodalTransformer(32, 48, 35, 40, 12, *syn.make_vocab(40), mixer_type="nope")
    assert set(m.hparams) >= {"max_input_height", "max_input_width", "max_seq_len", "w2i", "i2w", "ytest_i2w", "attn_window", "teacher_forcing_prob"}


def test_synthetic_batch_follows_collate_contract():
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(4, 48, 80, 12, 50, 49, 48, seed=1)
    assert x.dtype == torch.float32 and xl.dtype == torch.int32 and y_in.dtype == torch.int64 and y_out.dtype == torch.int64
    assert tuple(x.shape) == (4, 1, 48, 80) and int(xl[0]) == 3 * 10
    for i in range(4):
        n = int((y_in[i] != 0).sum())
        assert y_in[i, 0] == 49 and y_out[i, n - 1] == 48 and torch.equal(y_in[i, 1:n], y_out[i, : n - 1])
        w = int(xl[i]) // 3 * 8
        assert w >= 80 or float(x[i, 0, :, w:].min()) == 1.0   # right padding with the image pad value


def test_product_has_no_cpu_fallback():
    """The product path fails loudly on CPU tensors instead of silently computing elsewhere."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    with pytest.raises(RuntimeError, match="no CPU fallback|GPU only"):
        K.add(torch.zeros(4), torch.zeros(4))
