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


def test_split_multimodal_checkpoint_into_unimodal_checkpoints(tmp_path):
    """the multimodal -> unimodal split (reference src/utils/split_multimodal_ckpt.py) on a
    synthetic Lightning-layout checkpoint: state-dict keys become exactly the unimodal Transformer's, hyper-parameters are
    renamed, the ModelCheckpoint callback points at the new file, tensors are untouched."""
    import torch
    from omr_a2s_multimodal_transformer_amd import synthetic as syn
    from omr_a2s_multimodal_transformer_amd.ckpt_tools import split_both_ckpt_in_two

    V = 40
    sd = syn.seeded_state_dict(syn.multimodal_shapes(V, "attn_both", d=64, ff=64, layers=2), 5)
    best = str(tmp_path / "run" / "best.ckpt")
    ck = {"state_dict": sd, "epoch": 3,
          "hyper_parameters": dict(max_img_height=64, max_img_width=128, max_audio_height=48, max_audio_width=256, max_seq_len=20,
                                   mixer_type="attn_both", teacher_forcing_modality_prob=0.5, attn_window=-1),
          "callbacks": {"ModelCheckpoint{'monitor': 'val_sym-er'}": {"best_model_path": best, "best_model_score": 0.25,
                                                                    "kth_best_model_path": best, "best_k_models": {best: 0.25}}}}
    src = str(tmp_path / "mm.ckpt")
    torch.save(ck, src)
    img_path, aud_path = split_both_ckpt_in_two(src)
    assert img_path.endswith("mm_only_image_distorted.ckpt") and aud_path.endswith("mm_only_audio.ckpt")
    uni_keys = set(syn.transformer_shapes(V, 64, 64, 2).keys())
    for path, mod, h, w, suffix in ((img_path, "image", 64, 128, "image_distorted"), (aud_path, "audio", 48, 256, "audio")):
        one = torch.load(path, weights_only=True)
        got = {k for k in one["state_dict"] if not k.endswith("pe") and not k.endswith("pe_hwc")}
        assert got == {k for k in uni_keys if not k.endswith("pe")}, (sorted(got ^ uni_keys))
        assert torch.equal(one["state_dict"]["encoder.conv_blocks.0.conv1.weight"], sd[f"{mod}_encoder.conv_blocks.0.conv1.weight"])
        assert torch.equal(one["state_dict"]["decoder.out_layer.weight"], sd["decoder.out_layer.weight"])
        hp = one["hyper_parameters"]
        assert hp["max_input_height"] == h and hp["max_input_width"] == w and hp["max_seq_len"] == 20
        assert not any(k in hp for k in ("mixer_type", "teacher_forcing_modality_prob", "max_img_height", "max_audio_width"))
        cb = next(iter(one["callbacks"].values()))
        assert cb["best_model_path"].endswith(f"best_only_{suffix}.ckpt") and cb["best_k_models"] == {cb["best_model_path"]: 0.25}
        assert one["epoch"] == 3


def test_oracle_log_stft_properties():
    """oracle.log_stft (restated librosa pipeline, parity unpinned -- librosa absent): shape contract of preprocessing.py:13-30,
    range [0, 1] with the global peak at exactly 1, and a pure tone landing in the right frequency bin."""
    import numpy as np
    from oracle import ref_cpu as R
    sr, n = 22050, 22050 * 2 + 100
    y = np.sin(2 * np.pi * 1000.0 * np.arange(n) / sr)
    s = R.log_stft(y)
    assert s.shape == (195, 1 + n // 512) and s.dtype == np.float32
    assert s.min() >= 0.0 and s.max() == 1.0
    assert abs(int(np.argmax(s[:, s.shape[1] // 2])) - round(1000.0 * 2048 / sr)) <= 1
    silent = R.log_stft(np.zeros(4096))
    assert np.all(silent == 1.0)          # amplitude_to_db of all-amin input: 0 dB everywhere -> 1.0 after /80 + 1
