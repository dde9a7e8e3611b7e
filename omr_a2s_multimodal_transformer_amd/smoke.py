"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (used by __graft_entry__.smoke())."""
from __future__ import annotations

import random

import torch


def run() -> None:
    from oracle import ref_cpu as R  # checker only
    from . import synthetic as syn
    from .config import ModelConfig
    from .model import Transformer

    V, H, W, T = 60, 32, 96, 12
    cfg = ModelConfig(d_model=128, ff_dim=128, num_layers=2, dropout=0.0, encoder_dropout=0.0)
    w2i, i2w = syn.make_vocab(V)
    sd = syn.seeded_state_dict(syn.transformer_shapes(V, 128, 128, 2), 7)
    model = Transformer(H, W, 16, w2i, i2w, config=cfg)
    model.load_state_dict(sd, strict=False)
    model.flatten_parameters()
    model.train()
    random.seed(0)
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, H, W, T, V, w2i["<sos>"], w2i["<eos>"], seed=3)
    opt = model.configure_optimizers()
    opt.zero_grad()
    logits = model(x.cuda(), xl.cuda(), y_in.cuda())
    loss = model.compute_loss(logits, y_out.cuda())
    loss.backward()
    opt.step()
    torch.cuda.synchronize()

    for v in sd.values():
        v.requires_grad_(True)
    ocfg = R.OracleCfg(d_model=128, ff_dim=128, num_layers=2)
    ref_logits = R.transformer_forward(sd, x, xl, y_in, ocfg, H, W)
    ref_loss = R.ce_loss(ref_logits, y_out)
    ref_loss.backward()
    err = (logits.detach().float().cpu() - ref_logits.detach()).abs().max().item()
    rel = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    gname = "decoder.transformer_decoder.layers.0.linear1.weight"
    g_gpu = dict(model.named_parameters())[gname].grad.detach().double().norm().item()
    g_ref = sd[gname].grad.double().norm().item()
    print(f"smoke: loss {float(loss):.6f} (oracle {float(ref_loss):.6f}, rel {rel:.2e}); max|dlogits| {err:.2e}; "
          f"|grad {gname}| {g_gpu:.6e} vs {g_ref:.6e}")
    assert rel < 1e-3 and err < 1e-3 and abs(g_gpu - g_ref) / g_ref < 5e-3, "HIP path disagrees with the CPU oracle"
