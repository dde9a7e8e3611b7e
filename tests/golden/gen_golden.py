"""Generate golden vectors by IMPORTING THE REFERENCE (build container only; never on the GPU box).

    cd /tmp/scratch && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden.py

Writes tests/golden/*.npz.  Weights and inputs are produced by the repo's own seeded generators
(omr_a2s_multimodal_transformer_amd.synthetic) and loaded INTO the reference modules with
load_state_dict, so fixtures carry seeds + expected outputs, not weights.  Absent non-arithmetic
packages (lightning, torchinfo, librosa, torchvision, music21, pyMV2H, midi2audio) are stubbed as
SURVEY.md Appendix B describes; every tensor op runs in the reference's own code + torch CPU.
"""
import importlib.machinery
import math
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

import datasets  # noqa: F401,E402  (real package; must be imported before torchvision is stubbed)


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    @property
    def device(self):
        return next(self.parameters()).device


stub("lightning")
stub("lightning.pytorch", LightningModule=LightningModule, LightningDataModule=object)
stub("torchinfo", summary=lambda *a, **k: None)
stub("librosa")
tv = stub("torchvision")
tv.transforms = stub("torchvision.transforms", ToTensor=lambda: (lambda x: x))
stub("music21", converter=None)
stub("midi2audio", FluidSynth=None)
for n in ("pyMV2H", "pyMV2H.converter", "pyMV2H.metrics", "pyMV2H.utils"):
    stub(n)
stub("pyMV2H.converter.midi_converter", MidiConverter=None)
stub("pyMV2H.metrics.mv2h", mv2h=None)
stub("pyMV2H.utils.music", Music=None)
stub("pyMV2H.utils.mv2h", MV2H=None)

from src.transformer.decoder import Decoder, PositionalEncoding1D  # noqa: E402
from src.transformer.encoder import ConvBlock, DSCBlock, Encoder  # noqa: E402
from src.transformer.model import CrossAttention, MultimodalTransformer, PositionalEncoding2D, Transformer  # noqa: E402
from src.utils.metrics import compute_ed_metrics  # noqa: E402
from src.data.preprocessing import ar_batch_preparation_image, ar_batch_preparation_audio  # noqa: E402

from omr_a2s_multimodal_transformer_amd import synthetic as syn  # noqa: E402


def rnd(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def load_seeded(module: nn.Module, seed: int, prefix: str = ""):
    """Fill a reference module with the repo's seeded test weights (buffers such as pe kept)."""
    shapes = {prefix + k: tuple(v.shape) for k, v in module.named_parameters()}
    sd = syn.seeded_state_dict(shapes, seed, mode="test")
    own = {k[len(prefix):]: v for k, v in sd.items()}
    missing, unexpected = module.load_state_dict(own, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith(".pe") for m in missing), missing
    return sd


def zero_dropout(m: nn.Module):
    for mod in m.modules():
        if isinstance(mod, (nn.Dropout, nn.Dropout2d)):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


@torch.no_grad()
def f1_encoder():
    """F1: ConvBlock / DSCBlock / Encoder in eval mode on odd sizes."""
    random.seed(0)
    cb = ConvBlock(16, 32, stride=(2, 2)).eval()
    load_seeded(cb, 11, "cb.")
    x = rnd((2, 16, 13, 19), 101, -1, 1)
    db = DSCBlock(128, 128, stride=(1, 1)).eval()
    load_seeded(db, 12, "db.")
    xd = rnd((2, 128, 5, 7), 102, -1, 1)
    enc = Encoder(1).eval()
    load_seeded(enc, 13, "encoder.")
    xa = rnd((2, 1, 48, 80), 103)
    xb = rnd((1, 1, 195, 64), 104)
    save("f1_encoder", cb_out=cb(x), db_out=db(xd), enc_a=enc(xa), enc_b=enc(xb))


def f2_pe():
    save("f2_pe", pe2d=PositionalEncoding2D(256, 4, 6).pe, pe2d_128=PositionalEncoding2D(128, 3, 5).pe,
         pe1d=PositionalEncoding1D(32, 256).pe, pe1d_128=PositionalEncoding1D(16, 128).pe)


@torch.no_grad()
def f3_decoder():
    """F3: Decoder eval forward; (i) int lengths (+1.0 additive), (ii) bool mask, (iii) None; windows."""
    V, T, S, B = 64, 12, 20, 3
    out = {}
    g = torch.Generator().manual_seed(7)
    tgt = torch.randint(1, V, (B, T), generator=g)
    tgt[1, 9:] = 0
    tgt[2, 5:] = 0
    mem = rnd((B, S, 256), 301, -1, 1)
    lens = torch.tensor([20, 13, 7], dtype=torch.int32)
    bmask = torch.zeros(B, S, dtype=torch.bool)
    bmask[1, 13:] = True
    bmask[2, 7:] = True
    bmask[0, 3:6] = True
    out.update(tgt=tgt, lens=lens, bmask=bmask)
    for L in (1, 2):
        for win in (-1, 3, 100):
            dec = Decoder(V, 32, V, num_transformer_layers=L, attn_window=win).eval()
            load_seeded(dec, 20 + L, "decoder.")
            out[f"L{L}_w{win}_len"] = dec(tgt, mem, lens)
            out[f"L{L}_w{win}_bool"] = dec(tgt, mem, bmask)
            out[f"L{L}_w{win}_none"] = dec(tgt, mem, None)
    save("f3_decoder", **out)


@torch.no_grad()
def f4_cross_attention():
    """F4: CrossAttention quirk-2 mask tiling (B=3) and the four mixers incl. attn_both (quirk 3)."""
    ca = CrossAttention(256).eval()
    load_seeded(ca, 31, "cross_attn.")
    q = rnd((3, 9, 256), 401, -1, 1)
    kv = rnd((3, 11, 256), 402, -1, 1)
    lq = torch.tensor([5, 3, 2], dtype=torch.int32)
    lkv = torch.tensor([7, 4, 2], dtype=torch.int32)
    out = dict(ca_masked=ca(q, lq, kv, lkv)[0], ca_nomask=ca(q, None, kv, None)[0], lq=lq, lkv=lkv)
    w2i, i2w = syn.make_vocab(40)
    xi = rnd((3, 11, 256), 403, -1, 1)
    xa = rnd((3, 9, 256), 404, -1, 1)
    xli = torch.tensor([11, 6, 4], dtype=torch.int32)
    xla = torch.tensor([9, 5, 3], dtype=torch.int32)
    out.update(xli=xli, xla=xla)
    for mt in ("concat", "attn_img", "attn_audio", "attn_both"):
        random.seed(0)
        m = MultimodalTransformer(32, 64, 32, 64, 16, w2i, i2w, mixer_type=mt).eval()
        if mt != "concat":
            load_seeded(m.cross_attn, 31, "cross_attn.")
        x, xl = m.mixer(xi=xi, xa=xa, xli=xli, xla=xla)
        out[f"mix_{mt}_x"] = x
        out[f"mix_{mt}_xl"] = xl
        x2, xl2 = m.mixer(xi=xi, xa=xa, xli=None, xla=None)
        out[f"mix_{mt}_x_nolen"] = x2
        assert xl2 is None
    save("f4_cross_attention", **out)


def grads_summary(module: nn.Module, prefix=""):
    names, norms, heads = [], [], []
    for k, p in module.named_parameters():
        names.append(prefix + k)
        if p.grad is None:
            norms.append(-1.0)
            heads.append(np.zeros(8, np.float32))
        else:
            g = p.grad.detach().flatten()
            norms.append(float(g.double().norm()))
            h = np.zeros(8, np.float32)
            h[: min(8, g.numel())] = g[:8].numpy()
            heads.append(h)
    return np.array(names), np.array(norms), np.stack(heads)


def f5_f6_transformer():
    """F5: full Transformer.forward + CE + gradients, all dropout p=0 in TRAIN mode.
    F6: three Adam steps on a fixed batch (loss trajectory + parameter checksums)."""
    V = 50
    w2i, i2w = syn.make_vocab(V)
    random.seed(0)
    torch.manual_seed(0)
    m = Transformer(32, 64, 16, w2i, i2w, attn_window=-1)
    load_seeded(m, 41)
    zero_dropout(m)
    m.train()
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, w2i["<sos>"], w2i["<eos>"], seed=5)
    logits = m(x, xl, y_in)
    loss = m.compute_loss(logits, y_out)
    loss.backward()
    names, norms, heads = grads_summary(m)
    out = dict(logits=logits, loss=loss, grad_names=names, grad_norms=norms, grad_heads=heads, xl=xl, y_in=y_in, y_out=y_out)
    # F6
    load_seeded(m, 41)
    opt = m.configure_optimizers()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        l = m.compute_loss(m(x, xl, y_in), y_out)
        l.backward()
        opt.step()
        losses.append(float(l))
    sel = ["encoder.conv_blocks.0.conv1.weight", "encoder.dscblocks.3.conv3.point_conv.weight",
           "decoder.embedding.weight", "decoder.transformer_decoder.layers.7.linear2.weight",
           "decoder.transformer_decoder.layers.0.norm2.bias", "decoder.out_layer.bias"]
    sd = dict(m.named_parameters())
    out["adam_losses"] = np.array(losses)
    out["adam_sel"] = np.array(sel)
    out["adam_sums"] = np.array([float(sd[k].detach().double().sum()) for k in sel])
    out["adam_heads"] = np.stack([sd[k].detach().flatten()[:8].numpy() for k in sel])
    save("f5_transformer", **out)


def f5_multimodal():
    V = 40
    w2i, i2w = syn.make_vocab(V)
    xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, 32, 48, 9, V, w2i["<sos>"], w2i["<eos>"], seed=6)
    xa, xla, _, _ = syn.synthetic_unimodal_batch(3, 35, 40, 9, V, w2i["<sos>"], w2i["<eos>"], seed=7, pad_value=0.0)
    out = dict(xli=xli, xla=xla, y_in=y_in, y_out=y_out)
    for mt in ("concat", "attn_img", "attn_audio", "attn_both"):
        random.seed(0)
        m = MultimodalTransformer(32, 48, 35, 40, 12, w2i, i2w, mixer_type=mt)
        load_seeded(m, 51)
        zero_dropout(m)
        m.train()
        for modality in ("both", "image", "audio"):
            if modality != "both" and mt != "attn_both":
                continue  # single-modality branch is mixer independent; pin it once
            m.zero_grad()
            m.apply_teacher_forcing_modality = (lambda mod=modality: mod)
            logits = m(xi, xli, xa, xla, y_in, apply_teacher_forcing_modality=True)
            loss = m.compute_loss(logits, y_out)
            loss.backward()
            names, norms, _ = grads_summary(m)
            out[f"{mt}_{modality}_logits"] = logits
            out[f"{mt}_{modality}_loss"] = loss
            out[f"{mt}_{modality}_grad_names"] = names
            out[f"{mt}_{modality}_grad_norms"] = norms
    save("f5_multimodal", **out)


@torch.no_grad()
def f7_decode():
    """F7: validation_step greedy decode token ids; get_pred_seq_and_pred_prob_seq logits."""
    V = 30
    w2i, i2w = syn.make_vocab(V)
    out = {}
    for win in (-1, 4):
        random.seed(0)
        m = Transformer(32, 96, 14, w2i, i2w, attn_window=win).eval()
        load_seeded(m, 61)
        x = rnd((1, 1, 32, 96), 701)
        y = torch.tensor([[w2i["<sos>"], 5, 6, w2i["<eos>"]]])
        m.validation_step((x, y), 0)
        words = m.YHat[0]
        out[f"w{win}_tokens"] = np.array([w2i[w] for w in words])
        seq, probs = m.get_pred_seq_and_pred_prob_seq(x)
        assert seq == words
        out[f"w{win}_top1"] = np.array(probs)
        # margins (top1-top2) per step, re-running the decoder on the final prefix
        mem = m.pos_2d(m.encoder(x)).flatten(2).permute(0, 2, 1).contiguous()
        y_in = torch.tensor([[w2i["<sos>"]] + [w2i[w] for w in words[:-1]]])
        lg = m.decoder(y_in, mem, None)[0]  # [V, T]
        t2 = lg.topk(2, dim=0).values
        out[f"w{win}_margin"] = (t2[0] - t2[1])
    save("f7_decode", **out)


def f8_metrics():
    cases = [([["a", "b", "c"]], [["a", "c"]]),
             ([["a", "b"], ["c", "d", "e"]], [["a", "b"], ["c", "x", "e", "f"]]),
             ([["x"] * 5, ["y"]], [[], ["y"]])]
    sym = [compute_ed_metrics(t, p)["sym-er"] for t, p in cases]
    seq = [compute_ed_metrics(t, p)["seq-er"] for t, p in cases]
    save("f8_metrics", sym=np.array(sym), seq=np.array(seq))


def f9_c1_variant():
    """F9: the parameterised C1 variant (d=128, L=2) composed from reference classes."""
    V = 45
    random.seed(0)
    enc = Encoder(1)
    enc.dscblocks[3] = DSCBlock(128, 128, stride=(1, 1))
    pos = PositionalEncoding2D(128, 2, 8)
    dec = Decoder(V, 16, V, embedding_dim=128, ff_dim=128, num_transformer_layers=2)
    load_seeded(enc, 71, "encoder.")
    load_seeded(dec, 71, "decoder.")
    for mod in (enc, pos, dec):
        zero_dropout(mod)
        mod.train()
    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(2, 32, 64, 10, V, 44, 43, seed=8)
    f = pos(enc(x))
    mem = f.flatten(2).permute(0, 2, 1).contiguous()
    logits = dec(y_in, mem, xl)
    loss = nn.CrossEntropyLoss(ignore_index=0)(logits, y_out)
    loss.backward()
    n1, g1, _ = grads_summary(enc, "encoder.")
    n2, g2, _ = grads_summary(dec, "decoder.")
    save("f9_c1", logits=logits, loss=loss, grad_names=np.concatenate([n1, n2]), grad_norms=np.concatenate([g1, g2]),
         xl=xl, y_in=y_in, y_out=y_out)


def f10_collate():
    g = torch.Generator().manual_seed(3)
    items = []
    for h, w, n in ((10, 17, 5), (12, 9, 7), (7, 20, 3)):
        items.append((torch.rand((1, h, w), generator=g), w, torch.randint(1, 9, (n,), generator=g)))
    xi, xli, yi, yo = ar_batch_preparation_image(items)
    xa, xla, _, _ = ar_batch_preparation_audio(items)
    save("f10_collate", xi=xi, xli=xli, yi=yi, yo=yo, xa=xa, xla=xla)


if __name__ == "__main__":
    torch.set_num_threads(8)
    f1_encoder()
    f2_pe()
    f3_decoder()
    f4_cross_attention()
    f5_f6_transformer()
    f5_multimodal()
    f7_decode()
    f8_metrics()
    f9_c1_variant()
    f10_collate()
