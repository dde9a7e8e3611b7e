"""Round-3 attention kernels (csrc/attention.hip): augmented k-step (bias / reference maximum / log-sum-exp inside the MFMA
chain), lagged reference maximum, dropout keep bits read from pre-generated words.  Reference math: torch's
multi_head_attention_forward as nn.TransformerDecoderLayer / CrossAttention use it (decoder.py:86-95, model.py:289-355)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * scale


@pytest.mark.parametrize("B,H,T,S", [(2, 4, 150, 300), (1, 2, 12, 70), (3, 1, 64, 64), (1, 4, 33, 4096 + 17)])
def test_dropout_words_are_the_byte_mask_in_the_kernels_layout(B, H, T, S):
    """omr_attn_dropout_words packs exactly omr_attn_dropout_mask's function: word (32-query block, 64-key tile, mb, r), bit
    (query & 31) + 32 * hh  <->  key tile*64 + mb*32 + (r & 3) + 8 * (r >> 2) + 4 * hh."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    p, seed = 0.3, 1234567
    mask = K.attn_dropout_mask(B, H, T, S, p, seed, DEV).cpu().numpy()          # [B,H,T,S] uint8
    words = K.attn_dropout_words(B, H, T, S, p, seed, DEV).cpu().numpy().view(np.uint64)
    nqb, nkt = (T + 31) // 32, (S + 63) // 64
    words = words.reshape(B * H, nqb, nkt, 2, 16)
    bits = ((words[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8)     # [..., mb, r, lane]
    r = np.arange(16)
    koff = (r & 3) + 8 * (r >> 2)                                                 # key offset of register r at hh = 0
    full = np.zeros((B * H, nqb * 32, nkt * 64), np.uint8)
    for mb in range(2):
        for hh in range(2):
            # bits[bh, qb, kt, mb, r, qo + 32*hh] -> full[bh, qb*32 + qo, kt*64 + mb*32 + koff[r] + 4*hh]
            blk = bits[:, :, :, mb, :, 32 * hh:32 * hh + 32]                      # [bh, qb, kt, r, qo]
            for ri in range(16):
                full[:, :, :].reshape(B * H, nqb, 32, nkt, 64)[:, :, :, :, mb * 32 + koff[ri] + 4 * hh] = blk[:, :, :, ri, :].transpose(0, 1, 3, 2)
    np.testing.assert_array_equal(full[:, :T, :S].reshape(B, H, T, S), mask)
    assert abs(mask.mean() - (1 - p)) < 0.02


def _ref_attention(q, k, v, H, key_bias, causal, mask):
    B, T, d = q.shape
    S, hd = k.shape[1], d // H
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    if causal:
        s = s + torch.full((T, S), float("-inf")).triu(1)
    p = torch.softmax(s, dim=-1)
    if mask is not None:
        p = p * mask
    return (p @ vh).transpose(1, 2).reshape(B, T, d)


@pytest.mark.parametrize("T,S,causal,p", [(200, 333, False, 0.25), (192, 192, True, 0.1), (20, 200, False, 0.25), (130, 1000, False, 0.0)])
def test_attention_fwd_bwd_with_dropout_vs_torch_fp32(T, S, causal, p):
    """fp32 forward / dQ / dK / dV over several 64-key tiles, ragged edges, +1.0 and -inf key biases, against autograd through
    the written-out attention with the SAME keep mask (materialised by omr_attn_dropout_mask)."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    B, H, d, seed = 2, 2, 128, 99
    q, k, v = rnd((B, T, d), 1), rnd((B, S, d), 2), rnd((B, S, d), 3)
    bias = torch.zeros(B, S)
    bias[0, S - 40:] = 1.0                     # float padding mask: ADDED (decoder.py:186-188)
    bias[1, S - 25:] = float("-inf")           # bool mask semantics
    g = rnd((B, T, d), 4)
    mask = None
    if p > 0:
        mask = K.attn_dropout_mask(B, H, T, S, p, seed, DEV).cpu().float() / (1 - p)
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attention(qa, ka, va, H, bias, causal, mask)
    ref.backward(g)
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    kw = dict(causal=causal, key_bias=bias.to(DEV), dropout_p=p, seed=seed)
    o, lse = K.attn_fwd(qg, kg, vg, H, **kw)
    torch.testing.assert_close(o.cpu(), ref.detach(), rtol=1e-4, atol=2e-5)
    dq, dk, dv = torch.empty_like(qg), torch.empty_like(kg), torch.empty_like(vg)
    K.attn_bwd(qg, kg, vg, o, g.to(DEV), lse, dq, dk, dv, H, **kw)
    torch.testing.assert_close(dq.cpu(), qa.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dk.cpu(), ka.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dv.cpu(), va.grad, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_lagged_reference_maximum_growing_and_extreme_scores(dtype, tol):
    """The forward keeps a LAGGED reference maximum (it moves only when a tile exceeds it by 2^8).  Scores that grow tile after
    tile force the rescale branch in every tile; rows whose every score is very negative (far below the initial reference)
    and rows with one dominant key must come out like torch's softmax."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    B, H, T, S, d = 1, 1, 64, 640, 64
    k = rnd((B, S, d), 11)
    q = rnd((B, T, d), 12)
    k = k * torch.linspace(0.2, 30.0, S).view(1, S, 1)        # |score| grows with the key index: the maximum keeps moving
    q[:, 40:48] *= 40.0                                       # rows with huge scores of both signs
    bias = torch.zeros(B, S)
    bias[0, :] = -300.0                                       # everything far below the initial reference (2^-300 underflows without re-centring)
    v = rnd((B, S, d), 13)
    ref = _ref_attention(q.to(dtype).float(), k.to(dtype).float(), v.to(dtype).float(), H, bias, False, None)
    o, lse = K.attn_fwd(q.to(DEV, dtype), k.to(DEV, dtype), v.to(DEV, dtype), H, key_bias=bias.to(DEV))
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    err = (o.float().cpu() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err
