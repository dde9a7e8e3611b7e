"""Cross-attention kernels at the C2 shape (B=32, 4 heads x 64, T=512 queries, S=4096 keys), with and without dropout:
how much of the time is the mask generation.  Development aid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402
from tools.gemm_shapes import timeit                         # noqa: E402

dev = torch.device("cuda:0")
B, H, T, S, d = 32, 4, 512, 4096, 256
q = torch.randn(B, T, d, device=dev, dtype=torch.bfloat16)
kv = torch.randn(B, S, 2 * d, device=dev, dtype=torch.bfloat16)
k, v = kv[..., :d], kv[..., d:]
kb = torch.zeros(B, S, device=dev)
flops = 4.0 * B * H * T * S * (d // H)
t_g = timeit(lambda: K.attn_dropout_words(B, H, T, S, 0.1, 7, dev), 10)
print(f"dropout words (once per layer and step): {t_g:6.1f} us", flush=True)
for p in (0.0, 0.1):
    words = K.attn_dropout_words(B, H, T, S, p, 7, dev) if p > 0 else None
    for bias in (None, kb):
        o, lse = K.attn_fwd(q, k, v, H, key_bias=bias, dropout_p=p, seed=7, drop_words=words)
        do = torch.randn_like(o)
        dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        dk, dv = dkv[..., :d], dkv[..., d:]
        t_f = timeit(lambda: K.attn_fwd(q, k, v, H, key_bias=bias, dropout_p=p, seed=7, drop_words=words), 10)
        t_b = timeit(lambda: K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, H, key_bias=bias, dropout_p=p, seed=7, drop_words=words), 10)
        print(f"p={p} bias={'y' if bias is not None else 'n'}: fwd {t_f:6.1f} us ({flops / t_f / 1e6:5.0f} TF/s)   bwd (dq+dkv) {t_b:6.1f} us ({2.5 * flops / t_b / 1e6:5.0f} TF/s)", flush=True)
