import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
from tests.test_attention_r3_gpu import rnd
DEV = "cuda:0"
B, H, T, S, d, p, seed = 1, 1, 20, 128, 64, 0.25, 99
q, k = rnd((B, T, d), 1), rnd((B, S, d), 2)
mask = K.attn_dropout_mask(B, H, T, S, p, seed, DEV).cpu().float()[0, 0]
P = torch.softmax(q[0] @ k[0].t() / math.sqrt(d), dim=-1)
for off in (0, 64):
    v = torch.zeros(B, S, d)
    for j in range(64):
        v[0, off + j, j] = 1.0
    o, lse = K.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), H, dropout_p=p, seed=seed)
    got = o[0].cpu()                      # [T, 64] = dropped P[:, off:off+64] / (1-p)
    want = P[:, off:off + 64] * mask[:, off:off + 64] / (1 - p)
    bad = ~torch.isclose(got, want, rtol=1e-4, atol=1e-6)
    print("keys", off, "nan", int(torch.isnan(got).sum()), "bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()[:6].tolist()
        print("  first bad (q,key):", idx, "got", [float(got[i, j]) for i, j in idx], "want", [float(want[i, j]) for i, j in idx], "undropped", [float(P[i, off + j] / (1 - p)) for i, j in idx])
print("lse err", float((lse[0, 0].cpu() - torch.logsumexp(q[0] @ k[0].t() / math.sqrt(d), -1)).abs().max()))
