import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
DEV = "cuda:0"
B, H, T, S, p, seed = 1, 1, 64, 128, 0.25, 99
mask = K.attn_dropout_mask(B, H, T, S, p, seed, DEV).cpu().numpy()[0, 0]          # [T,S]
words = K.attn_dropout_words(B, H, T, S, p, seed, DEV).cpu().numpy().view(np.uint32)   # dwords
nqb, nkt = (T + 31) // 32, (S + 63) // 64
bad = 0
for key in range(S):
    kt, mbk, ko = key >> 6, (key >> 5) & 1, key & 31
    r, half = (ko & 3) + 4 * (ko >> 3), (ko >> 2) & 1
    for qb32 in range(nqb):
        dw = int(words[2 * (((0 * nqb + qb32) * nkt + kt) * 32 + mbk * 16 + r) + half])
        for qo in range(32):
            if ((dw >> qo) & 1) != mask[qb32 * 32 + qo, key]:
                bad += 1
print("emulated dkv indexing mismatches:", bad)
# now the kernel: V = identity trick to read the dropped P from dV?  use dO = one-hot to read pd^T: dV[key][d] = sum_q pd[q][key] dO[q][d]
d = 64
q = torch.zeros(B, T, d, device=DEV); k = torch.zeros(B, S, d, device=DEV); v = torch.zeros(B, S, d, device=DEV)
o, lse = K.attn_fwd(q, k, v, 1, dropout_p=p, seed=seed)      # uniform attention: P = 1/S
for qsel in (0, 5, 37, 63):
    g = torch.zeros(B, T, d, device=DEV); g[0, qsel, 0] = 1.0
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, g, lse, dq, dk, dv, 1, dropout_p=p, seed=seed)
    got = (dv[0, :, 0].cpu().numpy() > 0).astype(np.uint8)        # keep bits of query qsel over the keys, as the dkv kernel saw them
    want = mask[qsel]
    print("q", qsel, "mismatch", int((got != want).sum()), "first got", got[:40].tolist(), "want", want[:40].tolist())
print("---- p = 0 sanity, and fp32 vs bf16")
for dt in (torch.float32, torch.bfloat16):
    for pp in (0.0, p):
        qq, kk, vv = q.to(dt), k.to(dt), v.to(dt)
        o, lse = K.attn_fwd(qq, kk, vv, 1, dropout_p=pp, seed=seed)
        res = []
        for qsel in (0, 1, 4, 5, 8, 12, 37):
            g = torch.zeros(B, T, d, device=DEV, dtype=dt); g[0, qsel, 0] = 1.0
            dq, dk, dv = torch.empty_like(qq), torch.empty_like(kk), torch.empty_like(vv)
            K.attn_bwd(qq, kk, vv, o, g, lse, dq, dk, dv, 1, dropout_p=pp, seed=seed)
            got = (dv[0, :, 0].float().cpu().numpy() > 0).astype(np.uint8)
            res.append((qsel, int(got.sum()), int((got != mask[qsel]).sum())))
        print(dt, pp, "(q, kept count seen, mismatches vs mask):", res)
