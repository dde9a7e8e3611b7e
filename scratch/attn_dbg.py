import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
from tests.test_attention_r3_gpu import _ref_attention, rnd
DEV = "cuda:0"
for (T, S, causal, p, usebias) in [(20, 200, False, 0.0, False), (20, 200, False, 0.25, False), (20, 64, False, 0.25, False), (20, 128, False, 0.25, False), (32, 256, False, 0.25, False), (20, 200, False, 0.25, True)]:
    B, H, d, seed = 2, 2, 128, 99
    q, k, v = rnd((B, T, d), 1), rnd((B, S, d), 2), rnd((B, S, d), 3)
    bias = torch.zeros(B, S)
    if usebias:
        bias[0, S - 40:] = 1.0
        bias[1, S - 25:] = float("-inf")
    g = rnd((B, T, d), 4)
    mask = K.attn_dropout_mask(B, H, T, S, p, seed, DEV).cpu().float() / (1 - p) if p > 0 else None
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attention(qa, ka, va, H, bias, causal, mask)
    ref.backward(g)
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    kw = dict(causal=causal, key_bias=bias.to(DEV), dropout_p=p, seed=seed)
    o, lse = K.attn_fwd(qg, kg, vg, H, **kw)
    dq, dk, dv = torch.empty_like(qg), torch.empty_like(kg), torch.empty_like(vg)
    K.attn_bwd(qg, kg, vg, o, g.to(DEV), lse, dq, dk, dv, H, **kw)
    e = lambda a, b: ((a.cpu() - b).abs().max() / b.abs().max()).item()
    print(f"T={T} S={S} causal={causal} p={p} bias={usebias}: o {e(o, ref.detach()):.2e} dq {e(dq, qa.grad):.2e} dk {e(dk, ka.grad):.2e} dv {e(dv, va.grad):.2e}", flush=True)
    if e(o, ref.detach()) > 1e-3:
        err = (o.cpu() - ref.detach()).abs().view(B, T, H, d // H).amax(-1)
        print("   o err per (b,h):", err.amax(1).tolist(), "per q b0h0:", [round(x, 3) for x in err[0, :, 0].tolist()])
    if e(dk, ka.grad) > 1e-3:
        err = (dk.cpu() - ka.grad).abs().view(B, S, H, d // H).amax(-1)   # [B,S,H]
        print("   dk err per (b, h) max:", err.amax(1).tolist(), " worst keys b0h0:", err[0, :, 0].topk(5).indices.tolist(), " ratio sample:", (dk.cpu()[0, 5, :4] / ka.grad[0, 5, :4]).tolist())
