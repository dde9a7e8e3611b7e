"""Development aid: add + LayerNorm forward / backward kernels at the C2 decoder shape (16 384 rows x 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omr_a2s_multimodal_transformer_amd import kernels as K
from tools.gemm_shapes import timeit
M, d = 16384, 256
x = torch.randn(M, d, device="cuda").to(torch.bfloat16); res = torch.randn_like(x); dy = torch.randn_like(x)
g, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
out, mean, rstd = K.add_layernorm_fwd(x, res, g, b)
dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
print("fwd %.1f us" % timeit(lambda: K.add_layernorm_fwd(x, res, g, b), 30))
print("bwd %.1f us" % timeit(lambda: K.add_layernorm_bwd(dy, x, res, g, mean, rstd, dg, db), 30))
print("bwd+drop %.1f us" % timeit(lambda: K.add_layernorm_bwd(dy, x, res, g, mean, rstd, dg, db, drop_p=0.1, drop_seed=5), 30))
