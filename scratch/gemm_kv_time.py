import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
from tools.gemm_shapes import timeit
dev = "cuda:0"
M, N, Kd = 131072, 3072, 256
a = torch.randn(M, Kd, device=dev, dtype=torch.bfloat16)
w = torch.randn(N // 2 * 3, Kd, device=dev, dtype=torch.bfloat16) * 0.05
b = torch.randn(N // 2 * 3, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: K.gemm_row_groups(a, w, out, M, N, Kd, bias=b, group=(512, 768, 256, 1)), 10)
print(f"all-layer K|V projection {M}x{N}x{Kd}: {t:.1f} us  ({2.0*M*N*Kd/t/1e6:.0f} TF/s, output {M*N*2/t/1e3:.0f} GB/s)")
