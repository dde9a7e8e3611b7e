"""Where does a short-K forward GEMM spend its time?  Times the fused cross K|V shape (131072 x 3072) at several K and with /
without bias: the K = 64 point is prologue + epilogue, the slope is the cost of one BK = 64 iteration.  Development aid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402
from tools.gemm_shapes import timeit                         # noqa: E402

dev = torch.device("cuda:0")
for rows, nout in ((131072, 3072), (16384, 7000), (16384, 1024)):
    for kin in (64, 128, 256, 512, 1024):
        x = torch.randn(rows, kin, device=dev, dtype=torch.bfloat16)
        w = torch.randn(nout, kin, device=dev, dtype=torch.bfloat16)
        b = torch.zeros(nout, device=dev)
        y = torch.empty(rows, (nout + 7) // 8 * 8, device=dev, dtype=torch.bfloat16)[:, :nout]
        t0 = timeit(lambda: K.gemm(x, w, bias=None, out=y), 10)
        t1 = timeit(lambda: K.gemm(x, w, bias=b, out=y), 10)
        print(f"rows {rows} N {nout} K {kin:5d}: no-bias {t0:7.1f} us  bias {t1:7.1f} us   write-only floor {rows * nout * 2 / 5.4e6:6.1f} us", flush=True)
