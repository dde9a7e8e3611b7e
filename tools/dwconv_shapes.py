"""Timing of the depthwise 3x3 kernels at the DSC-block shapes of C2 (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402
from tools.gemm_shapes import timeit  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 32, 16, 256
for C in (128, 256):
    x = torch.randn(B, H, W, C, device=dev, dtype=torch.bfloat16)
    g = torch.randn(B, H, W, C, device=dev, dtype=torch.bfloat16)
    w = torch.randn(C, 9, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(C, device=dev)
    stats = K.instnorm_stats(x)
    dw = torch.zeros(C, 9, device=dev); db = torch.zeros(C, device=dev)
    mb = x.numel() * 2 / 1e6
    r = {
        "fwd": timeit(lambda: K.dwconv3x3(x, w, bias)),
        "fwd+norm": timeit(lambda: K.dwconv3x3(x, w, bias, in_stats=stats)),
        "dgrad": timeit(lambda: K.dwconv3x3(g, w, None, flip=True)),
        "dgrad+mask": timeit(lambda: K.dwconv3x3(g, w, None, flip=True, out_mask=x, mask_scale=1.0)),
        "wgrad": timeit(lambda: K.dwconv3x3_wgrad(x, g, dw, db)),
        "wgrad+norm": timeit(lambda: K.dwconv3x3_wgrad(x, g, dw, db, in_stats=stats)),
        "stats": timeit(lambda: K.instnorm_stats(x)),
    }
    print(f"C={C} tensor={mb:.0f} MB  " + "  ".join(f"{k}={v:.0f}us" for k, v in r.items()), flush=True)
