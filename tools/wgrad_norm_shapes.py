"""Weight gradient of the normalise-on-load convs (conv3 of every ConvBlock) and of the plain ones at the C2 shapes: time and
bytes moved per layer.  Development aid (GPU box): python tools/wgrad_norm_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402
from tools.gemm_shapes import timeit                         # noqa: E402
from tools.conv_shapes import LAYERS, B                      # noqa: E402

tot = 0.0
for name, H, W, ci, co, st in LAYERS:
    x = torch.randn(B, H, W, ci, device="cuda", dtype=torch.bfloat16)
    Ho, Wo = K.conv_out_hw(H, W, st)
    dy = torch.randn(B, Ho, Wo, co, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(co, 3, 3, ci, device="cuda"); db = torch.zeros(co, device="cuda")
    stats = K.instnorm_stats(x) if name.endswith(".c3") else None
    t = timeit(lambda: K.conv3x3_wgrad(x, dy, dw, stride=st, in_stats=stats, db=db), 10)
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    gf = 18.0 * ci * co * B * Ho * Wo / 1e9
    tot += t
    print(f"{name:7s} {ci:3d}->{co:3d} s{st} {'norm' if stats else '    '}: {t:6.0f} us  {gb / t * 1e3:5.2f} TB/s  {gf / t / 1e3:6.0f} TF/s", flush=True)
print(f"total {tot / 1e3:.2f} ms")
