"""Development aid: time the bf16 conv weight-gradient kernel at one encoder shape (run under a -DOMR_WGRAD_DEBUG build with
OMR_WGRAD_DBG=bits to ablate phases).  python tools/wgrad_ablate.py [cin cout H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omr_a2s_multimodal_transformer_amd import kernels as K
from tools.gemm_shapes import timeit
ci, co, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 16, 256, 2048)
B = 32
x = torch.randn(B, H, W, ci, device="cuda", dtype=torch.bfloat16)
dy = torch.randn(B, H, W, co, device="cuda", dtype=torch.bfloat16)
dw = torch.zeros(co, 3, 3, ci, device="cuda"); db = torch.zeros(co, device="cuda")
t = timeit(lambda: K.conv3x3_wgrad(x, dy, dw, db=db), 10)
print(f"dbg={os.environ.get('OMR_WGRAD_DBG', '0')} {ci}->{co} {H}x{W}: {t:.0f} us  ({(x.numel() + dy.numel()) * 2 / t / 1e6:.2f} TB/s)")
