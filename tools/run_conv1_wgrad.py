"""Launch the first-layer weight gradient (1 -> 16 channels, B=32, 256x2048, bf16) a few times: target of rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402
B, H, W = 32, 256, 2048
x = torch.rand((B, H, W, 1), device="cuda").to(torch.bfloat16)
dy = torch.randn((B, H, W, 16), device="cuda").to(torch.bfloat16)
dw = torch.zeros((16, 3, 3, 1), device="cuda"); db = torch.zeros(16, device="cuda")
for _ in range(5):
    K.conv3x3_wgrad(x, dy, dw, db=db)
torch.cuda.synchronize()
