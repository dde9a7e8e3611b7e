"""Launch only the benchmark's dominant kernel (conv3x3_mfma on conv_blocks.1.conv2: 32->32 channels, B=32, 256x2048, bf16)
a few times -- the target of the rocprofv3 passes whose summaries live in profiles/ (kernel-trace --stats, --pmc FETCH_SIZE,
--pmc WRITE_SIZE as separate passes, MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402

B, H, W, C = 32, 256, 2048, 32
x = torch.rand((B, H, W, C), device="cuda").to(torch.bfloat16)
w = (torch.rand((C, 3, 3, C), device="cuda") - 0.5).to(torch.bfloat16)
bias = torch.zeros(C, device="cuda")
for _ in range(6):
    y = K.conv3x3(x, w, bias, relu=True)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", B * H * W * 2 * C * 2)
