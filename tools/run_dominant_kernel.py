"""Launch only the benchmark's dominant kernel -- conv3x3_mfma on conv_blocks.1.conv2 (32->32 channels, B=32, 256x2048, bf16)
in the instantiation the training step runs: EPI = 1, bias + ReLU + fused InstanceNorm statistics of the output (the
MixDropout of the block lands on this conv in one step out of three; `drop` below adds it) -- a few times: the target of
the rocprofv3 passes whose summaries live in profiles/ (kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE as
separate passes, MI355X_MICROARCH.md HBM section).   python tools/run_dominant_kernel.py [drop]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402

B, H, W, C = 32, 256, 2048, 32
drop = (0.5, 1234, False) if len(sys.argv) > 1 and sys.argv[1] == "drop" else None
x = torch.rand((B, H, W, C), device="cuda").to(torch.bfloat16)
w = (torch.rand((C, 3, 3, C), device="cuda") - 0.5).to(torch.bfloat16)
bias = torch.zeros(C, device="cuda")
ws, slots = K.conv_stat_ws(B, H, W, C, x.device)
for _ in range(6):
    y = K.conv3x3(x, w, bias, relu=True, drop=drop, stat_mode=1, stat_ws=ws, stat_slots=slots)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", B * H * W * 2 * C * 2)
