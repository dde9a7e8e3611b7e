import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
B, H, W = 32, 256, 2048
x = torch.rand((B, H, W, 1), device="cuda").to(torch.bfloat16)
dy = torch.randn((B, H, W, 16), device="cuda").to(torch.bfloat16)
dw = torch.zeros((16, 3, 3, 1), device="cuda"); db = torch.zeros(16, device="cuda")
for _ in range(3): K.conv3x3_wgrad(x, dy, dw, db=db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): K.conv3x3_wgrad(x, dy, dw, db=db)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print(f"first-layer weight gradient: {t:.1f} us ({B*H*W*17*2/t/1e6:.2f} TB/s)")
