"""Phase timing of omr_conv3x3_bwd_fused (debug build only: make with -DOMR_FUSED_DEBUG): python tools/bwd_fused_prof.py [apply]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omr_a2s_multimodal_transformer_amd", "libomr_hip.so"))
B, H, W = 32, 256, 2048
dev = "cuda"
names = ["loop top", "dma wait", "barrier 1", "dma issue", "apply+mfma+store", "-", "loop end"]
for cout, cin in ((32, 32), (32, 16), (16, 16)):
    x = torch.rand((B, H, W, cin), device=dev).sub_(0.4).clamp_min_(0).to(torch.bfloat16)
    g = torch.randn((B, H, W, cout), device=dev).mul_(0.1).to(torch.bfloat16)
    w = (torch.rand((cout, 3, 3, cin), device=dev) - 0.5).to(torch.bfloat16)
    wf = K.conv3x3_weight_flip(w)
    dw = torch.zeros((cout, 3, 3, cin), device=dev); db = torch.zeros(cout, device=dev)
    norm = None
    if len(sys.argv) > 1:
        y = torch.rand((B, H, W, cout), device=dev).sub_(0.4).clamp_min_(0).to(torch.bfloat16)
        mean, rstd = K.instnorm_stats(y)
        ws, slots = K.conv_stat_ws(B, H, W, cout, dev); ws.zero_()
        K.instnorm_reduce_sums(ws, slots, B, cout)
        norm = (y, mean, rstd, ws, slots, True, 1.1)
    K.conv3x3_bwd_fused(g, x, wf, dw, db, True, 1.25, norm=norm)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    so.omr_fused_prof_read(out, 1)
    n = 4
    for _ in range(n):
        K.conv3x3_bwd_fused(g, x, wf, dw, db, True, 1.25, norm=norm)
    torch.cuda.synchronize()
    so.omr_fused_prof_read(out, 1)
    tiles = n * (H // 8) * (W // 32) // 8
    print(f"cout {cout} cin {cin} {'apply' if norm else 'plain'}: cycles per tile (block 0; {tiles} tiles)")
    for half, nm in ((0, "data-gradient wave 0"), (1, "weight-gradient wave 8")):
        vals = [out[half * 8 + k] / tiles for k in range(7)]
        print(f"   {nm:24s} " + "  ".join(f"{names[k]} {vals[k]:7.0f}" for k in range(7)) + f"   sum {sum(vals):7.0f}")
