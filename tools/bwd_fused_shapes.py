"""Time omr_conv3x3_bwd_fused against the two kernels it replaces at the C2 shapes (B = 32, 256 x 2048, bf16):
python tools/bwd_fused_shapes.py [apply]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402

B, H, W = 32, 256, 2048
dev = "cuda"


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cout, cin in ((32, 32), (32, 16), (16, 16)):
    x = torch.rand((B, H, W, cin), device=dev).sub_(0.4).clamp_min_(0).to(torch.bfloat16)
    g = torch.randn((B, H, W, cout), device=dev).mul_(0.1).to(torch.bfloat16)
    w = (torch.rand((cout, 3, 3, cin), device=dev) - 0.5).to(torch.bfloat16)
    wf = K.conv3x3_weight_flip(w)
    dw = torch.zeros((cout, 3, 3, cin), device=dev)
    db = torch.zeros(cout, device=dev)
    t_f = timed(lambda: K.conv3x3_bwd_fused(g, x, wf, dw, db, True, 1.25))
    t_d = timed(lambda: K.conv3x3(g, wf, None, out_hw=(H, W), out_mask=x, mask_scale=1.25))
    t_w = timed(lambda: K.conv3x3_wgrad(x, g, dw, db=db))
    gb = B * H * W * (cout + 2 * cin) * 2 / 1e9
    print(f"cout {cout} cin {cin}: fused {t_f:7.1f} us ({gb / t_f * 1e3:5.2f} TB/s on {gb:.2f} GB)   dgrad {t_d:7.1f} + wgrad {t_w:7.1f} = {t_d + t_w:7.1f} us", flush=True)
    if len(sys.argv) > 1:
        y = torch.rand((B, H, W, cout), device=dev).sub_(0.4).clamp_min_(0).to(torch.bfloat16)
        mean, rstd = K.instnorm_stats(y)
        ws, slots = K.conv_stat_ws(B, H, W, cout, dev)
        ws.zero_()
        K.instnorm_reduce_sums(ws, slots, B, cout)
        t_a = timed(lambda: K.conv3x3_bwd_fused(g, x, wf, dw, db, True, 1.25, norm=(y, mean, rstd, ws, slots, True, 1.1)))
        t_p = timed(lambda: K.instnorm_bwd_apply(g, y, mean, rstd, ws, slots, True, 1.1))
        print(f"    apply-on-load fused {t_a:7.1f} us   vs apply pass {t_p:7.1f} + dgrad + wgrad = {t_p + t_d + t_w:7.1f} us", flush=True)
    del x, g, dw

# the normalise-on-load form (16 -> 16): against omr_conv3x3_fwd stat_mode 2 + omr_conv3x3_wgrad with in_stats
x = torch.rand((B, H, W, 16), device=dev).sub_(0.4).clamp_min_(0).to(torch.bfloat16)
g = torch.randn((B, H, W, 16), device=dev).mul_(0.1).to(torch.bfloat16)
w = (torch.rand((16, 3, 3, 16), device=dev) - 0.5).to(torch.bfloat16)
wf = K.conv3x3_weight_flip(w)
dw = torch.zeros((16, 3, 3, 16), device=dev)
db = torch.zeros(16, device=dev)
mean, rstd = K.instnorm_stats(x)
ws, slots = K.conv_stat_ws(B, H, W, 16, dev)
t_f = timed(lambda: K.conv3x3_bwd_fused(g, x, wf, dw, db, False, 1.0, xnorm=(mean, rstd, ws, slots)))
t_d = timed(lambda: K.conv3x3(g, wf, None, out_hw=(H, W), stat_mode=2, stat_ws=ws, stat_slots=slots, stat_x=x, stat_stats=(mean, rstd)))
t_w = timed(lambda: K.conv3x3_wgrad(x, g, dw, in_stats=(mean, rstd), db=db))
print(f"normalise-on-load 16 -> 16: fused {t_f:7.1f} us   dgrad + sums {t_d:7.1f} + wgrad {t_w:7.1f} = {t_d + t_w:7.1f} us", flush=True)

# the strided normalise-on-load form (32 -> 32, stride 2): against the zero-dilated data gradient + sums and the strided weight gradient
x = torch.rand((B, H, W, 32), device=dev).sub_(0.4).clamp_min_(0).to(torch.bfloat16)
g = torch.randn((B, H // 2, W // 2, 32), device=dev).mul_(0.1).to(torch.bfloat16)
w = (torch.rand((32, 3, 3, 32), device=dev) - 0.5).to(torch.bfloat16)
wf = K.conv3x3_weight_flip(w)
dw = torch.zeros((32, 3, 3, 32), device=dev)
db = torch.zeros(32, device=dev)
mean, rstd = K.instnorm_stats(x)
ws, slots = K.conv_stat_ws(B, H, W, 32, dev)
t_f = timed(lambda: K.conv3x3_bwd_fused_s2(g, x, wf, dw, db, mean, rstd, ws, slots))
t_d = timed(lambda: K.conv3x3(g, wf, None, stride=(1, 1), dil=(2, 2), out_hw=(H, W), stat_mode=2, stat_ws=ws, stat_slots=slots, stat_x=x, stat_stats=(mean, rstd)))
t_w = timed(lambda: K.conv3x3_wgrad(x, g, dw, stride=(2, 2), in_stats=(mean, rstd), db=db))
print(f"strided normalise-on-load 32 -> 32: fused {t_f:7.1f} us   dgrad + sums {t_d:7.1f} + wgrad {t_w:7.1f} = {t_d + t_w:7.1f} us", flush=True)
