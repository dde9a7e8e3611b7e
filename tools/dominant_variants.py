"""Time the instantiations of the dominant conv kernel (32->32 ch, B=32, 256x2048, bf16): plain (EPI=0), statistics only
(EPI=1), statistics + elementwise / channel MixDropout, and the masked data gradient (EPI=0 with an epilogue mask)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omr_a2s_multimodal_transformer_amd import kernels as K

B, H, W, C = 32, 256, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.rand((B, H, W, C), device="cuda").to(torch.bfloat16)
w = (torch.rand((C, 3, 3, C), device="cuda") - 0.5).to(torch.bfloat16)
bias = torch.zeros(C, device="cuda")
ws, slots = K.conv_stat_ws(B, H, W, C, x.device)
st = K.instnorm_stats(x)
variants = {
    "plain (EPI=0)": lambda: K.conv3x3(x, w, bias, relu=True),
    "stats (EPI=1)": lambda: K.conv3x3(x, w, bias, relu=True, stat_mode=1, stat_ws=ws, stat_slots=slots),
    "stats + elementwise dropout": lambda: K.conv3x3(x, w, bias, relu=True, drop=(0.5, 7, False), stat_mode=1, stat_ws=ws, stat_slots=slots),
    "stats + channel dropout": lambda: K.conv3x3(x, w, bias, relu=True, drop=(0.25, 7, True), stat_mode=1, stat_ws=ws, stat_slots=slots),
    "masked dgrad (EPI=0 + mask)": lambda: K.conv3x3(x, w, None, out_mask=x, mask_scale=2.0),
    "dgrad + IN-backward sums (EPI=2)": lambda: K.conv3x3(x, w, None, stat_mode=2, stat_ws=ws, stat_slots=slots, stat_x=x, stat_stats=st),
}
gb = B * H * W * 2 * C * 2 / 1e9
for name, fn in variants.items():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:32s} {ms * 1e3:7.1f} us   {gb / ms:6.2f} TB/s algorithmic")
