"""Per-layer timing of the encoder's 3x3 convolutions at C2 (forward, data gradient, weight gradient) against the
HBM time of their algorithmic bytes.  Development aid (GPU box): python tools/conv_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402
from tools.gemm_shapes import timeit                         # noqa: E402

B = 32
LAYERS = [  # name, H, W, cin, cout, stride
    ("cb0.c1", 256, 2048, 1, 16, (1, 1)),
    ("cb0.c2", 256, 2048, 16, 16, (1, 1)), ("cb0.c3", 256, 2048, 16, 16, (1, 1)),
    ("cb1.c1", 256, 2048, 16, 32, (1, 1)), ("cb1.c2", 256, 2048, 32, 32, (1, 1)), ("cb1.c3", 256, 2048, 32, 32, (2, 2)),
    ("cb2.c1", 128, 1024, 32, 64, (1, 1)), ("cb2.c2", 128, 1024, 64, 64, (1, 1)), ("cb2.c3", 128, 1024, 64, 64, (2, 2)),
    ("cb3.c1", 64, 512, 64, 128, (1, 1)), ("cb3.c2", 64, 512, 128, 128, (1, 1)), ("cb3.c3", 64, 512, 128, 128, (2, 2)),
    ("cb4.c1", 32, 256, 128, 128, (1, 1)), ("cb4.c2", 32, 256, 128, 128, (1, 1)), ("cb4.c3", 32, 256, 128, 128, (2, 1)),
]


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    tot = [0.0, 0.0, 0.0, 0.0]
    print(f"{'layer':8s} {'MB':>6s} {'ideal us@4TB/s':>14s} | {'fwd':>7s} {'dgrad':>7s} {'wgrad':>7s}")
    for name, H, W, ci, co, st in LAYERS:
        x = torch.randn(B, H, W, ci, device=dev, dtype=dt)
        w = torch.randn(co, 3, 3, ci, device=dev, dtype=dt) * 0.05
        bias = torch.zeros(co, device=dev)
        Ho, Wo = K.conv_out_hw(H, W, st)
        dy = torch.randn(B, Ho, Wo, co, device=dev, dtype=dt)
        wf = K.conv3x3_weight_flip(w) if ci > 1 else None
        dw = torch.zeros(co, 3, 3, ci, device=dev)
        db = torch.zeros(co, device=dev)
        mb = (x.numel() + dy.numel()) * 2 / 1e6
        t_f = timeit(lambda: K.conv3x3(x, w, bias, stride=st, relu=True), 10)
        t_d = timeit(lambda: K.conv3x3(dy, wf, None, stride=(1, 1), dil=st, out_hw=(H, W), out_mask=x, mask_scale=1.0), 10) if ci > 1 else 0.0   # the first layer has no data gradient
        t_w = timeit(lambda: K.conv3x3_wgrad(x, dy, dw, stride=st, db=db), 10)
        t_n = 0.0
        if name.endswith(".c3"):                      # conv3 of a block reads its input through the fused InstanceNorm-apply
            st_in = K.instnorm_stats(x)
            t_n = timeit(lambda: K.conv3x3(x, w, bias, stride=st, relu=True, in_stats=st_in), 10)
        ideal = mb / 4.0
        gf = 2.0 * B * Ho * Wo * co * ci * 9 / 1e9
        floor = max(mb / 6.3, gf / 1.2)            # us: HBM at the achievable 6.3 TB/s | MFMA at the ~1.2 PF/s a tuned bf16 loop holds
        for i, t in enumerate((ideal, t_f, t_d, t_w)):
            tot[i] += t
        print(f"{name:8s} {mb:6.0f} {ideal:14.0f} | {t_f:7.0f} {t_d:7.0f} {t_w:7.0f}" + f"  | {gf:5.0f} GF floor {floor:4.0f} us  fwd x{t_f / floor:4.2f} dgrad x{t_d / floor:4.2f}" + (f"   fwd with IN-apply {t_n:5.0f}" if t_n else ""), flush=True)
    print("totals (ms): ideal %.2f fwd %.2f dgrad %.2f wgrad %.2f" % tuple(t / 1e3 for t in tot))


if __name__ == "__main__":
    main()
