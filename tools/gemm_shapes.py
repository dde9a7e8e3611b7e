"""Per-shape timing of the GEMMs one C2 training step issues (forward NT, data-gradient NN, weight-gradient TT).
Usage (GPU box): python tools/gemm_shapes.py  -> table on stdout.  Development aid, not part of the product path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K          # noqa: E402
from omr_a2s_multimodal_transformer_amd.functional import split_k_for  # noqa: E402

R, RM = 32 * 512, 32 * 4096
LINEARS = [  # name, rows, in, out, count per step
    ("self.qkv", R, 256, 768, 6), ("self.out", R, 256, 256, 6), ("cross.q", R, 256, 256, 6), ("cross.kv", RM, 256, 512, 0), ("kv.fused", RM, 256, 3072, 1),
    ("cross.out", R, 256, 256, 6), ("ff1", R, 256, 1024, 6), ("ff2", R, 1024, 256, 6), ("head", R, 256, 6997, 1),
    ("pc.128", RM, 128, 128, 9), ("pc.128-256", RM, 128, 256, 1), ("pc.256", RM, 256, 256, 2),
]


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'name':10s} {'rows':>7s} {'in':>5s} {'out':>5s} | {'fwd us':>8s} {'TF/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s} sk")
    for name, rows, kin, nout, cnt in LINEARS:
        x = torch.randn(rows, kin, device=dev, dtype=dt)
        ld = (nout + 7) // 8 * 8
        w = torch.randn(nout, kin, device=dev, dtype=dt)
        gy = torch.randn(rows, ld, device=dev, dtype=dt)[:, :nout]
        bias = torch.zeros(nout, device=dev)
        gw = torch.zeros(nout, kin, device=dev)
        gb = torch.zeros(nout, device=dev)
        ybuf = torch.empty(rows, ld, device=dev, dtype=dt)
        flops = 2.0 * rows * kin * nout
        sk = split_k_for(rows, nout, kin)
        t_f = timeit(lambda: K.gemm(x, w, bias=bias, out=ybuf[:, :nout]))
        t_d = timeit(lambda: K.gemm(gy, w, trans_b=True))
        t_w = timeit(lambda: K.gemm(gy, x, trans_a=True, trans_b=True, out=gw, accumulate=True, split_k=sk, colsum_a=gb))
        tot["fwd"] += t_f * cnt; tot["dgrad"] += t_d * cnt; tot["wgrad"] += t_w * cnt
        # yardstick only (never on the product path): what the vendor library does with the same forward / data-gradient shape
        bb = bias.to(dt)
        t_lf = timeit(lambda: torch.nn.functional.linear(x, w, bb))
        gyc = gy.contiguous()
        t_ld = timeit(lambda: torch.matmul(gyc, w))
        tot.setdefault("lib fwd", 0.0); tot.setdefault("lib dgrad", 0.0)
        tot["lib fwd"] += t_lf * cnt; tot["lib dgrad"] += t_ld * cnt
        print(f"{name:10s} {rows:7d} {kin:5d} {nout:5d} | {t_f:8.1f} {flops / t_f / 1e6:6.0f} | {t_d:8.1f} {flops / t_d / 1e6:6.0f} | {t_w:8.1f} {flops / t_w / 1e6:6.0f} {sk} | lib fwd {t_lf:7.1f} dgrad {t_ld:7.1f}")
    print("per-step totals (ms):", {k: round(v / 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
