"""Seeded shape fuzz for the kernels whose indexing is the most intricate (asynchronous LDS rings with hand-counted DMA
waits, image-border handling, row walkers, row-group GEMM views): many irregular shapes against torch CPU references.
Each case is small; the point is the spread of tile remainders, borders and channel tails."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_kernels_gpu import K, check, dev, nhwc, q, rnd  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


import os

_SEED_SHIFT = int(os.environ.get("OMR_FUZZ_SEED_SHIFT", "0"))      # OMR_FUZZ_SEED_SHIFT=k draws a different set of shapes


def _cases(seed, n, gen):
    rng = random.Random(seed + 1000003 * _SEED_SHIFT)
    return [gen(rng) for _ in range(n)]


CONV_BWD = _cases(1, 14, lambda r: (r.choice([16, 32, 64, 128]), r.choice([16, 32, 64, 128]), r.choice([(1, 1), (1, 1), (2, 2), (2, 1)]),
                                    r.randint(3, 41), r.randint(5, 139), r.choice([False, True]), r.randint(1, 3)))


@pytest.mark.parametrize("cin,cout,stride,H,W,norm,B", CONV_BWD)
def test_fuzz_conv_weight_gradient_bf16(cin, cout, stride, H, W, norm, B):
    """wgrad_dma: tile remainders in both directions, 1-3 images, all stride / normalisation variants."""
    x = q(F.relu(rnd((B, cin, H, W), 11) + 0.2), BF).requires_grad_(True)
    w = q(rnd((cout, cin, 3, 3), 12) / math.sqrt(cin * 9) * 2, BF).requires_grad_(True)
    bias = rnd((cout,), 13).requires_grad_(True)
    xin = R.instance_norm(x) if norm else x
    pre = F.conv2d(xin, w, bias, stride=stride, padding=1)
    g = q(rnd(tuple(pre.shape), 14), BF)
    pre.backward(g)
    k = K()
    xg = nhwc(x.detach()).to(dev(), BF)
    stats = k.instnorm_stats(xg) if norm else None
    dw, db = torch.zeros((cout, 3, 3, cin), device=dev()), torch.zeros(cout, device=dev())
    k.conv3x3_wgrad(xg, nhwc(g).to(dev(), BF), dw, stride=stride, in_stats=stats, db=db)
    n_red = B * pre.shape[2] * pre.shape[3]
    check(dw, w.grad.permute(0, 2, 3, 1), BF, scale=math.sqrt(n_red) * (2 if norm else 0.5), what="wgrad")
    check(db, bias.grad, BF, scale=math.sqrt(n_red) / 2, what="bias grad")


CONV_FWD = _cases(2, 10, lambda r: (r.choice([16, 32, 64]), r.choice([16, 32, 64, 128]), r.choice([(1, 1), (2, 2), (2, 1)]), r.randint(3, 37),
                                    r.randint(5, 101), r.choice([BF, F32])))


@pytest.mark.parametrize("cin,cout,stride,H,W,dtype", CONV_FWD)
def test_fuzz_conv_forward_and_masked_data_gradient(cin, cout, stride, H, W, dtype):
    B = 2
    x = q(F.relu(rnd((B, cin, H, W), 21)), dtype).requires_grad_(True)
    w = q(rnd((cout, cin, 3, 3), 22) / math.sqrt(cin * 9) * 2, dtype).requires_grad_(True)
    bias = rnd((cout,), 23)
    pre = F.conv2d(x, w, bias, stride=stride, padding=1)
    k = K()
    xg = nhwc(x.detach()).to(dev(), dtype)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().to(dev(), dtype)
    check(k.conv3x3(xg, wg, bias.to(dev()), stride=stride, relu=True), nhwc(F.relu(pre)), dtype, what="conv fwd")
    g = q(rnd(tuple(pre.shape), 24), dtype)
    pre.backward(g)
    dx = k.conv3x3(nhwc(g).to(dev(), dtype), k.conv3x3_weight_flip(wg), None, stride=(1, 1), dil=stride, out_hw=(H, W), out_mask=xg, mask_scale=2.0)
    check(dx, nhwc(x.grad * (x.detach() > 0) * 2.0), dtype, scale=4, what="masked dgrad")


DW = _cases(3, 8, lambda r: (r.choice([64, 128, 256]), r.randint(2, 23), r.randint(3, 70), r.choice([False, True]), r.choice([BF, F32])))


@pytest.mark.parametrize("C,H,W,norm,dtype", DW)
def test_fuzz_depthwise_walkers(C, H, W, norm, dtype):
    B = 2
    x = q(rnd((B, C, H, W), 31), dtype).requires_grad_(True)
    w = q(rnd((C, 1, 3, 3), 32) / 3, dtype).requires_grad_(True)
    bias = rnd((C,), 33).requires_grad_(True)
    xin = R.instance_norm(x) if norm else x
    y = F.conv2d(xin, w, bias, padding=1, groups=C)
    g = q(rnd(tuple(y.shape), 34), dtype)
    y.backward(g)
    k = K()
    xg = nhwc(x.detach()).to(dev(), dtype)
    wg = w.detach().reshape(C, 9).to(dev(), dtype)
    stats = k.instnorm_stats(xg) if norm else None
    check(k.dwconv3x3(xg, wg, bias.detach().to(dev()), in_stats=stats), nhwc(y), dtype, scale=2, what="dw fwd")
    dw, db = torch.zeros((C, 9), device=dev()), torch.zeros(C, device=dev())
    k.dwconv3x3_wgrad(xg, nhwc(g).to(dev(), dtype), dw, db, in_stats=stats)
    check(dw, w.grad.reshape(C, 9), dtype, scale=math.sqrt(B * H * W) * (2 if norm else 1), what="dw wgrad")
    check(db, bias.grad, dtype, scale=math.sqrt(B * H * W), what="dw bias grad")


@pytest.mark.parametrize("H,W,B", _cases(4, 6, lambda r: (r.randint(1, 70), r.randint(1, 300), r.randint(1, 3))))
def test_fuzz_first_layer(H, W, B):
    x = q(rnd((B, 1, H, W), 41, 0, 1), BF).requires_grad_(True)
    w = q(rnd((16, 1, 3, 3), 42) / 1.5, BF).requires_grad_(True)
    bias = rnd((16,), 43).requires_grad_(True)
    pre = F.conv2d(x, w, bias, padding=1)
    k = K()
    xg = nhwc(x.detach()).to(dev(), BF)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().to(dev(), BF)
    check(k.conv3x3(xg, wg, bias.detach().to(dev()), relu=True), nhwc(F.relu(pre)), BF, what="conv1 fwd")
    g = q(rnd(tuple(pre.shape), 44), BF)
    pre.backward(g)
    dw, db = torch.zeros((16, 3, 3, 1), device=dev()), torch.zeros(16, device=dev())
    k.conv3x3_wgrad(xg, nhwc(g).to(dev(), BF), dw, db=db)
    check(dw, w.grad.permute(0, 2, 3, 1), BF, scale=math.sqrt(B * H * W) / 2, what="conv1 wgrad")
    check(db, bias.grad, BF, scale=math.sqrt(B * H * W) / 2, what="conv1 bias grad")


@pytest.mark.parametrize("L,d,rows", _cases(5, 6, lambda r: (r.randint(2, 7), r.choice([64, 128, 256]), r.randint(100, 900))))
def test_fuzz_row_group_gemm(L, d, rows):
    """The three uses of the row-group view (FusedCrossKVFn): forward over K|V rows of L packed in_proj matrices, weight
    gradient into them, data gradient through them -- against explicit per-layer slices."""
    k = K()
    w_all = q(rnd((L * 3 * d, d), 51) / math.sqrt(d), BF)             # [Wq;Wk;Wv] x L, back to back
    b_all = rnd((L * 3 * d,), 52)
    x = q(rnd((rows, d), 53), BF)
    grp = (2 * d, 3 * d, d, 0)
    kv_rows = torch.cat([torch.arange(l * 3 * d + d, (l + 1) * 3 * d) for l in range(L)])
    wg, bg, xg = w_all.to(dev(), BF), b_all.to(dev()), x.to(dev(), BF)
    out = torch.empty((rows, L * 2 * d), dtype=BF, device=dev())
    k.gemm_row_groups(xg, wg, out, rows, L * 2 * d, d, bias=bg, group=grp[:3] + (1,))
    check(out, x @ w_all[kv_rows].t() + b_all[kv_rows], BF, scale=2, what="grouped fwd")
    g = q(rnd((rows, L * 2 * d), 54), BF)
    gg = g.to(dev(), BF)
    gw = torch.zeros((L * 3 * d, d), device=dev())
    gb = torch.zeros(L * 3 * d, device=dev())
    k.gemm_row_groups(gg, xg, gw, L * 2 * d, d, rows, trans_a=True, trans_b=True, accumulate=True, split_k=4, colsum_a=gb, group=grp[:3] + (3,))
    ref_gw = torch.zeros(L * 3 * d, d)
    ref_gw[kv_rows] = g.t() @ x
    ref_gb = torch.zeros(L * 3 * d)
    ref_gb[kv_rows] = g.sum(0)
    check(gw, ref_gw, BF, scale=math.sqrt(rows), what="grouped wgrad")           # q rows must stay exactly zero
    check(gb, ref_gb, BF, scale=math.sqrt(rows), what="grouped bias grad")
    assert float(gw.cpu()[:d].abs().max()) == 0.0
    dx = torch.empty((rows, d), dtype=BF, device=dev())
    k.gemm_row_groups(gg, wg, dx, rows, d, L * 2 * d, trans_b=True, group=grp[:3] + (2,))
    check(dx, g @ w_all[kv_rows], BF, scale=math.sqrt(L * 2 * d) / 2, what="grouped dgrad")
