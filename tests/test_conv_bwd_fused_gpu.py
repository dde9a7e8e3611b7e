"""omr_conv3x3_bwd_fused (csrc/conv_bwd_fused.hip): the one-pass backward of the <= 32-channel stride-1 convs against (i) the separate
data-gradient / weight-gradient kernels it replaces (same bf16 inputs: data gradient bit-identical -- same MFMA order --, weight
and bias gradients to fp32-atomic order) and (ii) torch's fp32 conv backward on the CPU (aten::convolution_backward, what
nn.Conv2d of the reference encoder.py:132-150 runs).  Shapes include tiles that overhang the image and a one-tile image."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _case(B, H, W, cin, cout, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn((B, H, W, cin), generator=gen).clamp_min(-0.3)            # a ReLU-like input: many exact zeros after the clamp below
    x = torch.where(x < 0, torch.zeros_like(x), x).to(BF)
    g = (torch.randn((B, H, W, cout), generator=gen) * 0.5).to(BF)
    w = (torch.randn((cout, 3, 3, cin), generator=gen) * 0.2).to(BF)
    return x, g, w


def _torch_ref(x, g, w, mask, scale):
    xn = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wn = w.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    bias = torch.zeros(w.shape[0], requires_grad=True)
    y = F.conv2d(xn, wn, bias, padding=1)
    y.backward(g.float().permute(0, 3, 1, 2))
    dx = xn.grad.permute(0, 2, 3, 1)
    if mask:
        dx = dx * (x.float() > 0) * scale
    return dx, wn.grad.permute(0, 2, 3, 1).contiguous(), bias.grad


@pytest.mark.parametrize("cout,cin", [(32, 32), (32, 16), (16, 16)])
@pytest.mark.parametrize("B,H,W", [(2, 24, 96), (1, 8, 32), (3, 13, 70)])
@pytest.mark.parametrize("mask", [True, False])
def test_fused_backward_matches_separate_kernels_and_torch(cout, cin, B, H, W, mask):
    x, g, w = _case(B, H, W, cin, cout, 7 * cout + cin + H)
    xd, gd, wd = x.to(DEV), g.to(DEV), w.to(DEV)
    wf = K.conv3x3_weight_flip(wd)
    scale = 1.25 if mask else 1.0
    dw = torch.zeros((cout, 3, 3, cin), device=DEV)
    db = torch.zeros(cout, device=DEV)
    dx = K.conv3x3_bwd_fused(gd, xd, wf, dw, db, mask, scale)
    # the kernels it replaces
    dw0 = torch.zeros_like(dw)
    db0 = torch.zeros_like(db)
    K.conv3x3_wgrad(xd, gd, dw0, db=db0)
    dx0 = K.conv3x3(gd, wf, None, out_hw=(H, W), out_mask=xd if mask else None, mask_scale=scale)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx0)
    tol = 2e-3 * float(dw0.abs().max()) + 1e-4
    assert float((dw - dw0).abs().max()) <= tol, float((dw - dw0).abs().max())
    assert float((db - db0).abs().max()) <= 2e-3 * float(db0.abs().max()) + 1e-4
    # torch fp32 on the same bf16-rounded operands
    rdx, rdw, rdb = _torch_ref(x, g, w, mask, scale)
    assert float((dx.float().cpu() - rdx).abs().max()) <= 1e-2 * float(rdx.abs().max())
    assert float((dw.cpu() - rdw).abs().max()) <= 2e-3 * float(rdw.abs().max())
    assert float((db.cpu() - rdb).abs().max()) <= 2e-3 * float(rdb.abs().max()) + 1e-3


def test_fused_backward_accumulates_and_skips_null_bias():
    x, g, w = _case(2, 16, 64, 32, 32, 5)
    xd, gd, wf = x.to(DEV), g.to(DEV), K.conv3x3_weight_flip(w.to(DEV))
    dw = torch.zeros((32, 3, 3, 32), device=DEV)
    K.conv3x3_bwd_fused(gd, xd, wf, dw, None, False)
    once = dw.clone()
    K.conv3x3_bwd_fused(gd, xd, wf, dw, None, False)
    torch.cuda.synchronize()
    assert float((dw - 2 * once).abs().max()) <= 1e-3 * float(once.abs().max())


@pytest.mark.parametrize("cout,cin", [(32, 32), (32, 16), (16, 16)])
@pytest.mark.parametrize("B,H,W", [(2, 24, 96), (3, 13, 70)])
def test_fused_backward_with_instancenorm_apply_on_load(cout, cin, B, H, W):
    """norm = (...): g is dL/d(InstanceNorm(y)); the kernel forms (y > 0) * scale * InstanceNorm-backward(g) while it loads.  Against
    omr_instnorm_bwd_apply (the stand-alone pass on the same sums) followed by the plain one-pass kernel: the data gradient within one
    bf16 rounding of the intermediate (the fused form evaluates the same affine map with its constants folded), weight / bias
    gradients to 2e-3."""
    x, ghat, w = _case(B, H, W, cin, cout, 11 * cout + cin + W)
    gen = torch.Generator().manual_seed(99)
    y = torch.randn((B, H, W, cout), generator=gen)
    y = torch.where(y < 0.2, torch.zeros_like(y), y).to(BF)
    xd, gd, yd, wf = x.to(DEV), ghat.to(DEV), y.to(DEV), K.conv3x3_weight_flip(w.to(DEV))
    mean, rstd = K.instnorm_stats(yd)
    ws, slots = K.conv_stat_ws(B, H, W, cout, DEV)
    xhat = (yd.float() - mean.view(B, 1, 1, cout)) * rstd.view(B, 1, 1, cout)
    sums = torch.stack([gd.double().sum(dim=(1, 2)), (gd.float() * xhat).double().sum(dim=(1, 2))], dim=-1)      # [B, C, 2]
    wsv = ws.view(-1)
    wsv.zero_()
    wsv[: B * slots * cout * 2].view(B, slots, cout, 2)[:, 0] = sums
    K.instnorm_reduce_sums(ws, slots, B, cout)
    scale = 1.6
    g_ref = K.instnorm_bwd_apply(gd, yd, mean, rstd, ws, slots, True, scale)
    dw0 = torch.zeros((cout, 3, 3, cin), device=DEV)
    db0 = torch.zeros(cout, device=DEV)
    dx0 = K.conv3x3_bwd_fused(g_ref, xd, wf, dw0, db0, True, 1.25)
    dw = torch.zeros_like(dw0)
    db = torch.zeros_like(db0)
    dx = K.conv3x3_bwd_fused(gd, xd, wf, dw, db, True, 1.25, norm=(yd, mean, rstd, ws, slots, True, scale))
    torch.cuda.synchronize()
    ref = float(dx0.float().abs().max())
    assert float((dx.float() - dx0.float()).abs().max()) <= 2e-2 * ref
    assert float((dx.float() - dx0.float()).norm() / dx0.float().norm()) < 4e-3
    assert float((dw - dw0).abs().max()) <= 4e-3 * float(dw0.abs().max())
    assert float((db - db0).abs().max()) <= 4e-3 * float(db0.abs().max()) + 1e-3
    assert torch.equal((dx == 0), (dx0 == 0)) or float(((dx == 0) != (dx0 == 0)).float().mean()) < 1e-3


@pytest.mark.parametrize("B,H,W", [(2, 24, 96), (3, 13, 70), (1, 8, 32)])
def test_fused_backward_of_the_normalise_on_load_conv(B, H, W):
    """xnorm = (...): ConvBlock 0's conv3 (16 -> 16, InstanceNorm applied on load).  Against the separate kernels: omr_conv3x3_fwd
    stat_mode 2 (data gradient + InstanceNorm-backward sums) and omr_conv3x3_wgrad with in_stats: data gradient bit-identical, the sums
    (after the slot reduction) to 3e-3 of the largest -- the one-pass form multiplies by the bf16 xhat it holds in LDS, the separate kernel by the
    fp32 one --, weight / bias gradients to 4e-3 (same reason), and the apply pass on either workspace gives the same gradient."""
    c = 16
    x, g, w = _case(B, H, W, c, c, 1000 + H)
    xd, gd, wf = x.to(DEV), g.to(DEV), K.conv3x3_weight_flip(w.to(DEV))
    mean, rstd = K.instnorm_stats(xd)
    ws0, slots = K.conv_stat_ws(B, H, W, c, DEV)
    dx0 = K.conv3x3(gd, wf, None, out_hw=(H, W), stat_mode=2, stat_ws=ws0, stat_slots=slots, stat_x=xd, stat_stats=(mean, rstd))
    dw0 = torch.zeros((c, 3, 3, c), device=DEV)
    db0 = torch.zeros(c, device=DEV)
    K.conv3x3_wgrad(xd, gd, dw0, in_stats=(mean, rstd), db=db0)
    ws, _ = K.conv_stat_ws(B, H, W, c, DEV)
    ws.view(-1).fill_(123.0)                          # every slot must be written
    dw = torch.zeros_like(dw0)
    db = torch.zeros_like(db0)
    dx = K.conv3x3_bwd_fused(gd, xd, wf, dw, db, False, 1.0, xnorm=(mean, rstd, ws, slots))
    torch.cuda.synchronize()
    assert torch.equal(dx, dx0)
    a0 = K.instnorm_bwd_apply(dx0, xd, mean, rstd, ws0, slots, True, 1.3)
    a1 = K.instnorm_bwd_apply(dx, xd, mean, rstd, ws, slots, True, 1.3)
    torch.cuda.synchronize()
    n = B * slots * c * 2
    s0, s1 = ws0.view(-1)[n:n + B * c * 2], ws.view(-1)[n:n + B * c * 2]      # the compact sums the apply pass left behind the slots
    assert float((s0 - s1).abs().max()) <= 3e-3 * float(s0.abs().max()) + 1e-6, (float((s0 - s1).abs().max()), float(s0.abs().max()))
    assert float((a0.float() - a1.float()).abs().max()) <= 1e-2 * float(a0.float().abs().max())
    assert float((dw - dw0).abs().max()) <= 4e-3 * float(dw0.abs().max())
    assert float((db - db0).abs().max()) <= 2e-3 * float(db0.abs().max()) + 1e-4


@pytest.mark.parametrize("B,H,W", [(2, 24, 96), (3, 16, 64), (1, 13, 70), (2, 9, 33)])
def test_fused_backward_of_the_strided_normalise_on_load_conv(B, H, W):
    """omr_conv3x3_bwd_fused_s2: ConvBlock 1's conv3 (32 -> 32, stride (2, 2), InstanceNorm applied on load).  Against the separate kernels
    -- omr_conv3x3_fwd as the zero-dilated data gradient with stat_mode 2, omr_conv3x3_wgrad with stride and in_stats -- on the same
    bf16 operands: data gradient equal to a bf16 rounding (odd sizes included: tiles that overhang, a last output row / column that
    sees one input row / column), sums to 3e-3 of the largest, weight / bias gradients to 4e-3."""
    c = 32
    gen = torch.Generator().manual_seed(500 + H)
    x = torch.randn((B, H, W, c), generator=gen)
    x = torch.where(x < 0, torch.zeros_like(x), x).to(BF)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    g = (torch.randn((B, Ho, Wo, c), generator=gen) * 0.5).to(BF)
    w = (torch.randn((c, 3, 3, c), generator=gen) * 0.2).to(BF)
    xd, gd, wf = x.to(DEV), g.to(DEV), K.conv3x3_weight_flip(w.to(DEV))
    mean, rstd = K.instnorm_stats(xd)
    ws0, slots = K.conv_stat_ws(B, H, W, c, DEV)
    dx0 = K.conv3x3(gd, wf, None, stride=(1, 1), dil=(2, 2), out_hw=(H, W), stat_mode=2, stat_ws=ws0, stat_slots=slots, stat_x=xd, stat_stats=(mean, rstd))
    dw0 = torch.zeros((c, 3, 3, c), device=DEV)
    db0 = torch.zeros(c, device=DEV)
    K.conv3x3_wgrad(xd, gd, dw0, stride=(2, 2), in_stats=(mean, rstd), db=db0)
    ws, _ = K.conv_stat_ws(B, H, W, c, DEV)
    ws.view(-1).fill_(77.0)
    dw = torch.zeros_like(dw0)
    db = torch.zeros_like(db0)
    dx = K.conv3x3_bwd_fused_s2(gd, xd, wf, dw, db, mean, rstd, ws, slots)
    torch.cuda.synchronize()
    # (the two kernels sum the taps in different orders: equal to a bf16 rounding, not to the bit)
    assert float((dx.float() - dx0.float()).abs().max()) <= 8e-3 * float(dx0.float().abs().max())
    assert float((dx.float() - dx0.float()).norm() / dx0.float().norm()) < 2e-3
    K.instnorm_bwd_apply(dx0, xd, mean, rstd, ws0, slots, True, 1.0)
    K.instnorm_bwd_apply(dx, xd, mean, rstd, ws, slots, True, 1.0)
    torch.cuda.synchronize()
    n = B * slots * c * 2
    s0, s1 = ws0.view(-1)[n:n + B * c * 2], ws.view(-1)[n:n + B * c * 2]
    assert float((s0 - s1).abs().max()) <= 3e-3 * float(s0.abs().max()) + 1e-6
    assert float((dw - dw0).abs().max()) <= 4e-3 * float(dw0.abs().max())
    assert float((db - db0).abs().max()) <= 2e-3 * float(db0.abs().max()) + 1e-4
