"""Host-side launchers: tensor-level wrappers over the C ABI (include/omr_hip.h) with shape checks.

No autograd here (see functional.py).  Layout conventions: encoder activations are NHWC tensors
[B,H,W,C]; 3x3 conv weights are given as their physical [COUT,3,3,CIN] view; token matrices are 2-D
row-major with unit inner stride (row stride may exceed the width).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch

from ._lib import cur_stream, dtype_code, lib, ptr, require_cuda

Tensor = torch.Tensor


def _rows2d(t: Tensor) -> Tuple[int, int, int]:
    assert t.dim() == 2 and t.stride(1) == 1, f"expected a row-major 2-D tensor, got shape {tuple(t.shape)} strides {t.stride()}"
    return t.shape[0], t.shape[1], t.stride(0)


def gemm(a: Tensor, b: Tensor, *, trans_a: bool = False, trans_b: bool = False, bias: Optional[Tensor] = None, relu: bool = False,
         out: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None, accumulate: bool = False, split_k: int = 1,
         colsum_a: Optional[Tensor] = None, drop: Optional[Tuple[float, int]] = None) -> Tensor:
    """C[M,N] (+)= act(opA(a) @ opB(b)^T + bias).  a is [M,K] ([K,M] if trans_a); b is [N,K] ([K,N] if trans_b)."""
    require_cuda(a, b, bias, out)
    assert a.dtype == b.dtype
    ar, ac, lda = _rows2d(a)
    br, bc, ldb = _rows2d(b)
    M, K = (ac, ar) if trans_a else (ar, ac)
    N, Kb = (bc, br) if trans_b else (br, bc)
    assert K == Kb, f"gemm inner dims differ: {K} vs {Kb}"
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
        assert not accumulate and split_k == 1, "accumulating gemm needs an initialised `out`"
    cr, cc, ldc = _rows2d(out)
    assert (cr, cc) == (M, N), f"gemm out shape {tuple(out.shape)} != {(M, N)}"
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if colsum_a is not None:
        assert trans_a and colsum_a.dtype == torch.float32 and colsum_a.numel() == M and colsum_a.is_contiguous()
    lib().call("omr_gemm", dtype_code(a.dtype), dtype_code(out.dtype), int(trans_a), int(trans_b), M, N, K, ptr(a), lda, ptr(b), ldb,
               ptr(out), ldc, ptr(bias), int(relu), int(accumulate), split_k, ptr(colsum_a), float(drop[0]) if drop else 0.0,
               (int(drop[1]) & (2**64 - 1)) if drop else 0, 0, 0, 0, 0, cur_stream())
    return out


def gemm_row_groups(a: Tensor, b: Tensor, out: Tensor, M: int, N: int, Kd: int, *, trans_a: bool = False, trans_b: bool = False,
                    bias: Optional[Tensor] = None, accumulate: bool = False, split_k: int = 1, colsum_a: Optional[Tensor] = None,
                    group: Tuple[int, int, int, int] = (0, 0, 0, 0)) -> Tensor:
    """omr_gemm over a row-group view of the weight-side operand (include/omr_hip.h): `group` = (rows per group, physical
    group stride, first row inside a group, operand 1|2|3).  The grouped tensor (b, or out/colsum_a/bias) is the PHYSICAL
    2-D block that contains every group; M, N, Kd are the logical GEMM sizes, so only strides are taken from the tensors."""
    require_cuda(a, b, bias, out, colsum_a)
    assert a.dtype == b.dtype and a.dim() == b.dim() == out.dim() == 2 and a.stride(1) == b.stride(1) == out.stride(1) == 1
    grp, stride, base, operand = group
    ngroups = {1: N, 2: Kd, 3: M}[operand] // grp
    phys_rows = (ngroups - 1) * stride + base + grp
    grouped = out if operand == 3 else b
    assert grouped.shape[0] >= phys_rows, f"grouped operand has {grouped.shape[0]} rows, the view needs {phys_rows}"
    for vec, need in ((bias, operand == 1), (colsum_a, operand == 3)):
        if vec is not None:
            assert vec.dtype == torch.float32 and vec.is_contiguous() and vec.numel() >= (phys_rows if need else 0)
    lib().call("omr_gemm", dtype_code(a.dtype), dtype_code(out.dtype), int(trans_a), int(trans_b), M, N, Kd, ptr(a), a.stride(0), ptr(b), b.stride(0),
               ptr(out), out.stride(0), ptr(bias), 0, int(accumulate), split_k, ptr(colsum_a), 0.0, 0, grp, stride, base, operand, cur_stream())
    return out


class _DwProblem(ctypes.Structure):
    """omr_dw_problem of include/omr_hip.h."""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("rows", ctypes.c_int), ("n_out", ctypes.c_int), ("n_in", ctypes.c_int), ("row_group", ctypes.c_int),
                ("ld_dy", ctypes.c_long), ("ld_x", ctypes.c_long), ("ld_dw", ctypes.c_long),
                ("row_group_stride", ctypes.c_int), ("row_group_base", ctypes.c_int)]


class _DecodeLinearArgs(ctypes.Structure):
    """omr_decode_linear_args of include/omr_hip.h."""
    _fields_ = [(n, ctypes.c_int) for n in ("dtype", "pro", "M", "N", "K", "relu", "n0", "nsplit", "H", "hd", "vocab", "pad_")] + \
               [("eps", ctypes.c_float), ("pad2_", ctypes.c_float),
                ("x", ctypes.c_void_p), ("ldx", ctypes.c_long), ("res", ctypes.c_void_p), ("ldres", ctypes.c_long),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("xn_out", ctypes.c_void_p),
                ("tokens", ctypes.c_void_p), ("emb", ctypes.c_void_p), ("pe_row", ctypes.c_void_p), ("part", ctypes.c_void_p),
                ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("out0", ctypes.c_void_p), ("ld0", ctypes.c_long),
                ("out1", ctypes.c_void_p), ("ld1", ctypes.c_long), ("out32", ctypes.c_void_p), ("ld32", ctypes.c_long),
                ("amax_idx", ctypes.c_void_p), ("amax_val", ctypes.c_void_p), ("amax_part", ctypes.c_void_p),
                ("w8", ctypes.c_void_p), ("w8_scale", ctypes.c_void_p)]


def decode_linear(w: Tensor, bias: Optional[Tensor], *, x: Optional[Tensor] = None, res: Optional[Tensor] = None, ln=None, tokens: Optional[Tensor] = None,
                  emb: Optional[Tensor] = None, pe_row: Optional[Tensor] = None, part: Optional[Tensor] = None, heads: int = 0, relu: bool = False,
                  n0: Optional[int] = None, want32: bool = False, want_argmax: bool = False):
    """One linear of a decode position with the preceding element-wise step folded into its input rows (omr_decode_linear):
    x alone -> plain rows; x + res + ln=(gamma, beta, eps) -> LayerNorm(x + res); tokens + emb + pe_row -> embedding row + pe;
    part [M*heads, nsplit, hd+2] + heads -> merged key-split attention.  Returns (out0 [M, n0], out1 [M, N-n0] or None,
    built rows [M, K] or None, fp32 copy or None) and, with want_argmax, (first index of the row maximum int64 [M], its value fp32 [M])."""
    require_cuda(w, bias, x, res, tokens, emb, pe_row, part)
    N, K = w.shape
    dt = w.dtype
    a = _DecodeLinearArgs()
    built = None
    if part is not None:
        pro, M = 3, part.shape[0] // heads
        assert part.dtype == torch.float32 and part.is_contiguous() and part.shape[2] - 2 == K // heads
        a.part, a.nsplit, a.H, a.hd = part.data_ptr(), part.shape[1], heads, K // heads
    elif tokens is not None:
        pro, M = 2, tokens.numel()
        assert tokens.dtype == torch.int64 and emb.dtype == dt and emb.is_contiguous() and pe_row.dtype == torch.float32 and pe_row.numel() == K
        built = torch.empty((M, K), dtype=dt, device=w.device)
        a.tokens, a.emb, a.pe_row, a.vocab, a.xn_out = tokens.data_ptr(), emb.data_ptr(), pe_row.data_ptr(), emb.shape[0], built.data_ptr()
    elif ln is not None:
        pro, M = 1, x.shape[0]
        gamma, beta, eps = ln
        assert x.dtype == res.dtype == dt and x.stride(1) == res.stride(1) == 1 and gamma.dtype == beta.dtype == torch.float32
        built = torch.empty((M, K), dtype=dt, device=w.device)
        a.x, a.ldx, a.res, a.ldres, a.gamma, a.beta, a.eps, a.xn_out = x.data_ptr(), x.stride(0), res.data_ptr(), res.stride(0), gamma.data_ptr(), beta.data_ptr(), eps, built.data_ptr()
    else:
        pro, M = 0, x.shape[0]
        assert x.dtype == dt and x.stride(1) == 1 and x.shape[1] == K
        a.x, a.ldx = x.data_ptr(), x.stride(0)
    n0 = N if n0 is None else n0
    out0 = torch.empty((M, n0), dtype=dt, device=w.device)
    out1 = torch.empty((M, N - n0), dtype=dt, device=w.device) if n0 < N else None
    out32 = torch.empty((M, N), dtype=torch.float32, device=w.device) if want32 else None
    assert w.is_contiguous() and (bias is None or (bias.dtype == torch.float32 and bias.numel() == N))
    a.dtype, a.pro, a.M, a.N, a.K, a.relu, a.n0 = dtype_code(dt), pro, M, N, K, int(relu), n0
    a.w, a.bias, a.out0, a.ld0 = w.data_ptr(), (bias.data_ptr() if bias is not None else None), out0.data_ptr(), n0
    if out1 is not None:
        a.out1, a.ld1 = out1.data_ptr(), N - n0
    if out32 is not None:
        a.out32, a.ld32 = out32.data_ptr(), N
    amax = None
    if want_argmax:
        idx = torch.empty(M, dtype=torch.int64, device=w.device)
        val = torch.empty(M, dtype=torch.float32, device=w.device)
        part = torch.empty(2 * M * ((N + 15) // 16), dtype=torch.float32, device=w.device)
        a.amax_idx, a.amax_val, a.amax_part = idx.data_ptr(), val.data_ptr(), part.data_ptr()
        amax = (idx, val)
    lib().call("omr_decode_linear", ctypes.byref(a), cur_stream())
    return (out0, out1, built, out32) if amax is None else (out0, out1, built, out32, amax)


def linear_wgrad_grouped(problems) -> None:
    """Weight / bias gradients of several linear layers in ONE launch (omr_linear_wgrad_grouped).  problems: sequence of
    (dy [rows, n_out], x [rows, n_in], dw fp32 [n_out(+), n_in] accumulated in place, db fp32 or None, group) with
    group = None or (rows per group, physical group stride, first row inside a group) for a row-group view of dw / db."""
    if not problems:
        return
    arr = (_DwProblem * len(problems))()
    dt = problems[0][0].dtype
    for q, (dy, x, dw, db, group) in zip(arr, problems):
        require_cuda(dy, x, dw, db)
        assert dy.dtype == x.dtype == dt and dw.dtype == torch.float32 and dy.dim() == x.dim() == dw.dim() == 2
        assert dy.stride(1) == x.stride(1) == dw.stride(1) == 1 and dy.shape[0] == x.shape[0]
        rows, n_out = dy.shape
        n_in = x.shape[1]
        grp, gstride, gbase = group if group is not None else (0, 0, 0)
        phys_rows = n_out if not grp else (n_out // grp - 1) * gstride + gbase + grp
        assert dw.shape[1] == n_in and dw.shape[0] >= phys_rows, (tuple(dw.shape), n_out, n_in)
        if db is not None:
            assert db.dtype == torch.float32 and db.is_contiguous() and db.numel() >= phys_rows
        q.dy, q.x, q.dw, q.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
        q.rows, q.n_out, q.n_in, q.row_group = rows, n_out, n_in, grp
        q.ld_dy, q.ld_x, q.ld_dw, q.row_group_stride, q.row_group_base = dy.stride(0), x.stride(0), dw.stride(0), gstride, gbase
    lib().call("omr_linear_wgrad_grouped", dtype_code(dt), len(problems), ctypes.byref(arr), cur_stream())


def quantize_rows_fp8(x: Tensor) -> Tuple[Tensor, Tensor]:
    """x [M, K] (fp32 / bf16, unit inner stride) -> (OCP e4m3 codes uint8 [M, K], fp32 scale per row [M]): x ~ codes * scale."""
    require_cuda(x)
    M, K, ld = _rows2d(x)
    q = torch.empty((M, (K + 15) // 16 * 16), dtype=torch.uint8, device=x.device)
    if q.shape[1] != K:
        q.zero_()                                # padded k columns must read as fp8 zeros
    scale = torch.empty(M, dtype=torch.float32, device=x.device)
    lib().call("omr_quantize_rows_fp8", dtype_code(x.dtype), ptr(x), ld, ptr(q), q.stride(0), ptr(scale), M, K, cur_stream())
    return q, scale


def gemm_fp8(a8: Tensor, scale_a: Tensor, w8: Tensor, scale_w: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
             out_dtype: torch.dtype = torch.bfloat16, out: Optional[Tensor] = None) -> Tensor:
    """C[M, N] = (a8 . w8^T) * scale_a[m] * scale_w[n] + bias on the fp8 MFMA (a8 [M, K], w8 [N, K]: quantize_rows_fp8 outputs)."""
    require_cuda(a8, w8, scale_a, scale_w, bias, out)
    assert a8.dtype == w8.dtype == torch.uint8 and a8.shape[1] == w8.shape[1] and a8.stride(1) == w8.stride(1) == 1
    M, K = a8.shape
    N = w8.shape[0]
    assert scale_a.numel() == M and scale_w.numel() == N and scale_a.dtype == scale_w.dtype == torch.float32
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    assert tuple(out.shape) == (M, N) and out.stride(1) == 1
    lib().call("omr_gemm_fp8", dtype_code(out.dtype), M, N, K, ptr(a8), a8.stride(0), ptr(scale_a), ptr(w8), w8.stride(0), ptr(scale_w), ptr(out),
               out.stride(0), ptr(bias), int(relu), cur_stream())
    return out


def cast(x: Tensor, dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    require_cuda(x)
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x, dtype=dtype)
    lib().call("omr_cast", ptr(x), dtype_code(x.dtype), ptr(out), dtype_code(dtype), x.numel(), cur_stream())
    return out


def add(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    require_cuda(a, b)
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    lib().call("omr_add", dtype_code(a.dtype), ptr(a), ptr(b), ptr(out), a.numel(), cur_stream())
    return out


def relu_bwd(dy: Tensor, y: Tensor, scale: float = 1.0) -> Tensor:
    require_cuda(dy, y)
    assert dy.shape == y.shape and dy.dtype == y.dtype and dy.is_contiguous() and y.is_contiguous()
    dx = torch.empty_like(dy)
    lib().call("omr_relu_bwd", dtype_code(dy.dtype), ptr(dy), ptr(y), ptr(dx), dy.numel(), float(scale), cur_stream())
    return dx


def dropout(x: Tensor, p: float, seed: int, channel_mode: bool = False) -> Tensor:
    """Elementwise dropout, or per-(sample, channel) dropout of an NHWC tensor when channel_mode."""
    require_cuda(x)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    per_sample = x.numel() // x.shape[0]
    lib().call("omr_dropout", dtype_code(x.dtype), ptr(x), ptr(out), x.numel(), float(p), int(seed) & (2**64 - 1), int(channel_mode),
               per_sample, x.shape[-1], cur_stream())
    return out


def embed_pe(tokens: Tensor, table: Tensor, pe: Tensor) -> Tensor:
    """tokens int64 [B,T]; table [V,d]; pe fp32 [>=T, d] -> [B,T,d]."""
    require_cuda(tokens, table, pe)
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and table.is_contiguous()
    B, T = tokens.shape
    V, d = table.shape
    assert pe.dtype == torch.float32 and pe.is_contiguous() and pe.shape[-1] == d and pe.shape[-2] >= T, "sequence longer than the PE table"
    out = torch.empty((B, T, d), dtype=table.dtype, device=table.device)
    lib().call("omr_embed_pe_fwd", dtype_code(table.dtype), ptr(tokens), ptr(table), ptr(pe), ptr(out), B * T, T, d, V, cur_stream())
    return out


def embed_bwd(tokens: Tensor, dout: Tensor, dtable: Tensor, pad_idx: int) -> None:
    require_cuda(tokens, dout, dtable)
    assert dtable.dtype == torch.float32 and dout.is_contiguous()
    V, d = dtable.shape
    lib().call("omr_embed_bwd", dtype_code(dout.dtype), ptr(tokens), ptr(dout), ptr(dtable), tokens.numel(), d, pad_idx, V, cur_stream())


def add_pe2d(x: Tensor, pe_hwc: Tensor) -> Tensor:
    """x NHWC [B,h,w,C] + pe_hwc fp32 [maxh,maxw,C][:h,:w]."""
    require_cuda(x, pe_hwc)
    B, h, w, C = x.shape
    maxh, maxw, Cp = pe_hwc.shape
    assert Cp == C and pe_hwc.dtype == torch.float32 and pe_hwc.is_contiguous() and x.is_contiguous()
    assert h <= maxh and w <= maxw, f"feature map {h}x{w} exceeds the positional-encoding table {maxh}x{maxw}"
    out = torch.empty_like(x)
    lib().call("omr_add_pe2d", dtype_code(x.dtype), ptr(x), ptr(pe_hwc), ptr(out), B, h, w, C, maxh, maxw, cur_stream())
    return out


def colsum_into(dy2d: Tensor, db: Tensor) -> None:
    """db[n] += sum_m dy2d[m, n]  (db fp32)."""
    require_cuda(dy2d, db)
    M, N, ld = _rows2d(dy2d)
    assert db.dtype == torch.float32 and db.numel() == N
    lib().call("omr_colsum", dtype_code(dy2d.dtype), ptr(dy2d), ptr(db), M, N, ld, cur_stream())


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
              grad_scale: float = 1.0, p_lowp: Optional[Tensor] = None) -> None:
    require_cuda(p, g, m, v)
    n = p.numel()
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    if p_lowp is not None:
        assert p_lowp.dtype == torch.bfloat16 and p_lowp.numel() == n
    lib().call("omr_adam", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_lowp), n, step, lr, betas[0], betas[1], eps, grad_scale, cur_stream())


def argmax(x: Tensor) -> Tuple[Tensor, Tensor]:
    """x fp32 [n] or [rows, n] (unit inner stride) -> (indices int64 [rows], top-1 values fp32 [rows])."""
    require_cuda(x)
    assert x.dtype == torch.float32 and x.dim() in (1, 2) and x.stride(-1) == 1
    rows, n, ld = (1, x.numel(), x.numel()) if x.dim() == 1 else (x.shape[0], x.shape[1], x.stride(0))
    idx = torch.empty(rows, dtype=torch.int64, device=x.device)
    val = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib().call("omr_argmax", ptr(x), rows, n, ld, ptr(idx), ptr(val), cur_stream())
    return idx, val


def weighted_argmax(la: Tensor, lb: Tensor, alpha: float) -> Tuple[Tensor, Tensor]:
    """argmax(alpha * softmax(la) + (1 - alpha) * softmax(lb)) of two fp32 logit rows [n] -> (index int64 [1], probability fp32 [1])."""
    require_cuda(la, lb)
    assert la.dtype == lb.dtype == torch.float32 and la.dim() == lb.dim() == 1 and la.numel() == lb.numel() and la.is_contiguous() and lb.is_contiguous()
    idx = torch.empty(1, dtype=torch.int64, device=la.device)
    prob = torch.empty(1, dtype=torch.float32, device=la.device)
    lib().call("omr_weighted_argmax", ptr(la), ptr(lb), la.numel(), float(alpha), ptr(idx), ptr(prob), cur_stream())
    return idx, prob


def topk_logprob(x: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """x fp32 [rows, n] -> (token ids int64 [rows, k], log-probabilities fp32 [rows, k]), best first."""
    require_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and 0 < k <= x.shape[1]
    idx = torch.empty((x.shape[0], k), dtype=torch.int64, device=x.device)
    val = torch.empty((x.shape[0], k), dtype=torch.float32, device=x.device)
    lib().call("omr_topk_logprob", ptr(x), x.shape[0], x.shape[1], x.stride(0), k, ptr(idx), ptr(val), cur_stream())
    return idx, val


# ------------------------------------------------------------------------------------------------ normalisation

def _stat_ws(B: int, C: int, slots: int, device, zero: bool) -> Tensor:
    """fp64 statistics workspace: partial slots [B][slots][C][2] + compact sums [B][C][2] (include/omr_hip.h, normalisation)."""
    n = lib().query("omr_instnorm_workspace_bytes", B, C, slots) // 8
    return (torch.zeros if zero else torch.empty)(n, dtype=torch.float64, device=device)


def conv_stat_ws(B: int, Ho: int, Wo: int, C: int, device) -> Tuple[Tensor, int]:
    """Workspace + slot count for a conv launch with a fused statistics epilogue (stat_mode 1 / 2); the launch writes every slot."""
    slots = lib().query("omr_conv3x3_stat_slots", B, Ho, Wo)
    return _stat_ws(B, C, slots, device, False), slots


def instnorm_stats(x: Tensor, eps: float = 1e-3) -> Tuple[Tensor, Tensor]:
    """x NHWC -> (mean, rstd) fp32 [B,C]."""
    require_cuda(x)
    B, H, W, C = x.shape
    assert x.is_contiguous()
    mean = torch.empty((B, C), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = _stat_ws(B, C, lib().query("omr_instnorm_slots", B, H * W), x.device, False)
    lib().call("omr_instnorm_stats", dtype_code(x.dtype), ptr(x), ptr(mean), ptr(rstd), B, H * W, C, eps, ptr(ws), cur_stream())
    return mean, rstd


def instnorm_finalize(ws: Tensor, slots: int, B: int, C: int, HW: int, eps: float = 1e-3) -> Tuple[Tensor, Tensor]:
    """fp64 {sum, sum of squares} slots (conv_stat_ws layout) -> (mean, rstd) fp32 [B,C]."""
    require_cuda(ws)
    mean = torch.empty((B, C), dtype=torch.float32, device=ws.device)
    rstd = torch.empty_like(mean)
    lib().call("omr_instnorm_finalize", ptr(ws), slots, ptr(mean), ptr(rstd), B, HW, C, eps, cur_stream())
    return mean, rstd


def instnorm_bwd_apply(dxhat: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, ws: Tensor, slots: int, relu_mask: bool, relu_scale: float = 1.0) -> Tensor:
    """InstanceNorm backward apply step with the {sum g, sum g*xhat} sums already in the slots of ws (conv_stat_ws layout)."""
    require_cuda(dxhat, x, ws)
    B, H, W, C = x.shape
    assert dxhat.shape == x.shape and dxhat.is_contiguous() and x.is_contiguous() and ws.dtype == torch.float64
    dx = torch.empty_like(x)
    lib().call("omr_instnorm_bwd_apply", dtype_code(x.dtype), ptr(dxhat), ptr(x), ptr(mean), ptr(rstd), ptr(dx), B, H * W, C, int(relu_mask),
               float(relu_scale), ptr(ws), slots, cur_stream())
    return dx


def instnorm_bwd(dxhat: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, relu_mask: bool, relu_scale: float = 1.0) -> Tensor:
    require_cuda(dxhat, x)
    B, H, W, C = x.shape
    assert dxhat.shape == x.shape and dxhat.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    ws = _stat_ws(B, C, lib().query("omr_instnorm_slots", B, H * W), x.device, False)
    lib().call("omr_instnorm_bwd", dtype_code(x.dtype), ptr(dxhat), ptr(x), ptr(mean), ptr(rstd), ptr(dx), B, H * W, C, int(relu_mask),
               float(relu_scale), ptr(ws), cur_stream())
    return dx


def add_layernorm_fwd(x: Tensor, res: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float = 1e-5, drop_p: float = 0.0, drop_seed: int = 0):
    """LayerNorm(dropout(x) + res); drop_p = 0 is the plain residual add."""
    require_cuda(x, res, gamma, beta)
    d = x.shape[-1]
    M = x.numel() // d
    assert x.is_contiguous() and (res is None or (res.is_contiguous() and res.shape == x.shape))
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == d
    out = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib().call("omr_add_layernorm_fwd", dtype_code(x.dtype), ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(out), ptr(mean), ptr(rstd), M, d, eps,
               float(drop_p), int(drop_seed) & (2**64 - 1), cur_stream())
    return out, mean, rstd


def add_layernorm_bwd(dy: Tensor, x: Tensor, res: Optional[Tensor], gamma: Tensor, mean: Tensor, rstd: Tensor, dgamma: Tensor, dbeta: Tensor,
                      drop_p: float = 0.0, drop_seed: int = 0):
    """Returns ds (gradient of res, and of x without dropout); with drop_p > 0 returns (ds, dx)."""
    require_cuda(dy, x)
    d = x.shape[-1]
    M = x.numel() // d
    assert dy.is_contiguous() and dy.shape == x.shape and dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32
    ds = torch.empty_like(x)
    dx = torch.empty_like(x) if drop_p > 0.0 else None
    lib().call("omr_add_layernorm_bwd", dtype_code(x.dtype), ptr(dy), ptr(x), ptr(res), ptr(gamma), ptr(mean), ptr(rstd), ptr(ds), ptr(dgamma),
               ptr(dbeta), M, d, float(drop_p), int(drop_seed) & (2**64 - 1), ptr(dx), cur_stream())
    return ds if dx is None else (ds, dx)


# ------------------------------------------------------------------------------------------------ convolutions

def conv_out_hw(H: int, W: int, stride: Tuple[int, int]) -> Tuple[int, int]:
    """k=3, pad=1: out = ceil(in / stride)."""
    return (H + stride[0] - 1) // stride[0], (W + stride[1] - 1) // stride[1]


def conv3x3(x: Tensor, w_phys: Tensor, bias: Optional[Tensor], stride=(1, 1), relu: bool = False, in_stats=None, out_mask: Optional[Tensor] = None,
            mask_scale: float = 1.0, dil=(1, 1), out_hw: Optional[Tuple[int, int]] = None, drop=None, stat_mode: int = 0,
            stat_ws: Optional[Tensor] = None, stat_slots: int = 0, stat_x: Optional[Tensor] = None, stat_stats=None) -> Tensor:
    """x NHWC [B,H,W,CIN]; w_phys [COUT,3,3,CIN] contiguous; returns NHWC [B,Ho,Wo,COUT].
    drop = (p, seed, channel_mode): fused dropout after the ReLU.  stat_mode 1/2: fused per-(image, channel) reductions of
    the stored output into the fp64 slot workspace (stat_ws, stat_slots) = conv_stat_ws(...) (see include/omr_hip.h)."""
    require_cuda(x, w_phys, bias, out_mask, stat_ws, stat_x)
    B, H, W, CIN = x.shape
    COUT = w_phys.shape[0]
    assert x.is_contiguous() and w_phys.is_contiguous() and tuple(w_phys.shape) == (COUT, 3, 3, CIN) and w_phys.dtype == x.dtype
    if out_hw is None:
        out_hw = conv_out_hw(H, W, stride)
    Ho, Wo = out_hw
    y = torch.empty((B, Ho, Wo, COUT), dtype=x.dtype, device=x.device) if stat_mode != 4 else None      # mode 4 takes sums only
    mean = rstd = None
    if in_stats is not None:
        mean, rstd = in_stats
        assert tuple(mean.shape) == (B, CIN) and mean.dtype == torch.float32
    if out_mask is not None:
        assert tuple(out_mask.shape) == (B, Ho, Wo, COUT) and out_mask.is_contiguous() and out_mask.dtype == x.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == COUT
    p, seed, chan = drop if drop is not None else (0.0, 0, False)
    smean = srstd = None
    if stat_mode:
        assert stat_ws is not None and stat_ws.dtype == torch.float64 and stat_slots >= 1 and stat_ws.numel() >= B * stat_slots * COUT * 2
    if stat_mode >= 2:
        smean, srstd = stat_stats
        assert stat_x is not None and tuple(stat_x.shape) == (B, Ho, Wo, COUT) and stat_x.is_contiguous() and tuple(smean.shape) == (B, COUT)
        assert stat_ws.numel() >= B * stat_slots * COUT * 2 + B * COUT * 2
    lib().call("omr_conv3x3_fwd", dtype_code(x.dtype), ptr(x), ptr(w_phys), ptr(bias), ptr(y), ptr(mean), ptr(rstd), ptr(out_mask), float(mask_scale),
               B, H, W, CIN, COUT, stride[0], stride[1], dil[0], dil[1], Ho, Wo, int(relu), float(p), int(seed) & (2**64 - 1), int(chan),
               int(stat_mode), ptr(stat_ws), int(stat_slots), ptr(stat_x), ptr(smean), ptr(srstd), cur_stream())
    return y


def instnorm_reduce_sums(ws: Tensor, slots: int, B: int, C: int) -> None:
    """Per-image sums of the slots a conv3x3(stat_mode=2 / 4) launch filled, into the compact region of the same workspace."""
    require_cuda(ws)
    lib().call("omr_instnorm_reduce_sums", ptr(ws), slots, B, C, cur_stream())


def conv3x3_weight_flip(w_phys: Tensor) -> Tensor:
    """[COUT,3,3,CIN] -> [CIN,3,3,COUT] with mirrored taps (weights of the data-gradient conv)."""
    require_cuda(w_phys)
    COUT, _, _, CIN = w_phys.shape
    wd = torch.empty((CIN, 3, 3, COUT), dtype=w_phys.dtype, device=w_phys.device)
    lib().call("omr_conv3x3_weight_flip", dtype_code(w_phys.dtype), ptr(w_phys), ptr(wd), COUT, CIN, cur_stream())
    return wd


class _FlipDesc(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("wd", ctypes.c_void_p), ("cout", ctypes.c_int), ("cin", ctypes.c_int)]


def conv3x3_weight_flip_table(pairs):
    """Host table for conv3x3_weight_flip_grouped: pairs of (w_phys [COUT,3,3,CIN], wd [CIN,3,3,COUT]) in one dtype.  Built once
    (the tensors are fixed views of the flat parameter buffers) and reused every optimizer step."""
    arr = (_FlipDesc * len(pairs))()
    dt = pairs[0][0].dtype
    for q, (w, wd) in zip(arr, pairs):
        require_cuda(w, wd)
        COUT, _, _, CIN = w.shape
        assert w.dtype == wd.dtype == dt and w.is_contiguous() and wd.is_contiguous() and tuple(wd.shape) == (CIN, 3, 3, COUT)
        q.w, q.wd, q.cout, q.cin = w.data_ptr(), wd.data_ptr(), COUT, CIN
    return dtype_code(dt), len(pairs), arr


def conv3x3_weight_flip_grouped(table) -> None:
    """All data-gradient weight re-layouts of a model in ONE launch (omr_conv3x3_weight_flip_grouped)."""
    code, n, arr = table
    lib().call("omr_conv3x3_weight_flip_grouped", code, n, ctypes.byref(arr), cur_stream())


def conv3x3_wgrad(x: Tensor, dy: Tensor, dw_phys: Tensor, stride=(1, 1), in_stats=None, db: Optional[Tensor] = None) -> None:
    """dw_phys (fp32 [COUT,3,3,CIN], accumulated in place) += grad; db (fp32 [COUT]) += bias grad (same pass over dy)."""
    require_cuda(x, dy, dw_phys, db)
    if db is not None:
        assert db.dtype == torch.float32 and db.numel() == dy.shape[-1]
    B, H, W, CIN = x.shape
    _, Ho, Wo, COUT = dy.shape
    assert (Ho, Wo) == conv_out_hw(H, W, stride) and dy.shape[0] == B
    assert dw_phys.dtype == torch.float32 and dw_phys.is_contiguous() and tuple(dw_phys.shape) == (COUT, 3, 3, CIN)
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype
    mean, rstd = in_stats if in_stats is not None else (None, None)
    lib().call("omr_conv3x3_wgrad", dtype_code(x.dtype), ptr(x), ptr(dy), ptr(dw_phys), ptr(db), ptr(mean), ptr(rstd), B, H, W, CIN, COUT, stride[0], stride[1],
               Ho, Wo, cur_stream())


def conv3x3_bwd_fused_ok(x: Tensor, g: Tensor, stride=(1, 1)) -> bool:
    """Shapes omr_conv3x3_bwd_fused covers: bf16, stride 1, (COUT, CIN) in {(32, 32), (32, 16), (16, 16)}."""
    return (x.dtype == torch.bfloat16 and tuple(stride) == (1, 1) and x.dim() == 4 and (g.shape[-1], x.shape[-1]) in ((32, 32), (32, 16), (16, 16)))


def conv3x3_bwd_fused(g: Tensor, x: Tensor, w_flipped: Tensor, dw_phys: Tensor, db: Optional[Tensor], mask_input: bool, mask_scale: float = 1.0,
                      norm=None, xnorm=None) -> Tensor:
    """Data gradient, weight gradient and bias gradient of a stride-1 3x3 conv in one pass (bf16, <= 32 channels): returns
    dx = conv^T(g) [* (x > 0) * mask_scale]; dw_phys / db (fp32, accumulated in place).  norm = (y, mean, rstd, ws, slots, relu_mask,
    relu_scale): g is the gradient w.r.t. InstanceNorm(y) and the InstanceNorm backward (sums in ws, reduced by
    instnorm_reduce_sums) + the ReLU / dropout mask of y are applied on load.  xnorm = (mean, rstd, ws, slots) (16 -> 16 channels): the
    conv normalised x on load; returns dL/dxhat and fills the InstanceNorm-backward slots of ws (conv_stat_ws layout)."""
    require_cuda(g, x, w_flipped, dw_phys, db)
    B, H, W, CIN = x.shape
    COUT = g.shape[-1]
    assert conv3x3_bwd_fused_ok(x, g) and g.shape[:3] == x.shape[:3] and g.dtype == x.dtype and g.is_contiguous() and x.is_contiguous()
    assert tuple(w_flipped.shape) == (CIN, 3, 3, COUT) and w_flipped.dtype == x.dtype and w_flipped.is_contiguous()
    assert dw_phys.dtype == torch.float32 and dw_phys.is_contiguous() and tuple(dw_phys.shape) == (COUT, 3, 3, CIN)
    if db is not None:
        assert db.dtype == torch.float32 and db.numel() == COUT
    dx = torch.empty_like(x)
    if norm is None:
        ny = mean = rstd = ws = None
        slots, relu_mask, relu_scale = 0, False, 1.0
    else:
        ny, mean, rstd, ws, slots, relu_mask, relu_scale = norm
        require_cuda(ny, mean, rstd, ws)
        assert ny.shape == g.shape and ny.is_contiguous() and ny.dtype == g.dtype and ws.dtype == torch.float64
    xm = xr = xws = None
    xslots = 0
    if xnorm is not None:
        xm, xr, xws, xslots = xnorm
        require_cuda(xm, xr, xws)
        assert xws.dtype == torch.float64 and norm is None and not mask_input
    lib().call("omr_conv3x3_bwd_fused", ptr(g), ptr(x), ptr(w_flipped), ptr(dx), ptr(dw_phys), ptr(db), B, H, W, CIN, COUT, int(mask_input), float(mask_scale),
               ptr(ny), ptr(mean), ptr(rstd), ptr(ws), int(slots), int(relu_mask), float(relu_scale), ptr(xm), ptr(xr), ptr(xws), int(xslots), cur_stream())
    return dx


def conv3x3_bwd_fused_s2(g: Tensor, x: Tensor, w_flipped: Tensor, dw_phys: Tensor, db: Optional[Tensor], mean: Tensor, rstd: Tensor, ws: Tensor,
                         slots: int) -> Tensor:
    """One-pass backward of the stride-(2,2), 32 -> 32 channel conv that normalised x on load (bf16): returns dL/dxhat, accumulates
    dw_phys / db and fills the InstanceNorm-backward slots of ws (conv_stat_ws layout for x's shape)."""
    require_cuda(g, x, w_flipped, dw_phys, db, mean, rstd, ws)
    B, H, W, C = x.shape
    assert C == 32 and x.dtype == torch.bfloat16 and g.dtype == x.dtype and tuple(g.shape) == (B, (H + 1) // 2, (W + 1) // 2, 32)
    assert g.is_contiguous() and x.is_contiguous() and tuple(w_flipped.shape) == (32, 3, 3, 32) and w_flipped.is_contiguous() and w_flipped.dtype == x.dtype
    assert dw_phys.dtype == torch.float32 and dw_phys.is_contiguous() and tuple(dw_phys.shape) == (32, 3, 3, 32) and ws.dtype == torch.float64
    dx = torch.empty_like(x)
    lib().call("omr_conv3x3_bwd_fused_s2", ptr(g), ptr(x), ptr(w_flipped), ptr(dx), ptr(dw_phys), ptr(db), B, H, W, ptr(mean), ptr(rstd), ptr(ws), int(slots),
               cur_stream())
    return dx


def dwconv3x3(x: Tensor, w: Tensor, bias: Optional[Tensor], in_stats=None, out_mask: Optional[Tensor] = None, mask_scale: float = 1.0,
              flip: bool = False) -> Tensor:
    """Depthwise 3x3 on NHWC; w is [C,9] (= [C,1,3,3] storage)."""
    require_cuda(x, w, bias, out_mask)
    B, H, W, C = x.shape
    assert x.is_contiguous() and w.is_contiguous() and w.numel() == C * 9 and w.dtype == x.dtype
    y = torch.empty_like(x)
    mean, rstd = in_stats if in_stats is not None else (None, None)
    if out_mask is not None:
        assert out_mask.shape == x.shape and out_mask.is_contiguous()
    lib().call("omr_dwconv3x3", dtype_code(x.dtype), ptr(x), ptr(w), ptr(bias), ptr(y), ptr(mean), ptr(rstd), ptr(out_mask), float(mask_scale), B, H, W,
               C, int(flip), cur_stream())
    return y


def dwconv3x3_wgrad(x: Tensor, dy: Tensor, dw: Tensor, db: Optional[Tensor], in_stats=None) -> None:
    require_cuda(x, dy, dw, db)
    B, H, W, C = x.shape
    assert dy.shape == x.shape and dw.dtype == torch.float32 and dw.numel() == C * 9 and x.is_contiguous() and dy.is_contiguous()
    mean, rstd = in_stats if in_stats is not None else (None, None)
    lib().call("omr_dwconv3x3_wgrad", dtype_code(x.dtype), ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(mean), ptr(rstd), B, H, W, C, cur_stream())


# ------------------------------------------------------------------------------------------------ attention

def _bts(t: Tensor) -> Tuple[int, int]:
    assert t.dim() == 3 and t.stride(2) == 1, f"expected [B,rows,cols] with unit inner stride, got {tuple(t.shape)} / {t.stride()}"
    return t.stride(1), t.stride(0)


def attn_dropout_words(B: int, H: int, T: int, S: int, p: float, seed: int, device) -> Tensor:
    """The keep bits of the attention-probability dropout for (p, seed) in the kernels' word layout (omr_attn_dropout_words):
    int64 [omr_attn_dropout_words_count].  Generated once per (layer, step); forward and backward read the same buffer."""
    n = lib().query("omr_attn_dropout_words_count", B, H, T, S)
    out = torch.empty(n, dtype=torch.int64, device=device)
    lib().call("omr_attn_dropout_words", ptr(out), B, H, T, S, float(p), int(seed) & (2**64 - 1), cur_stream())
    return out


def attn_fwd(q: Tensor, k: Tensor, v: Tensor, nhead: int, *, causal: bool = False, window: int = -1, key_bias: Optional[Tensor] = None,
             blk_lq: Optional[Tensor] = None, blk_lkv: Optional[Tensor] = None, dropout_p: float = 0.0, seed: int = 0,
             drop_words: Optional[Tensor] = None):
    """q [B,T,d], k/v [B,S,d] (may be strided views of packed projections) -> (o [B,T,d], lse [B,H,T]).  dropout_p > 0: the keep
    bits come from drop_words (attn_dropout_words; generated here from (dropout_p, seed) when not given)."""
    require_cuda(q, k, v, key_bias, blk_lq, blk_lkv)
    B, T, d = q.shape
    S = k.shape[1]
    assert k.shape == (B, S, d) and v.shape == (B, S, d) and d % nhead == 0 and q.dtype == k.dtype == v.dtype
    hd = d // nhead
    o = torch.empty((B, T, d), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, nhead, T), dtype=torch.float32, device=q.device)
    if key_bias is not None:
        assert key_bias.dtype == torch.float32 and tuple(key_bias.shape) == (B, S) and key_bias.is_contiguous()
    for t in (blk_lq, blk_lkv):
        if t is not None:
            assert t.dtype == torch.int32 and t.numel() == B and t.is_contiguous()
    (ldq, bsq), (ldk, bsk), (ldv, bsv), (ldo, bso) = _bts(q), _bts(k), _bts(v), _bts(o)
    ws, nws = _attn_ws(B, nhead, T, S, hd, causal, False, q.device)
    if dropout_p > 0.0 and drop_words is None:
        drop_words = attn_dropout_words(B, nhead, T, S, dropout_p, seed, q.device)
    lib().call("omr_attn_fwd_ws", dtype_code(q.dtype), ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, nhead, T, S, hd,
               int(causal), int(window), ptr(key_bias), ptr(blk_lq), ptr(blk_lkv), float(dropout_p), int(seed) & (2**64 - 1), ptr(drop_words), ptr(ws), nws,
               cur_stream())
    return o, lse


def attn_fwd_split_partials(q: Tensor, k: Tensor, v: Tensor, nhead: int):
    """Decode attention (q [B,1,d]) stopped before its merge pass: -> (partials [B*H, nsplit, hd+2], nsplit); with nsplit == 1
    the first return value is the finished output [B,1,d] instead (omr_attn_fwd_split_partials)."""
    require_cuda(q, k, v)
    B, T, d = q.shape
    S, hd = k.shape[1], d // nhead
    assert T == 1
    o = torch.empty((B, T, d), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, nhead, T), dtype=torch.float32, device=q.device)
    n = lib().query("omr_attn_split_workspace_floats", B, nhead, T, S, hd)
    ws = torch.empty(max(n, 1), dtype=torch.float32, device=q.device)
    ns = ctypes.c_int(0)
    (ldq, bsq), (ldk, bsk), (ldv, bsv), (ldo, bso) = _bts(q), _bts(k), _bts(v), _bts(o)
    lib().call("omr_attn_fwd_split_partials", dtype_code(q.dtype), ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, nhead, T, S,
               hd, ptr(ws), n, ctypes.byref(ns), cur_stream())
    if ns.value <= 1:
        return o, 1
    return ws[:B * nhead * ns.value * (hd + 2)].view(B * nhead, ns.value, hd + 2), ns.value


def _attn_ws(B: int, H: int, T: int, S: int, hd: int, causal: bool, backward: bool, device):
    """Scratch for the key split of the query-per-lane attention kernels (include/omr_hip.h omr_attn_workspace_floats)."""
    n = lib().query("omr_attn_workspace_floats", B, H, T, S, hd, int(causal), int(backward))
    return (torch.empty(n, dtype=torch.float32, device=device), n) if n > 0 else (None, 0)


def attn_bwd(q, k, v, o, dout, lse, dq, dk, dv, nhead: int, *, causal=False, window=-1, key_bias=None, blk_lq=None, blk_lkv=None,
             dropout_p: float = 0.0, seed: int = 0, drop_words: Optional[Tensor] = None) -> None:
    """Writes dq [B,T,d], dk/dv [B,S,d] (views allowed, unit inner stride)."""
    require_cuda(q, k, v, o, dout, dq, dk, dv)
    B, T, d = q.shape
    S = k.shape[1]
    hd = d // nhead
    delta = torch.empty((B, nhead, T), dtype=torch.float32, device=q.device)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape and dout.shape == o.shape
    (ldq, bsq), (ldk, bsk), (ldv, bsv), (ldo, bso), (lddo, bsdo) = _bts(q), _bts(k), _bts(v), _bts(o), _bts(dout)
    (lddq, bsdq), (lddk, bsdk), (lddv, bsdv) = _bts(dq), _bts(dk), _bts(dv)
    ws, nws = _attn_ws(B, nhead, T, S, hd, causal, True, q.device)
    if dropout_p > 0.0 and drop_words is None:
        drop_words = attn_dropout_words(B, nhead, T, S, dropout_p, seed, q.device)
    lib().call("omr_attn_bwd_ws", dtype_code(q.dtype), ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), ldq, ldk,
               ldv, ldo, lddo, lddq, lddk, lddv, bsq, bsk, bsv, bso, bsdo, bsdq, bsdk, bsdv, B, nhead, T, S, hd, int(causal), int(window),
               ptr(key_bias), ptr(blk_lq), ptr(blk_lkv), float(dropout_p), int(seed) & (2**64 - 1), ptr(drop_words), ptr(ws), nws, cur_stream())


def attn_dropout_mask(B: int, H: int, T: int, S: int, p: float, seed: int, device) -> Tensor:
    """uint8 [B,H,T,S] keep-mask of the attention-probability dropout for (p, seed): test / debug entry."""
    out = torch.empty((B, H, T, S), dtype=torch.uint8, device=device)
    lib().call("omr_attn_dropout_mask", ptr(out), B, H, T, S, float(p), int(seed) & (2**64 - 1), cur_stream())
    return out


# ------------------------------------------------------------------------------------------------ loss

def ce_fwd(logits2d: Tensor, target: Tensor, V: int, pad_idx: int):
    """logits2d [M, ldv>=V]; target int64 [M] -> (loss fp32 [1], lse [M], acc2 fp64 [2])."""
    require_cuda(logits2d, target)
    M, cols, ldv = _rows2d(logits2d)
    assert cols >= V and target.numel() == M and target.dtype == torch.int64 and target.is_contiguous()
    lse = torch.empty(M, dtype=torch.float32, device=logits2d.device)
    acc2 = torch.empty(2, dtype=torch.float64, device=logits2d.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits2d.device)
    lib().call("omr_ce_fwd", dtype_code(logits2d.dtype), ptr(logits2d), ptr(target), ptr(lse), ptr(acc2), ptr(loss), M, V, ldv, pad_idx, cur_stream())
    return loss, lse, acc2


def ce_bwd(logits2d: Tensor, target: Tensor, lse: Tensor, acc2: Tensor, V: int, pad_idx: int, grad_scale: float = 1.0,
           grad_out: Optional[Tensor] = None) -> Tensor:
    require_cuda(logits2d, target, grad_out)
    if grad_out is not None:
        assert grad_out.dtype == torch.float32 and grad_out.numel() == 1
    M, cols, ldv = _rows2d(logits2d)
    dl = torch.empty((M, ldv), dtype=logits2d.dtype, device=logits2d.device)
    lib().call("omr_ce_bwd", dtype_code(logits2d.dtype), ptr(logits2d), ptr(target), ptr(lse), ptr(acc2), ptr(dl), M, V, ldv, pad_idx, float(grad_scale),
               ptr(grad_out), cur_stream())
    return dl[:, :cols]


def round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m
