"""The DMA-fed tall GEMM (csrc/gemm_tall.hip): C[M][256] = A[M][K] . B[K][256] for at least 192 row tiles of 256 -- the
data gradient of the all-layer K|V projection (functional.FusedCrossKVFn.backward; decoder.py:86-95).  Against torch fp32 on
the same bf16 operands: plain and through the reduction-side row-group view, with a ragged last row tile, strided operands,
one and several k-tiles."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF)


def _check(got, ref, kdim):
    err = (got.float().cpu() - ref).abs().max().item()
    tol = 2 ** -8 * max(1.0, ref.abs().max().item()) + 1e-3 * math.sqrt(kdim)
    assert err <= tol, f"max err {err} > {tol}"


@pytest.mark.parametrize("M,Kd", [(192 * 256, 64), (192 * 256 + 37, 192), (200 * 256 - 255, 640), (512 * 256, 128)])
def test_tall_gemm_matches_torch(M, Kd):
    a = _rnd((M, Kd), 1)
    b = _rnd((Kd, 256), 2, 1 / math.sqrt(Kd))
    out = K.gemm(a.to(DEV), b.to(DEV), trans_b=True)
    assert out.shape == (M, 256) and out.dtype == BF
    # rows sampled over the whole height (the fp32 product of everything is slow on the host), always with the ragged tail
    rows = torch.cat([torch.arange(0, M, 997), torch.arange(max(0, M - 300), M)])
    _check(out[rows.to(DEV)], a[rows].float() @ b.float(), Kd)


def test_tall_gemm_strided_operands_and_row_tail_untouched():
    """lda / ldb / ldc larger than the logical widths; rows of the output buffer beyond M stay as they were."""
    M, Kd = 192 * 256 + 5, 128
    a_buf, b_buf = _rnd((M, Kd + 64), 3), _rnd((Kd, 256 + 8), 4, 0.1)
    a, b = a_buf[:, 8:8 + Kd], b_buf[:, 8:8 + 256]
    out_buf = torch.full((M + 300, 256 + 16), 7.0, dtype=BF, device=DEV)
    out = out_buf[:M, 16:]
    ag, bg = a_buf.to(DEV)[:, 8:8 + Kd], b_buf.to(DEV)[:, 8:8 + 256]
    K.gemm(ag, bg, trans_b=True, out=out)
    rows = torch.cat([torch.arange(0, M, 1499), torch.arange(M - 270, M)])
    _check(out[rows.to(DEV)], a[rows].float() @ b.float(), Kd)
    assert float((out_buf[M:].float() - 7.0).abs().max()) == 0.0
    assert float((out_buf[:M, :16].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("L,d", [(2, 256), (6, 256)])
def test_tall_gemm_through_reduction_row_groups(L, d):
    """The K|V rows of L packed in_proj matrices ([Wq;Wk;Wv] back to back) as the reduction side -- what
    FusedCrossKVFn.backward hands over -- against explicit per-layer slices."""
    M = 192 * 256 + 11
    w_all = _rnd((L * 3 * d, d), 5, 1 / math.sqrt(d))
    g = _rnd((M, L * 2 * d), 6)
    kv_rows = torch.cat([torch.arange(l * 3 * d + d, (l + 1) * 3 * d) for l in range(L)])
    dx = torch.empty((M, d), dtype=BF, device=DEV)
    K.gemm_row_groups(g.to(DEV), w_all.to(DEV), dx, M, d, L * 2 * d, trans_b=True, group=(2 * d, 3 * d, d, 2))
    rows = torch.cat([torch.arange(0, M, 1201), torch.arange(M - 280, M)])
    _check(dx[rows.to(DEV)], g[rows].float() @ w_all[kv_rows].float(), L * 2 * d)
