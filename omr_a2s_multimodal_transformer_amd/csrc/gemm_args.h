// Arguments of the GEMM kernel (gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

struct GemmArgs {
    const void* A; const void* B; void* C; const float* bias;
    int M, N, K; long lda, ldb, ldc;
    int relu, accum, atomic, ksplit_len;
    float* colsum_a;   // transA only: colsum_a[m] += sum_k A[k][m] (bias gradient of a linear layer), fused into the dW GEMM
    int mt, nt, chunk, total;   // tile grid and XCD chunking (filled by the launchers)
    unsigned drop_thresh; float drop_scale; unsigned long long drop_seed;   // fused dropout on the bf16/f32 output (0 = off)
    // Row-group view of the weight-side operand (0 = off): logical row i lives at physical row (i / grp) * grp_stride +
    // grp_base + i % grp.  grp_operand: 1 = rows of B / entries of bias (index n), 2 = reduction rows of a transposed B
    // (index k), 3 = rows of C / entries of colsum_a (index m).  Lets ONE GEMM run over the K|V rows of all decoder layers'
    // packed in_proj matrices where they lie in the flat parameter buffer ([Wq;Wk;Wv] blocks back to back).
    int grp, grp_stride, grp_base, grp_operand;
    // fp8 operands (omr_gemm_fp8): C = (A8 . B8^T) * scale_a[m] * scale_b[n] + bias -- per-row quantisation scales of both operands
    const float* scale_a; const float* scale_b;
};

