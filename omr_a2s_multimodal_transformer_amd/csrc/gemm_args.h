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
};

