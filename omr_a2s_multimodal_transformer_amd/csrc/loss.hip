// Cross-entropy over the vocabulary with ignore_index (gfx950).
// Reference: CrossEntropyLoss(ignore_index=PAD) on logits [B,V,T] (model.py:109,166) =
// mean over targets != PAD of  logsumexp_v(logits) - logits[target].
// Internally logits are row-major [M = B*T][ldv >= V] (the [B,V,T] view is a stride permutation).
#include "omr_common.h"
#include "omr_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const long* __restrict__ target, float* __restrict__ lse,
                                                     double* __restrict__ acc /* [0]=sum loss, [1]=count */, int V, long ldv, int pad_idx) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const T* lr = logits + row * ldv;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, to_f32(lr[v]));
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) s += __expf(to_f32(lr[v]) - mx);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = mx + logf(red[0] + red[1] + red[2] + red[3]);
        lse[row] = l;
        long t = target[row];
        if (t != pad_idx && t >= 0 && t < V) {
            atomicAdd(&acc[0], (double)(l - to_f32(lr[t])));
            atomicAdd(&acc[1], 1.0);
        }
    }
}

__global__ void ce_finalize_kernel(const double* __restrict__ acc, float* __restrict__ loss) { loss[0] = (float)(acc[0] / acc[1]); }

// dlogits = (softmax - onehot) * gscale / count on rows whose target != PAD, else 0.
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const long* __restrict__ target, const float* __restrict__ lse,
                                                     const double* __restrict__ acc, T* __restrict__ dlogits, int V, long ldv, int pad_idx,
                                                     float gscale, const float* __restrict__ grad_out) {
    const long row = blockIdx.x;
    const long t = target[row];
    const bool live = (t != pad_idx && t >= 0 && t < V);
    const float l = lse[row];
    const float sc = live ? gscale * (grad_out ? grad_out[0] : 1.f) / (float)acc[1] : 0.f;
    const T* lr = logits + row * ldv;
    T* dr = dlogits + row * ldv;
    for (int v = threadIdx.x; v < ldv; v += 256) {
        float d = 0.f;
        if (live && v < V) d = (__expf(to_f32(lr[v]) - l) - (v == t ? 1.f : 0.f)) * sc;
        dr[v] = from_f32<T>(d);
    }
}

}  // namespace

extern "C" int omr_ce_fwd(int dtype, const void* logits, const long* target, float* lse, double* acc2, float* loss_out, long M, int V, long ldv,
                          int pad_idx, void* stream) {
    if (M <= 0 || V <= 0 || ldv < V) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(acc2, 0, 2 * sizeof(double), s) != hipSuccess) return OMR_ERR_LAUNCH;
    if (dtype == OMR_F32) hipLaunchKernelGGL((ce_fwd_kernel<float>), (int)M, 256, 0, s, (const float*)logits, target, lse, acc2, V, ldv, pad_idx);
    else if (dtype == OMR_BF16) hipLaunchKernelGGL((ce_fwd_kernel<bf16>), (int)M, 256, 0, s, (const bf16*)logits, target, lse, acc2, V, ldv, pad_idx);
    else return OMR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_finalize_kernel, 1, 1, 0, s, (const double*)acc2, loss_out);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_ce_bwd(int dtype, const void* logits, const long* target, const float* lse, const double* acc2, void* dlogits, long M, int V,
                          long ldv, int pad_idx, float grad_scale, const float* grad_out, void* stream) {
    if (M <= 0 || V <= 0 || ldv < V) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMR_F32) hipLaunchKernelGGL((ce_bwd_kernel<float>), (int)M, 256, 0, s, (const float*)logits, target, lse, acc2, (float*)dlogits, V, ldv, pad_idx, grad_scale, grad_out);
    else if (dtype == OMR_BF16) hipLaunchKernelGGL((ce_bwd_kernel<bf16>), (int)M, 256, 0, s, (const bf16*)logits, target, lse, acc2, (bf16*)dlogits, V, ldv, pad_idx, grad_scale, grad_out);
    else return OMR_ERR_UNSUPPORTED;
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
