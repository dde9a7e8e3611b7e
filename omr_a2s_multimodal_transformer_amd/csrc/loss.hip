// Cross-entropy over the vocabulary with ignore_index (gfx950).
// Reference: CrossEntropyLoss(ignore_index=PAD) on logits [B,V,T] (model.py:109,166) =
// mean over targets != PAD of  logsumexp_v(logits) - logits[target].
// Internally logits are row-major [M = B*T][ldv >= V] (the [B,V,T] view is a stride permutation).
#include "omr_common.h"
#include "omr_hip.h"

namespace {

// One workgroup per row: a single vectorised pass keeps a running (max, sum of exp) per thread (online softmax), then
// combines across the workgroup.  Only lse[row] is written: the loss itself is reduced by ce_finalize_kernel in a fixed
// order (no atomics: device-scope atomics on one address serialise at the memory side, and the sum stays deterministic).
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, float* __restrict__ lse, int V, long ldv) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    __shared__ float red_m[4], red_s[4];
    const long row = blockIdx.x;
    const T* lr = logits + row * ldv;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = -INFINITY, s = 0.f;
    const bool vec_ok = (ldv % VEC) == 0 && (((uintptr_t)logits) & 15) == 0;
    if (vec_ok) {
        for (int v0 = threadIdx.x * VEC; v0 < V; v0 += 256 * VEC) {
            const F f = *reinterpret_cast<const F*>(lr + v0);      // v0 + VEC <= ldv: the padding columns are readable
            float x[VEC], cm = -INFINITY;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { x[e] = (v0 + e < V) ? to_f32(f[e]) : -INFINITY; cm = fmaxf(cm, x[e]); }
            const float nm = fmaxf(mx, cm);
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) add += __expf(x[e] - nm);
            s = s * __expf(mx - nm) + add;
            mx = nm;
        }
    } else {
        for (int v = threadIdx.x; v < V; v += 256) {
            const float x = to_f32(lr[v]), nm = fmaxf(mx, x);
            s = s * __expf(mx - nm) + __expf(x - nm);
            mx = nm;
        }
    }
    const float wm = wave_max(mx);
    s = wave_sum(mx == -INFINITY ? 0.f : s * __expf(mx - wm));
    if (lane == 0) { red_m[wv] = wm; red_s[wv] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) t += red_m[i] == -INFINITY ? 0.f : red_s[i] * __expf(red_m[i] - m);
        lse[row] = m + logf(t);
    }
}

// acc[0] = sum over live rows of lse - logit[target], acc[1] = number of live rows, loss = mean.  One workgroup, fixed order.
template <typename T>
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const T* __restrict__ logits, const long* __restrict__ target, const float* __restrict__ lse,
                                                           double* __restrict__ acc, float* __restrict__ loss, long M, int V, long ldv, int pad_idx) {
    __shared__ double rs[16], rc[16];
    double s = 0.0, c = 0.0;
    for (long row = threadIdx.x; row < M; row += 1024) {
        const long t = target[row];
        if (t != pad_idx && t >= 0 && t < V) { s += (double)(lse[row] - to_f32(logits[row * ldv + t])); c += 1.0; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tc = 0.0;
        for (int i = 0; i < 16; ++i) { ts += rs[i]; tc += rc[i]; }
        acc[0] = ts; acc[1] = tc;
        loss[0] = (float)(ts / tc);
    }
}

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
    if (dtype == OMR_F32) {
        hipLaunchKernelGGL((ce_fwd_kernel<float>), (int)M, 256, 0, s, (const float*)logits, lse, V, ldv);
        hipLaunchKernelGGL((ce_finalize_kernel<float>), 1, 1024, 0, s, (const float*)logits, target, (const float*)lse, acc2, loss_out, M, V, ldv, pad_idx);
    } else if (dtype == OMR_BF16) {
        hipLaunchKernelGGL((ce_fwd_kernel<bf16>), (int)M, 256, 0, s, (const bf16*)logits, lse, V, ldv);
        hipLaunchKernelGGL((ce_finalize_kernel<bf16>), 1, 1024, 0, s, (const bf16*)logits, target, (const float*)lse, acc2, loss_out, M, V, ldv, pad_idx);
    } else return OMR_ERR_UNSUPPORTED;
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
