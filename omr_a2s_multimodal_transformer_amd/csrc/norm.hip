// Normalisation kernels (gfx950), all statistics in fp32/fp64:
//   InstanceNorm2d(eps=1e-3, affine=False, no running stats) on NHWC maps  -- encoder.py:151-156,210-215
//   residual + LayerNorm(eps=1e-5) of the post-norm decoder layer          -- torch nn/modules/transformer.py:1146-1154
#include "omr_common.h"
#include "omr_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// InstanceNorm statistics over x[B][HW][C] (NHWC): per (b, c) mean and 1/sqrt(biased var + eps).
// DETERMINISTIC two-stage reduction (no atomics): stage 1 -- each block reduces a slab of pixels for all channels (per-thread
// fp32 partials over a short run, combined over the block's threads in a fixed order in fp64) and stores the result into
// its own slot ws[b][slot][c][2] = (sum, sum of squares); stage 2 adds an image's slots in index order and finalises.
// The same slot layout is written by the conv kernels' fused statistics epilogue (conv3x3_mfma.h).
// The apply is fused into the consumer's tile load (conv3 / depthwise conv3), never materialised.
// Thread t owns channel group (t % (C/VEC)) and walks pixels with 16-byte loads (fully coalesced: NHWC rows are contiguous).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, double* __restrict__ ws, long HW, int C,
                                                               int pix_per_block) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [nphase][C][2]
    const int b = blockIdx.y;
    const int ncg = C / VEC;                       // channel groups; divides blockDim for the encoder's widths
    const int cg = threadIdx.x % ncg, phase = threadIdx.x / ncg, nphase = blockDim.x / ncg;
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = p0 + pix_per_block < HW ? p0 + pix_per_block : HW;
    const long base = (long)b * HW * C;
    float s[VEC], q[VEC], mu[VEC], rs[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        s[e] = q[e] = 0.f;
        mu[e] = BWD ? mean[(long)b * C + cg * VEC + e] : 0.f;
        rs[e] = BWD ? rstd[(long)b * C + cg * VEC + e] : 1.f;
    }
    for (long p = p0 + phase; p < p1; p += nphase) {
        const F xv = *reinterpret_cast<const F*>(x + base + p * C + cg * VEC);
        if (BWD) {
            const F gv = *reinterpret_cast<const F*>(g + base + p * C + cg * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float gg = to_f32(gv[e]);
                s[e] += gg;
                q[e] += gg * ((to_f32(xv[e]) - mu[e]) * rs[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float v = to_f32(xv[e]);
                s[e] += v;
                q[e] += v * v;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        red[(phase * C + cg * VEC + e) * 2 + 0] = s[e];
        red[(phase * C + cg * VEC + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    double* out = ws + ((long)b * gridDim.x + blockIdx.x) * C * 2;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        double acc = 0.0;
        for (int ph = 0; ph < nphase; ++ph) acc += (double)red[ph * 2 * C + i];
        out[i] = acc;
    }
}
// Stage 2.  One workgroup per image: value v = (channel, {sum | sum of squares}) of slot k lives at ws[b][k][v]; the 256 threads
// are dealt out as (v, part) so that `parts` threads share a value, each adding the slots k = part, part + parts, ... in that
// order, and the partial sums are then added in part order -- fixed order, no atomics, and no thread walks more than
// slots / parts slots (a single thread per value was a 20-27 us latency chain per launch).
// FINAL: mean / rstd; else: the compact sums [b][C][2] that the backward apply kernel reads.
template <bool FINAL>
__global__ __launch_bounds__(256) void instnorm_reduce_slots_kernel(const double* __restrict__ ws, int slots, int C, float* __restrict__ mean,
                                                                    float* __restrict__ rstd, double* __restrict__ compact, double inv_hw, float eps) {
    __shared__ double red[512];
    const int b = blockIdx.x, NV = 2 * C, tid = threadIdx.x;
    const double* base = ws + (long)b * slots * NV;
    for (int v0 = 0; v0 < NV; v0 += 256) {                       // NV <= 256 in one round (C <= 128), two rounds for C = 256
        const int nv = NV - v0 < 256 ? NV - v0 : 256;            // values of this round (a power of two for the encoder's widths)
        const int parts = 256 / nv > 0 ? 256 / nv : 1;
        const int v = tid % nv, part = tid / nv;
        double acc = 0.0;
        if (part < parts)
            for (int k = part; k < slots; k += parts) acc += base[(long)k * NV + v0 + v];
        __syncthreads();
        if (part < parts) red[part * nv + v] = acc;
        __syncthreads();
        if (tid < nv) {
            double s = 0.0;
            for (int q = 0; q < parts; ++q) s += red[q * nv + tid];
            red[256 + tid] = s;
        }
        __syncthreads();
        if (FINAL) {
            if (tid < nv / 2) {                                   // channel c = (v0 + 2 tid) / 2
                const double m = red[256 + 2 * tid] * inv_hw;
                double var = red[256 + 2 * tid + 1] * inv_hw - m * m;
                if (var < 0) var = 0;
                const long o = (long)b * C + v0 / 2 + tid;
                mean[o] = (float)m;
                rstd[o] = (float)(1.0 / sqrt(var + (double)eps));
            }
        } else if (tid < nv) {
            compact[(long)b * NV + v0 + tid] = red[256 + tid];
        }
    }
}

// InstanceNorm backward.  With xhat = (x - mean) * rstd and g = dL/dxhat:
//   dx = rstd * (g - mean_hw(g) - xhat * mean_hw(g * xhat))
// Stage 1 accumulates sum(g), sum(g*xhat) per (b,c) in fp64; stage 2 applies, and optionally folds
// in the ReLU(+dropout) backward of the producer of x: dx *= (x > 0) * relu_scale.
// Apply: thread owns one channel group (statistics loaded once into registers) and walks its pixels: no index divisions.
template <typename T>
__global__ __launch_bounds__(256) void instnorm_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const double* __restrict__ ws, T* __restrict__ dx, long HW,
                                                                 int C, int pix_per_block, float inv_hw, int relu_mask, float relu_scale) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    const int b = blockIdx.y;
    const int ncg = C / VEC;
    const int cg = threadIdx.x % ncg, phase = threadIdx.x / ncg, nphase = blockDim.x / ncg;
    float mu[VEC], rs[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const long bc = (long)b * C + cg * VEC + e;
        mu[e] = mean[bc]; rs[e] = rstd[bc];
        s1[e] = (float)(ws[2 * bc] * (double)inv_hw); s2[e] = (float)(ws[2 * bc + 1] * (double)inv_hw);
    }
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = p0 + pix_per_block < HW ? p0 + pix_per_block : HW;
    const long base = (long)b * HW * C + cg * VEC;
#pragma unroll 4
    for (long p = p0 + phase; p < p1; p += nphase) {
        const F gv = *reinterpret_cast<const F*>(g + base + p * C), xv = *reinterpret_cast<const F*>(x + base + p * C);
        F o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xf = to_f32(xv[e]);
            float d = rs[e] * (to_f32(gv[e]) - s1[e] - (xf - mu[e]) * rs[e] * s2[e]);
            if (relu_mask) d = xf > 0.f ? d * relu_scale : 0.f;
            o[e] = from_f32<T>(d);
        }
        *reinterpret_cast<F*>(dx + base + p * C) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// out = LayerNorm(dropout(x) + res) * gamma + beta.  One wave per row; lane l owns the PER = d/64 CONSECUTIVE elements
// [l*PER, (l+1)*PER) so every tensor is touched with one vector load/store per lane per row.  Saves mean and rstd.
// The sublayer dropout that precedes every post-norm residual (torch nn/modules/transformer.py:1146-1154) is fused:
// drop_thresh != 0 applies the counter-based mask of omr_dropout (same seed/index convention, element index row*d + c)
// to x; the backward regenerates it.
template <typename T, int PER>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
                                                         float* __restrict__ rstd, long M, float eps, uint32_t drop_thresh, float drop_scale,
                                                         uint64_t drop_seed) {
    typedef __attribute__((ext_vector_type(PER))) T VT;
    constexpr int d = PER * 64;
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const VT xv = *reinterpret_cast<const VT*>(x + row * d + lane * PER);
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = to_f32(xv[i]);
    if (drop_thresh) {
        // the unfused path rounds the dropped value to T before the add: keep that rounding so both paths agree to the bit
#pragma unroll
        for (int i = 0; i < PER; ++i)
            v[i] = drop_keep(drop_seed, (uint64_t)(row * d + lane * PER + i), drop_thresh) ? to_f32(from_f32<T>(v[i] * drop_scale)) : 0.f;
    }
    if (res) {
        const VT rv = *reinterpret_cast<const VT*>(res + row * d + lane * PER);
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] += to_f32(rv[i]);
    }
    float mu, rs;
    ln_row<PER>(v, gamma, beta, lane, eps, mu, rs);
    VT o;
#pragma unroll
    for (int i = 0; i < PER; ++i) o[i] = from_f32<T>(v[i]);
    *reinterpret_cast<VT*>(out + row * d + lane * PER) = o;
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// Backward: s = dropout(x) + res, xhat = (s - mean) * rstd, gh = dy * gamma
//   ds = rstd * (gh - mean_d(gh) - xhat * mean_d(gh * xhat))      (gradient of res; of x too when there is no dropout)
//   dx = keep ? ds / (1-p) : 0                                     (second output, only with the fused dropout)
//   dgamma[c] += sum_rows dy * xhat ; dbeta[c] += sum_rows dy     (register partials per wave -> LDS -> one atomic per column per block)
template <typename T, int PER>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, T* __restrict__ ds, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, long M, int rows_per_block, uint32_t drop_thresh, float drop_scale,
                                                         uint64_t drop_seed, T* __restrict__ dx) {
    typedef __attribute__((ext_vector_type(PER))) T VT;
    typedef __attribute__((ext_vector_type(PER))) float VF;
    constexpr int d = PER * 64;
    __shared__ float cg[d], cb[d];
    for (int i = threadIdx.x; i < d; i += blockDim.x) { cg[i] = 0.f; cb[i] = 0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const VF gm = *reinterpret_cast<const VF*>(gamma + lane * PER);
    float ag[PER], ab[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) ag[i] = ab[i] = 0.f;
    // the next row's operands are requested before this row's math (a wave walks its rows one after another)
    const long rend = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
    VT xn, gn, rn;
    float mun = 0.f, rsn = 1.f;
    auto fetch = [&](long row) {
        if (row < rend) {
            mun = mean[row]; rsn = rstd[row];
            xn = *reinterpret_cast<const VT*>(x + row * d + lane * PER);
            gn = *reinterpret_cast<const VT*>(dy + row * d + lane * PER);
            if (res) rn = *reinterpret_cast<const VT*>(res + row * d + lane * PER);
        }
    };
    fetch(r0 + wv);
    for (long row = r0 + wv; row < rend; row += nw) {
        const float mu = mun, rs = rsn;
        const VT xv = xn, gv = gn, rv = rn;
        fetch(row + nw);
        float sv[PER];
        bool keep[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) { sv[i] = to_f32(xv[i]); keep[i] = true; }
        if (drop_thresh) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                keep[i] = drop_keep(drop_seed, (uint64_t)(row * d + lane * PER + i), drop_thresh);
                sv[i] = keep[i] ? to_f32(from_f32<T>(sv[i] * drop_scale)) : 0.f;
            }
        }
        if (res) {
#pragma unroll
            for (int i = 0; i < PER; ++i) sv[i] += to_f32(rv[i]);
        }
        float xh[PER], gh[PER], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const float g = to_f32(gv[i]);
            xh[i] = (sv[i] - mu) * rs;
            gh[i] = g * gm[i];
            ag[i] += g * xh[i];
            ab[i] += g;
            s1 += gh[i]; s2 += gh[i] * xh[i];
        }
        s1 = wave_sum(s1) * (1.f / d); s2 = wave_sum(s2) * (1.f / d);
        VT o;
#pragma unroll
        for (int i = 0; i < PER; ++i) o[i] = from_f32<T>(rs * (gh[i] - s1 - xh[i] * s2));
        *reinterpret_cast<VT*>(ds + row * d + lane * PER) = o;
        if (drop_thresh) {
            VT od;
#pragma unroll
            for (int i = 0; i < PER; ++i) od[i] = keep[i] ? from_f32<T>(to_f32(o[i]) * drop_scale) : from_f32<T>(0.f);
            *reinterpret_cast<VT*>(dx + row * d + lane * PER) = od;
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) { atomicAdd(&cg[lane * PER + i], ag[i]); atomicAdd(&cb[lane * PER + i], ab[i]); }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        atomicAdd(&dgamma[c], cg[c]);
        atomicAdd(&dbeta[c], cb[c]);
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                         \
    if ((dtype) == OMR_F32) { typedef float T; CALL; }  \
    else if ((dtype) == OMR_BF16) { typedef bf16 T; CALL; } \
    else return OMR_ERR_UNSUPPORTED;

// Pixels per workgroup of the statistics passes: 2048 on the big maps, fewer on the small DSC maps so that at least ~1000
// workgroups are in flight (16 x 256 maps with 2048 pixels per workgroup left 3/4 of the CUs idle).
static int stat_pixels_per_block(long HW, int B) {
    long ppb = 2048;
    while (ppb > 64 && cdiv(HW, ppb) * (long)B < 1024) ppb /= 2;
    return (int)ppb;
}
static size_t partial_lds_bytes(int dtype) { return (size_t)256 * (dtype == OMR_BF16 ? 8 : 4) * 2 * sizeof(float); }   // [nphase][C][2] with nphase * C/VEC = 256

/* slots per image of the stand-alone statistics passes (omr_instnorm_stats / omr_instnorm_bwd) */
extern "C" int omr_instnorm_slots(int B, long HW) { return (B <= 0 || HW <= 0) ? OMR_ERR_ARG : cdiv(HW, stat_pixels_per_block(HW, B)); }
/* bytes of a statistics workspace with `slots` slots per image: partials [B][slots][C][2] + compact sums [B][C][2], fp64 */
extern "C" long omr_instnorm_workspace_bytes(int B, int C, int slots) { return ((long)B * slots * C * 2 + (long)B * C * 2) * sizeof(double); }

extern "C" int omr_instnorm_stats(int dtype, const void* x, float* mean, float* rstd, int B, long HW, int C, float eps, void* workspace,
                                  void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || !workspace) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (C % vec || 256 % (C / vec)) return OMR_ERR_UNSUPPORTED;
    int ppb = stat_pixels_per_block(HW, B);
    dim3 grid(cdiv(HW, ppb), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL((instnorm_partial_kernel<T, false>), grid, 256, partial_lds_bytes(dtype), s, (const T*)x, (const T*)nullptr,
                                         (const float*)nullptr, (const float*)nullptr, (double*)workspace, HW, C, ppb));
    hipLaunchKernelGGL((instnorm_reduce_slots_kernel<true>), B, 256, 0, s, (const double*)workspace, (int)grid.x, C, mean, rstd, (double*)nullptr, 1.0 / (double)HW, eps);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

/* mean / rstd from the fp64 {sum, sum of squares} slots that a producer kernel filled (omr_conv3x3_fwd stat_mode 1) */
extern "C" int omr_instnorm_finalize(const void* workspace, int slots, float* mean, float* rstd, int B, long HW, int C, float eps, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || slots < 1 || !workspace) return OMR_ERR_ARG;
    hipLaunchKernelGGL((instnorm_reduce_slots_kernel<true>), B, 256, 0, (hipStream_t)stream, (const double*)workspace, slots, C, mean, rstd, (double*)nullptr, 1.0 / (double)HW, eps);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

/* the image-wise sums of the slots a producer kernel filled (omr_conv3x3_fwd stat_mode 2 / 4), added in slot order into the
 * compact region [B][C][2] behind them: what omr_conv3x3_fwd stat_mode 5 (and omr_instnorm_bwd_apply) read */
extern "C" int omr_instnorm_reduce_sums(void* workspace, int slots, int B, int C, void* stream) {
    if (B <= 0 || C <= 0 || slots < 1 || !workspace) return OMR_ERR_ARG;
    double* ws = (double*)workspace;
    hipLaunchKernelGGL((instnorm_reduce_slots_kernel<false>), B, 256, 0, (hipStream_t)stream, (const double*)ws, slots, C, (float*)nullptr, (float*)nullptr,
                       ws + (long)B * slots * C * 2, 0.0, 0.f);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

static int launch_bwd_apply(int dtype, const void* dxhat, const void* x, const float* mean, const float* rstd, void* dx, int B, long HW, int C,
                            int relu_mask, float relu_scale, double* ws, int slots, hipStream_t s) {
    double* compact = ws + (long)B * slots * C * 2;
    hipLaunchKernelGGL((instnorm_reduce_slots_kernel<false>), B, 256, 0, s, (const double*)ws, slots, C, (float*)nullptr, (float*)nullptr, compact, 0.0, 0.f);
    int ppb2 = 1024;
    dim3 grid2(cdiv(HW, ppb2), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T>), grid2, 256, 0, s, (const T*)dxhat, (const T*)x, mean, rstd,
                                         (const double*)compact, (T*)dx, HW, C, ppb2, (float)(1.0 / (double)HW), relu_mask, relu_scale));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

/* apply step of the InstanceNorm backward with {sum g, sum g*xhat} already in the slots of `workspace`
 * (omr_conv3x3_fwd stat_mode 2 fuses that reduction into the data-gradient conv that produces dxhat) */
extern "C" int omr_instnorm_bwd_apply(int dtype, const void* dxhat, const void* x, const float* mean, const float* rstd, void* dx, int B, long HW,
                                      int C, int relu_mask, float relu_scale, void* workspace, int slots, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || slots < 1 || !workspace) return OMR_ERR_ARG;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (C % vec || 256 % (C / vec)) return OMR_ERR_UNSUPPORTED;
    return launch_bwd_apply(dtype, dxhat, x, mean, rstd, dx, B, HW, C, relu_mask, relu_scale, (double*)workspace, slots, (hipStream_t)stream);
}

extern "C" int omr_instnorm_bwd(int dtype, const void* dxhat, const void* x, const float* mean, const float* rstd, void* dx, int B, long HW,
                                int C, int relu_mask, float relu_scale, void* workspace, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || !workspace) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (C % vec || 256 % (C / vec)) return OMR_ERR_UNSUPPORTED;
    int ppb = stat_pixels_per_block(HW, B);
    dim3 grid(cdiv(HW, ppb), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL((instnorm_partial_kernel<T, true>), grid, 256, partial_lds_bytes(dtype), s, (const T*)x, (const T*)dxhat, mean, rstd,
                                         (double*)workspace, HW, C, ppb));
    return launch_bwd_apply(dtype, dxhat, x, mean, rstd, dx, B, HW, C, relu_mask, relu_scale, (double*)workspace, (int)grid.x, s);
}

#define LN_DISPATCH_PER(KERNEL, ...)                                                                       \
    switch (d / 64) {                                                                                      \
        case 2: hipLaunchKernelGGL((KERNEL<T, 2>), __VA_ARGS__); break;                                    \
        case 4: hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__); break;                                    \
        case 8: hipLaunchKernelGGL((KERNEL<T, 8>), __VA_ARGS__); break;                                    \
        default: return OMR_ERR_UNSUPPORTED;                                                               \
    }

extern "C" int omr_add_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* out, float* mean,
                                     float* rstd, long M, int d, float eps, float drop_p, unsigned long long drop_seed, void* stream) {
    if (M <= 0 || d <= 0 || d % 64 || drop_p < 0.f || drop_p >= 1.f) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int grid = cdiv(M, 4);
    const uint32_t thresh = OMR_DROP_THRESH16(drop_p);
    const float scale = 1.f / (1.f - drop_p);
    DISPATCH_T(dtype, { LN_DISPATCH_PER(add_ln_fwd_kernel, grid, 256, 0, s, (const T*)x, (const T*)res, gamma, beta, (T*)out, mean, rstd, M, eps, thresh, scale,
                                        (uint64_t)drop_seed) });
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_add_layernorm_bwd(int dtype, const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                                     const float* rstd, void* ds, float* dgamma, float* dbeta, long M, int d, float drop_p,
                                     unsigned long long drop_seed, void* dx, void* stream) {
    if (M <= 0 || d <= 0 || d % 64 || drop_p < 0.f || drop_p >= 1.f) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rpb = 64;                        // fewer, longer workgroups: each ends with 2 d atomics onto the same 16 cache lines
    int grid = cdiv(M, rpb);
    const uint32_t thresh = OMR_DROP_THRESH16(drop_p);
    const float scale = 1.f / (1.f - drop_p);
    if (thresh && !dx) return OMR_ERR_ARG;
    DISPATCH_T(dtype, { LN_DISPATCH_PER(add_ln_bwd_kernel, grid, 256, 0, s, (const T*)dy, (const T*)x, (const T*)res, gamma, mean, rstd, (T*)ds, dgamma, dbeta, M, rpb,
                                        thresh, scale, (uint64_t)drop_seed, (T*)dx) });
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
