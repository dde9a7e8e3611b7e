// HBM-bound elementwise / gather / reduction kernels (gfx950): 16-byte vector access per lane,
// grid-stride loops capped at 2048 blocks (256 CUs x 8).
#include "omr_common.h"
#include "omr_hip.h"

namespace {

constexpr int EW_BLOCK = 256;
inline int ew_grid(long nvec) { long g = (nvec + EW_BLOCK - 1) / EW_BLOCK; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

template <typename T> struct Vec {
    typedef typename Frag<T>::type type;
    static constexpr int N = Frag<T>::N;
};

// ---------------------------------------------------------------- cast
template <typename TS, typename TD> __global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        d[i] = from_f32<TD>(to_f32(s[i]));
}

// ---------------------------------------------------------------- out = a + b
template <typename T> __global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long n) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    long nv = n / N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        V x = reinterpret_cast<const V*>(a)[i], y = reinterpret_cast<const V*>(b)[i], z;
#pragma unroll
        for (int e = 0; e < N; ++e) z[e] = from_f32<T>(to_f32(x[e]) + to_f32(y[e]));
        reinterpret_cast<V*>(o)[i] = z;
    }
    if (blockIdx.x == 0)
        for (long i = nv * N + threadIdx.x; i < n; i += blockDim.x) o[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
}

// ---------------------------------------------------------------- dx = dy * (y > 0) * scale   (ReLU [+dropout] backward)
template <typename T> __global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long n, float scale) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    long nv = n / N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        V g = reinterpret_cast<const V*>(dy)[i], yy = reinterpret_cast<const V*>(y)[i], z;
#pragma unroll
        for (int e = 0; e < N; ++e) z[e] = from_f32<T>(to_f32(yy[e]) > 0.f ? to_f32(g[e]) * scale : 0.f);
        reinterpret_cast<V*>(dx)[i] = z;
    }
    if (blockIdx.x == 0)
        for (long i = nv * N + threadIdx.x; i < n; i += blockDim.x)
            dx[i] = from_f32<T>(to_f32(y[i]) > 0.f ? to_f32(dy[i]) * scale : 0.f);
}

// ---------------------------------------------------------------- dropout (elementwise, or per (b, channel) for Dropout2d on NHWC)
// out = keep ? x / (1-p) : 0.   channel_mode: the mask index is b*C + c (nn.Dropout2d zeroes whole
// channels, encoder.py:99); otherwise the flat element index.
template <typename T, bool CH> __global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ o, long n, uint32_t thresh, float scale,
                                                              uint64_t seed, long per_sample, int C) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    if (CH) {
        // one keep/drop decision per (sample, channel): blockIdx.y = sample, 32-bit index math inside the sample
        const long sbase = (long)blockIdx.y * per_sample;
        const uint32_t nv = (uint32_t)(per_sample / N);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
            const uint32_t c0 = (i * N) % (uint32_t)C;
            const V v = reinterpret_cast<const V*>(x + sbase)[i];
            V r;
#pragma unroll
            for (int e = 0; e < N; ++e)
                r[e] = drop_keep(seed, (uint64_t)blockIdx.y * C + c0 + e, thresh) ? from_f32<T>(to_f32(v[e]) * scale) : from_f32<T>(0.f);
            reinterpret_cast<V*>(o + sbase)[i] = r;
        }
        return;
    }
    const long nv = n / N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        const V v = reinterpret_cast<const V*>(x)[i];
        V r;
#pragma unroll
        for (int e = 0; e < N; ++e) r[e] = drop_keep(seed, (uint64_t)(i * N + e), thresh) ? from_f32<T>(to_f32(v[e]) * scale) : from_f32<T>(0.f);
        reinterpret_cast<V*>(o)[i] = r;
    }
    if (blockIdx.x == 0)
        for (long k = nv * N + threadIdx.x; k < n; k += blockDim.x)
            o[k] = drop_keep(seed, (uint64_t)k, thresh) ? from_f32<T>(to_f32(x[k]) * scale) : from_f32<T>(0.f);
}

// ---------------------------------------------------------------- embedding gather + 1-D positional encoding
// out[m, :] = table[tok[m], :] + pe[m % T, :]      (decoder.py:124; no sqrt(d) scaling)
template <typename T> __global__ void embed_pe_kernel(const long* __restrict__ tok, const T* __restrict__ table, const float* __restrict__ pe,
                                                      T* __restrict__ out, long M, int Tlen, int d, int vocab) {
    long total = M * d;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / d; int c = (int)(i % d);
        long t = tok[m];
        float e = (t >= 0 && t < vocab) ? to_f32(table[t * d + c]) : 0.f;
        out[i] = from_f32<T>(e + pe[(m % Tlen) * d + c]);
    }
}
// dTable[tok[m], :] += dOut[m, :]   (fp32 atomics; the PAD row gets no gradient: nn.Embedding padding_idx)
template <typename T> __global__ void embed_bwd_kernel(const long* __restrict__ tok, const T* __restrict__ dout, float* __restrict__ dtable,
                                                       long M, int d, int pad_idx, int vocab) {
    long total = M * d;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / d; int c = (int)(i % d);
        long t = tok[m];
        if (t == pad_idx || t < 0 || t >= vocab) continue;
        atomicAdd(&dtable[t * d + c], to_f32(dout[i]));
    }
}

// ---------------------------------------------------------------- 2-D positional encoding on NHWC feature maps
// out[b,i,j,c] = x[b,i,j,c] + pe[i,j,c]  with pe stored [maxh][maxw][C]   (model.py:45-47 then flatten :147)
template <typename T> __global__ void add_pe2d_kernel(const T* __restrict__ x, const float* __restrict__ pe, T* __restrict__ out,
                                                      int B, int h, int w, int C, int maxw) {
    long total = (long)B * h * w * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C); long p = i / C; int j = (int)(p % w); long q = p / w; int ii = (int)(q % h);
        out[i] = from_f32<T>(to_f32(x[i]) + pe[((long)ii * maxw + j) * C + c]);
    }
}
// the same with 16-byte fragments (C a multiple of the fragment width, 16-byte aligned tensors): one index decode per fragment
template <typename T> __global__ void add_pe2d_vec_kernel(const T* __restrict__ x, const float* __restrict__ pe, T* __restrict__ out,
                                                          int B, int h, int w, int C, int maxw) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    const int cv = C / VEC;
    const long total = (long)B * h * w * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC; const long p = i / cv; const int j = (int)(p % w); const long q = p / w; const int ii = (int)(q % h);
        const F xv = *reinterpret_cast<const F*>(x + p * C + c);
        const float* pp = pe + ((long)ii * maxw + j) * C + c;
        F o;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[e + k] = from_f32<T>(to_f32(xv[e + k]) + pv[k]);
        }
        *reinterpret_cast<F*>(out + p * C + c) = o;
    }
}

// ---------------------------------------------------------------- column sums:  db[n] += sum_m dY[m, n]   (bias gradients)
// Block = 64 columns x 4 row phases over a slab of rows; LDS combine, one fp32 atomic per column per block.
template <typename T> __global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ db, long M, int N, long ld, int rows_per_block) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float s = 0.f;
    if (col < N)
        for (long r = r0 + ph; r < r1; r += 4) s += to_f32(dy[r * ld + col]);
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && col < N) atomicAdd(&db[col], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// ---------------------------------------------------------------- fused Adam over one flat buffer
// torch optim/adam.py:347 single-tensor math (model.py:134-139: lr 1e-4, betas (0.9,0.999), eps 1e-8):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// Optionally writes the bf16 compute copy of the updated parameter in the same pass.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            bf16* __restrict__ p_lp, long n, float lr_over_bc1, float b1, float b2, float eps, float inv_sqrt_bc2, float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        float pi = p[i] - lr_over_bc1 * (mi / denom);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (p_lp) p_lp[i] = (bf16)pi;
    }
}

// ---------------------------------------------------------------- argmax over fp32 rows (greedy decode, model.py:187)
// First-max-index tie rule (torch.argmax).  Also returns top-1 value (model.py:253 topk(1)).  One workgroup per row.
__global__ void argmax_kernel(const float* __restrict__ x0, int n, long ld, long* __restrict__ idx_out0, float* __restrict__ val_out0) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* x = x0 + (long)blockIdx.x * ld;
    long* idx_out = idx_out0 + blockIdx.x;
    float* val_out = val_out0 ? val_out0 + blockIdx.x : nullptr;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = x[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            float v2 = sv[threadIdx.x + o]; int i2 = si[threadIdx.x + o];
            if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { idx_out[0] = si[0]; if (val_out) val_out[0] = sv[0]; }
}

// ---------------------------------------------------------------- weighted late fusion (weighted_multimodal/test.py:50-61)
// token = argmax(alpha * softmax(la) + (1 - alpha) * softmax(lb)), first-index tie rule; also returns the mixed probability.
// One workgroup, three passes over the two logit rows (V <= a few thousand: L2-resident).  The mix is two roundings of
// products plus one add like torch's `alpha * p + (1 - alpha) * q` (no fma contraction), so ties break the same way.
__global__ void weighted_argmax_kernel(const float* __restrict__ la, const float* __restrict__ lb, int n, float wa, float wb,
                                       long* __restrict__ idx_out, float* __restrict__ prob_out) {
    __shared__ float sa[256], sb[256];
    __shared__ int si[256];
    const int tid = threadIdx.x;
    float ma = -INFINITY, mb = -INFINITY;
    for (int i = tid; i < n; i += 256) { ma = fmaxf(ma, la[i]); mb = fmaxf(mb, lb[i]); }
    sa[tid] = ma; sb[tid] = mb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) { sa[tid] = fmaxf(sa[tid], sa[tid + o]); sb[tid] = fmaxf(sb[tid], sb[tid + o]); } __syncthreads(); }
    ma = sa[0]; mb = sb[0];
    __syncthreads();
    float ea = 0.f, eb = 0.f;
    for (int i = tid; i < n; i += 256) { ea += expf(la[i] - ma); eb += expf(lb[i] - mb); }
    sa[tid] = ea; sb[tid] = eb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) { sa[tid] += sa[tid + o]; sb[tid] += sb[tid + o]; } __syncthreads(); }
    const float za = sa[0], zb = sb[0];
    __syncthreads();
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
        const float pa = expf(la[i] - ma) / za, pb = expf(lb[i] - mb) / zb;
        const float v = __fadd_rn(__fmul_rn(wa, pa), __fmul_rn(wb, pb));
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    sa[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v2 = sa[tid + o]; const int i2 = si[tid + o];
            if (v2 > sa[tid] || (v2 == sa[tid] && i2 < si[tid])) { sa[tid] = v2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) { idx_out[0] = si[0]; if (prob_out) prob_out[0] = sa[0]; }
}

// ---------------------------------------------------------------- top-k log-probabilities per row (beam search)
// out_val[row][j] = log_softmax(x[row])[out_idx[row][j]], j-th largest, ties broken towards the smaller index (so k = 1 is
// argmax_kernel's pick).  One workgroup per row: a max / sum-exp pass, then k selection passes over the candidates that
// come after the previous pick in (value descending, index ascending) order.
__global__ void topk_logprob_kernel(const float* __restrict__ x0, int n, long ld, int k, long* __restrict__ idx_out, float* __restrict__ val_out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* x = x0 + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, x[i]);
    sv[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) sv[tid] = fmaxf(sv[tid], sv[tid + o]); __syncthreads(); }
    mx = sv[0];
    __syncthreads();
    float se = 0.f;
    for (int i = tid; i < n; i += 256) se += expf(x[i] - mx);
    sv[tid] = se;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) sv[tid] += sv[tid + o]; __syncthreads(); }
    const float lse = mx + logf(sv[0]);
    __syncthreads();
    float last_v = INFINITY; int last_i = -1;
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            const float v = x[i];
            const bool cand = v < last_v || (v == last_v && i > last_i);
            if (cand && (v > best || (v == best && i < bi))) { best = v; bi = i; }
        }
        sv[tid] = best; si[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const float v2 = sv[tid + o]; const int i2 = si[tid + o];
                if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) { sv[tid] = v2; si[tid] = i2; }
            }
            __syncthreads();
        }
        last_v = sv[0]; last_i = si[0];
        if (tid == 0) { idx_out[(long)blockIdx.x * k + j] = last_i; val_out[(long)blockIdx.x * k + j] = last_v - lse; }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- log-STFT post-processing (audio front end)
// spec [frames][2*bins] = (re | im) halves of the windowed DFT (one fp32 GEMM against the Hann-weighted cos / -sin basis).
// Pass 1: magnitude in place of `re` and the global maximum (non-negative floats order like their bit patterns).
__global__ void stft_mag_max_kernel(float* __restrict__ spec, long frames, int bins, unsigned* __restrict__ maxbits) {
    const long n = frames * bins;
    float mx = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long f = i / bins; const int k = (int)(i % bins);
        const float re = spec[f * 2 * bins + k], im = spec[f * 2 * bins + bins + k];
        const float m = sqrtf(re * re + im * im);
        spec[f * 2 * bins + k] = m;
        mx = fmaxf(mx, m);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(mx));
}
// Pass 2: out[bin][frame] = max(20 log10(max(amin, m)) - 20 log10(max(amin, max)), -top_db) / top_db + 1
// (librosa.amplitude_to_db(ref=np.max) with its default top_db = 80, then the reference's /80 + 1; preprocessing.py:27-28)
__global__ void stft_db_kernel(const float* __restrict__ spec, long frames, int bins, const unsigned* __restrict__ maxbits, float* __restrict__ out) {
    const float amin = 1e-5f, top_db = 80.f;
    const float ref_db = 20.f * log10f(fmaxf(amin, __uint_as_float(*maxbits)));
    const long n = frames * bins;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i / frames); const long f = i % frames;       // output-major: coalesced stores
        const float db = fmaxf(20.f * log10f(fmaxf(amin, spec[f * 2 * bins + k])) - ref_db, -top_db);
        out[i] = db / top_db + 1.f;
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                         \
    if ((dtype) == OMR_F32) { typedef float T; CALL; }  \
    else if ((dtype) == OMR_BF16) { typedef bf16 T; CALL; } \
    else return OMR_ERR_UNSUPPORTED;

extern "C" int omr_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream) {
    if (n <= 0) return OMR_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = ew_grid(n);
    if (src_dtype == OMR_F32 && dst_dtype == OMR_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, EW_BLOCK, 0, s, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == OMR_BF16 && dst_dtype == OMR_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, EW_BLOCK, 0, s, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == OMR_F32 && dst_dtype == OMR_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, EW_BLOCK, 0, s, (const float*)src, (float*)dst, n);
    else if (src_dtype == OMR_BF16 && dst_dtype == OMR_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, EW_BLOCK, 0, s, (const bf16*)src, (bf16*)dst, n);
    else return OMR_ERR_UNSUPPORTED;
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_add(int dtype, const void* a, const void* b, void* out, long n, void* stream) {
    if (n <= 0) return OMR_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), ew_grid(n / Frag<T>::N + 1), EW_BLOCK, 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)out, n));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_relu_bwd(int dtype, const void* dy, const void* y, void* dx, long n, float scale, void* stream) {
    if (n <= 0) return OMR_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T>), ew_grid(n / Frag<T>::N + 1), EW_BLOCK, 0, (hipStream_t)stream, (const T*)dy, (const T*)y, (T*)dx, n, scale));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_dropout(int dtype, const void* x, void* out, long n, float p, unsigned long long seed, int channel_mode,
                           long per_sample, int C, void* stream) {
    if (n <= 0) return OMR_OK;
    if (p < 0.f || p >= 1.f) return OMR_ERR_ARG;
    uint32_t thresh = OMR_DROP_THRESH16(p);
    float scale = 1.f / (1.f - p);
    hipStream_t s = (hipStream_t)stream;
    if (channel_mode) {
        const int vec = dtype == OMR_BF16 ? 8 : 4;
        if (per_sample <= 0 || n % per_sample || C % vec || per_sample % C || per_sample > 0x7fffffffL) return OMR_ERR_ARG;
        const int B = (int)(n / per_sample);
        dim3 grid(ew_grid(per_sample / vec) > 256 ? 256 : ew_grid(per_sample / vec), B);
        DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T, true>), grid, EW_BLOCK, 0, s, (const T*)x, (T*)out, n, thresh, scale, (uint64_t)seed, per_sample, C));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T, false>), ew_grid(n / Frag<T>::N + 1), EW_BLOCK, 0, s, (const T*)x, (T*)out, n, thresh, scale,
                                             (uint64_t)seed, per_sample, C));
    }
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_embed_pe_fwd(int dtype, const long* tokens, const void* table, const float* pe, void* out, long M, int T_len, int d,
                                int vocab, void* stream) {
    if (M <= 0 || d <= 0 || T_len <= 0) return OMR_ERR_ARG;
    DISPATCH_T(dtype, hipLaunchKernelGGL((embed_pe_kernel<T>), ew_grid(M * d), EW_BLOCK, 0, (hipStream_t)stream, tokens, (const T*)table, pe, (T*)out, M, T_len, d, vocab));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_embed_bwd(int dtype, const long* tokens, const void* dout, float* dtable, long M, int d, int pad_idx, int vocab,
                             void* stream) {
    if (M <= 0 || d <= 0) return OMR_ERR_ARG;
    DISPATCH_T(dtype, hipLaunchKernelGGL((embed_bwd_kernel<T>), ew_grid(M * d), EW_BLOCK, 0, (hipStream_t)stream, tokens, (const T*)dout, dtable, M, d, pad_idx, vocab));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_add_pe2d(int dtype, const void* x, const float* pe_hwc, void* out, int B, int h, int w, int C, int maxh, int maxw,
                            void* stream) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0) return OMR_ERR_ARG;
    if (h > maxh || w > maxw) return OMR_ERR_ARG;  // feature map larger than the PE table
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (C % vec == 0 && !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)pe_hwc) & 15)) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((add_pe2d_vec_kernel<T>), ew_grid((long)B * h * w * (C / vec)), EW_BLOCK, 0, (hipStream_t)stream, (const T*)x, pe_hwc, (T*)out, B, h, w, C, maxw));
        OMR_CHECK_LAUNCH();
        return OMR_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_pe2d_kernel<T>), ew_grid((long)B * h * w * C), EW_BLOCK, 0, (hipStream_t)stream, (const T*)x, pe_hwc, (T*)out, B, h, w, C, maxw));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_colsum(int dtype, const void* dy, float* db, long M, int N, long ld, void* stream) {
    if (M <= 0 || N <= 0) return OMR_ERR_ARG;
    int rpb = 128;
    dim3 grid(cdiv(N, 64), cdiv(M, rpb));
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, 256, 0, (hipStream_t)stream, (const T*)dy, db, M, N, ld, rpb));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_adam(float* p, const float* g, float* m, float* v, void* p_bf16, long n, int step, float lr, float b1, float b2,
                        float eps, float grad_scale, void* stream) {
    if (n <= 0) return OMR_OK;
    if (step < 1) return OMR_ERR_ARG;
    double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, ew_grid(n), EW_BLOCK, 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16, n, (float)(lr / bc1), b1, b2, eps,
                       (float)(1.0 / sqrt(bc2)), grad_scale);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_log_stft_post(float* spec, long frames, int bins, unsigned* max_ws, float* out, void* stream) {
    if (frames <= 0 || bins <= 0 || !spec || !max_ws || !out) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(max_ws, 0, sizeof(unsigned), s) != hipSuccess) return OMR_ERR_LAUNCH;
    hipLaunchKernelGGL(stft_mag_max_kernel, ew_grid(frames * bins), EW_BLOCK, 0, s, spec, frames, bins, max_ws);
    hipLaunchKernelGGL(stft_db_kernel, ew_grid(frames * bins), EW_BLOCK, 0, s, (const float*)spec, frames, bins, (const unsigned*)max_ws, out);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_topk_logprob(const float* x, int rows, int n, long ld, int k, long* idx_out, float* val_out, void* stream) {
    if (n <= 0 || rows <= 0 || ld < n || k <= 0 || k > n || !idx_out || !val_out) return OMR_ERR_ARG;
    hipLaunchKernelGGL(topk_logprob_kernel, rows, 256, 0, (hipStream_t)stream, x, n, ld, k, idx_out, val_out);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_weighted_argmax(const float* logits_a, const float* logits_b, int n, float alpha, long* idx_out, float* prob_out, void* stream) {
    if (n <= 0 || !logits_a || !logits_b || !idx_out) return OMR_ERR_ARG;
    // the reference multiplies fp32 tensors by the Python floats alpha and (1 - alpha): each is rounded to fp32 once
    hipLaunchKernelGGL(weighted_argmax_kernel, 1, 256, 0, (hipStream_t)stream, logits_a, logits_b, n, alpha, (float)(1.0 - (double)alpha), idx_out, prob_out);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_argmax(const float* x, int rows, int n, long ld, long* idx_out, float* val_out, void* stream) {
    if (n <= 0 || rows <= 0 || ld < n) return OMR_ERR_ARG;
    hipLaunchKernelGGL(argmax_kernel, rows, 256, 0, (hipStream_t)stream, x, n, ld, idx_out, val_out);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
