// KV-cached greedy decoding as ONE host call per run of tokens (SURVEY.md section 8b `decode_step`, section 8f rank 1).
// Reference loop: Transformer.validation_step / get_pred_seq_and_pred_prob_seq (src/transformer/model.py:182-193,247-260)
// re-runs the whole decoder over the prefix for every token and reads the argmax back to the host each time.  Here the
// host-side executor below issues the ~12 kernels per decoder layer of ONE new position back to back from C++ (no Python, no
// per-kernel argument marshalling), appends that position's self-attention K|V to the cache by letting the K|V projection
// GEMM write straight into its cache row, reads the cross-attention K|V that were projected once per input, and chains the
// chosen token to the next step THROUGH DEVICE MEMORY (the embedding kernel of step t+1 reads the token the argmax kernel of
// step t wrote), so n_steps tokens are produced without a single host synchronisation.  Same kernels and the same per-row
// arithmetic as the training forward pass: the tokens equal the full re-run's (tests/test_model_gpu.py).
#include <type_traits>

#include "omr_common.h"
#include "omr_hip.h"

namespace {

struct Ws {          // activation scratch of one step, carved out of the caller's workspace
    char* x; char* x2; char* q; char* o; char* proj; char* h; char* logits; float* logits32; float* lse; float* mean; float* rstd;
    float* split; long split_floats; float* apart;
    unsigned char* a8; float* sa8;          // fp8 mode: the quantised input rows of the current GEMM and their scales
};

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

size_t carve(const omr_decode_desc& d, char* base, Ws* w) {
    const size_t es = d.dtype == OMR_BF16 ? 2 : 4, B = (size_t)d.B;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    char* x = take(B * d.d * es); char* x2 = take(B * d.d * es); char* q = take(B * d.d * es); char* o = take(B * d.d * es); char* proj = take(B * d.d * es);
    char* h = take(B * (size_t)d.ff * es); char* logits = take(B * (size_t)d.ldv * es);
    float* l32 = (float*)take(B * (size_t)d.ldv * 4); float* lse = (float*)take(B * (size_t)d.nhead * 4);
    float* mean = (float*)take(B * 4); float* rstd = (float*)take(B * 4);
    const int smax = d.S > d.max_len ? d.S : d.max_len;                // key-split partials of the longer of the two attentions
    const long sf = omr_attn_split_workspace_floats(d.B, d.nhead, 1, smax, d.d / d.nhead);
    float* split = (float*)take((size_t)sf * 4);
    float* apart = (float*)take(2 * B * (size_t)((d.V + 15) / 16) * 4);      // greedy pick: the head kernel's per-workgroup candidates
    const size_t kmax = (size_t)(d.d > d.ff ? d.d : d.ff);
    unsigned char* a8 = (unsigned char*)take(d.fp8 ? B * ((kmax + 15) / 16 * 16) : 0);
    float* sa8 = (float*)take(d.fp8 ? B * 4 : 0);
    if (w) *w = Ws{x, x2, q, o, proj, h, logits, l32, lse, mean, rstd, split, sf, apart, a8, sa8};
    return off;
}

#define TRY(call) do { int rc__ = (call); if (rc__ != OMR_OK) return rc__; } while (0)

// ------------------------------------------------------------------------------------------------
// Row linear of a decode position (omr_decode_linear).  A position is a chain of ~50 dependent launches of almost no work, so
// what counts is how FEW launches there are and how short each one's dependent latency is -- not MFMA throughput (M = the
// batch rows of one position).  Workgroup = 16 output columns x 16 k-lanes; the weight chunks of a thread are requested
// first, the input rows are built while they fly (LayerNorm of the previous sub-layer / embedding / merge of the key-split
// attention partials: the element-wise kernels that used to sit between the GEMMs), then a fixed-order fp32 dot product per
// (row, column): chunks in ascending k, the 16 k-lanes combined by a fixed cross-lane tree.  blockIdx.y picks RM rows; nothing in a row's
// arithmetic depends on M or on the other rows.
constexpr int RM = 8, NOUT = 16, KL = 16, WCH = 8;      // rows per workgroup, columns per workgroup, k-lanes, prefetched weight chunks per thread
constexpr int MAXSPLIT = 64, MAXHS = 512;                // key splits the merge prologue takes (attention.hip choose_split caps a decode
                                                         // row at 64 splits of >= 256 keys: the reference's largest memory, 12 696 tokens,
                                                         // is 50); heads x splits

// Sum over the 16 k-lanes of a column (= one DPP row): four cross-lane adds, every lane ends with the total.  (The generic
// __shfl_xor butterfly is ~7 instructions per step through the LDS crossbar.)
__device__ __forceinline__ float klane_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// e4m3 (OCP) weight chunk -> fp32: 8 codes per lane and chunk (gfx950 converts two codes per v_cvt_pk_f32_fp8)
struct W8Chunk { uint2 v; };
__device__ __forceinline__ void w_unpack(const W8Chunk& c, float (&f)[8]) {
    const auto p0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.v.x, false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.v.x, true);
    const auto p2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.v.y, false), p3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.v.y, true);
    f[0] = p0[0]; f[1] = p0[1]; f[2] = p1[0]; f[3] = p1[1]; f[4] = p2[0]; f[5] = p2[1]; f[6] = p3[0]; f[7] = p3[1];
}
__device__ __forceinline__ void w_unpack(const bf16x8& c, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = to_f32(c[e]);
}
__device__ __forceinline__ void w_unpack(const f32x4& c, float (&f)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = c[e];
}

template <typename T, bool W8>
__global__ __launch_bounds__(256) void decode_linear_kernel(omr_decode_linear_args a) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = W8 ? 8 : Frag<T>::N;                            // weight elements per chunk (fp8: 8 codes = 8 bytes)
    typedef typename std::conditional<W8, W8Chunk, F>::type WF;
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [RM][K]: the rows as the GEMM sees them (values rounded to T)
    __shared__ float mls[4 * 2 * MAXHS];                                 // prologue 3: per-wave (max | sum) strips
    __shared__ float cand[RM][NOUT];                                     // greedy pick: the workgroup's rounded outputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = tid / KL, kl = tid % KL;
    const int n = blockIdx.x * NOUT + nl, K = a.K, nch = K / VEC;
    typedef typename std::conditional<W8, unsigned char, T>::type WT;
    const WT* wrow = (W8 ? (const WT*)a.w8 : (const WT*)a.w) + (long)(n < a.N ? n : 0) * K;

    // K is a multiple of KL * VEC (host check): chunk kl + i * KL exists for every lane or for none, so the loops over a thread's
    // chunks have block-uniform bounds and the loads carry no per-lane test (columns past N read row 0 and are never stored)
    const int cpt = nch / KL;
    WF wv[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i)
        if (i < cpt) wv[i] = *reinterpret_cast<const WF*>(wrow + (kl + i * KL) * VEC);
    const int per = K / 64;                                              // prologues 1-3: a wave builds a row, lane = `per` consecutive columns
    {
        const int r0 = blockIdx.y * RM, rm = min(RM, a.M - r0);
        for (int r = wave; r < rm; r += 4) {
            const long m = r0 + r;
            float* xr = xs + r * K;
            if (a.pro == 0) {
                const T* src = (const T*)a.x + m * a.ldx;
                for (int k = lane; k < K; k += 64) xr[k] = to_f32(src[k]);
            } else if (a.pro == 1) {        // add + LayerNorm, same lane layout and summation order as add_ln_fwd_kernel (norm.hip)
                const T* y = (const T*)a.x + m * a.ldx + lane * per;
                const T* rs_ = (const T*)a.res + m * a.ldres + lane * per;
                auto run = [&](auto per_c) {
                    constexpr int PER = decltype(per_c)::value;
                    float v[PER], mu, rstd;
#pragma unroll
                    for (int i = 0; i < PER; ++i) v[i] = to_f32(y[i]) + to_f32(rs_[i]);
                    ln_row<PER>(v, a.gamma, a.beta, lane, a.eps, mu, rstd);
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const T o = from_f32<T>(v[i]);
                        xr[lane * PER + i] = to_f32(o);
                        if (blockIdx.x == 0) ((T*)a.xn_out)[m * K + lane * PER + i] = o;
                    }
                };
                if (per == 2) run(std::integral_constant<int, 2>());
                else if (per == 4) run(std::integral_constant<int, 4>());
                else run(std::integral_constant<int, 8>());
            } else if (a.pro == 2) {        // embedding + positional row (embed_pe_kernel, elementwise.hip)
                const long t = a.tokens[m];
                const bool ok = t >= 0 && t < a.vocab;
                for (int i = 0; i < per; ++i) {
                    const int k = lane * per + i;
                    const T o = from_f32<T>((ok ? to_f32(((const T*)a.emb)[t * K + k]) : 0.f) + a.pe_row[k]);
                    xr[k] = to_f32(o);
                    if (blockIdx.x == 0) ((T*)a.xn_out)[m * K + k] = o;
                }
            } else {                        // merge of the key-split partial softmaxes: attn_split_merge_kernel's arithmetic in its
                                            // order.  The (max, sum) pairs of the row's H * nsplit partials go through a per-wave LDS
                                            // strip first (one global round trip for all of them); a lane's `per` columns lie in one head
                const int hs = a.H * a.nsplit, stride = a.hd + 2;
                float* ml = mls + wave * (2 * MAXHS);
                for (int l = lane; l < hs; l += 64) {
                    const float* P = a.part + (m * hs + l) * stride;
                    ml[l] = P[a.hd];
                    ml[MAXHS + l] = P[a.hd + 1];
                }
                // the LDS queue of a wave is in order: the reads below follow the writes above
                const int h = (lane * per) / a.hd, dch = lane * per - h * a.hd;
                const float* mh = ml + h * a.nsplit;
                const float* P = a.part + ((m * a.H + h) * a.nsplit) * stride + dch;
                float mm = -INFINITY;
                for (int j = 0; j < a.nsplit; ++j) mm = fmaxf(mm, mh[j]);
                float l_tot = 0.f, o[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll 4
                for (int j = 0; j < a.nsplit; ++j) {
                    const float mj = mh[j];
                    const float wj = mj == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mj - mm);
                    l_tot += mh[MAXHS + j] * wj;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (i < per) o[i] += P[j * stride + i] * wj;
                }
                const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i < per) xr[lane * per + i] = to_f32(from_f32<T>(o[i] * inv));
            }
        }
        __syncthreads();
        float acc[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) acc[r] = 0.f;
        auto slice_rows = [&](auto nr_c) {       // one row (bs 1, the reference's loop) takes the lean single-row body
            constexpr int NR = decltype(nr_c)::value;
#pragma unroll
            for (int i = 0; i < WCH; ++i)
                if (i < cpt) {
                    const float* xc = xs + (kl + i * KL) * VEC;
                    float wf[VEC];
                    w_unpack(wv[i], wf);
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[r] = fmaf(wf[e], xc[r * K + e], acc[r]);
                }
            for (int i = WCH; i < cpt; ++i) {                            // K beyond the prefetched chunks
                const WF wx = *reinterpret_cast<const WF*>(wrow + (kl + i * KL) * VEC);
                const float* xc = xs + (kl + i * KL) * VEC;
                float wf[VEC];
                w_unpack(wx, wf);
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[r] = fmaf(wf[e], xc[r * K + e], acc[r]);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = klane_sum(acc[r]);
        };
        if (rm == 1) slice_rows(std::integral_constant<int, 1>());
        else slice_rows(std::integral_constant<int, RM>());             // rows past rm: arithmetic on stale LDS, never stored
        if (kl == 0 && n < a.N) {
            const float bv = a.bias ? a.bias[n] : 0.f, wsc = W8 ? a.w8_scale[n] : 1.f;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                if (r >= rm) break;
                float v = W8 ? fmaf(acc[r], wsc, bv) : acc[r] + bv;
                if (a.relu) v = fmaxf(v, 0.f);
                const T o = from_f32<T>(v);
                const long m = r0 + r;
                if (n < a.n0) ((T*)a.out0)[m * a.ld0 + n] = o;
                else ((T*)a.out1)[m * a.ld1 + (n - a.n0)] = o;
                if (a.out32) a.out32[m * a.ld32 + n] = to_f32(o);
                cand[r][nl] = to_f32(o);
            }
        }
        // ---- greedy pick, first half: this workgroup's candidate per row (value, column); decode_pick_kernel reduces the
        //      ceil(N/16) candidates of a row.  (Letting the last workgroup to finish do that -- counter + agent-scope fences --
        //      was measured: 20 us slower per position than the second launch.)
        if (a.amax_part) {
            if (kl == 0 && n >= a.N)
#pragma unroll
                for (int r = 0; r < RM; ++r) cand[r][nl] = -INFINITY;
            __syncthreads();
            if (tid < rm) {
                float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < NOUT; ++c) {
                    const float v = cand[tid][c];
                    if (v > best) { best = v; bi = blockIdx.x * NOUT + c; }            // ascending columns: the first maximum stays
                }
                float* pp = a.amax_part + ((long)(r0 + tid) * gridDim.x + blockIdx.x) * 2;
                pp[0] = best; pp[1] = __int_as_float(bi);
            }
        }
    }
}

// greedy pick, second half: one wave per row over the G (value, column) candidates; first index of the maximum (torch.argmax)
__global__ __launch_bounds__(64) void decode_pick_kernel(const float* __restrict__ part, int G, long* __restrict__ idx_out, float* __restrict__ val_out) {
    const int lane = threadIdx.x;
    const float* pp = part + (long)blockIdx.x * G * 2;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < G; i += 64) {
        const float v = pp[2 * i]; const int ii = __float_as_int(pp[2 * i + 1]);
        if (v > best || (v == best && ii < bi)) { best = v; bi = ii; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(best, o, 64); const int ii = __shfl_xor(bi, o, 64);
        if (v > best || (v == best && ii < bi)) { best = v; bi = ii; }
    }
    if (lane == 0) { idx_out[blockIdx.x] = bi; if (val_out) val_out[blockIdx.x] = best; }
}

}  // namespace

extern "C" int omr_decode_linear(const omr_decode_linear_args* ap, void* stream) {
    if (!ap) return OMR_ERR_ARG;
    const omr_decode_linear_args& a = *ap;
    const int vec = (a.dtype == OMR_BF16 || a.w8) ? 8 : 4;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % vec || a.K > 2048 || (!a.w && !a.w8) || !a.out0 || a.n0 < 0) return OMR_ERR_ARG;
    if (a.w8 ? (!a.w8_scale || ((uintptr_t)a.w8 & 7)) : (((uintptr_t)a.w & 15) != 0)) return OMR_ERR_ARG;
    if (a.n0 < a.N && !a.out1) return OMR_ERR_ARG;
    if (a.pro < 0 || a.pro > 3) return OMR_ERR_ARG;
    if (a.K % (KL * vec)) return OMR_ERR_UNSUPPORTED;                    // whole 16-lane chunk groups (128 bf16 / 64 fp32 columns)
    if (a.amax_idx && !a.amax_part) return OMR_ERR_ARG;
    if (a.pro && (a.K % 64 || a.K / 64 > 16)) return OMR_ERR_ARG;
    if (a.pro == 1 && a.K != 128 && a.K != 256 && a.K != 512) return OMR_ERR_UNSUPPORTED;      // the widths omr_add_layernorm_fwd takes
    if ((a.pro == 0 || a.pro == 1) && !a.x) return OMR_ERR_ARG;
    if (a.pro == 1 && (!a.res || !a.gamma || !a.beta || !a.xn_out)) return OMR_ERR_ARG;
    if (a.pro == 2 && (!a.tokens || !a.emb || !a.pe_row || !a.xn_out)) return OMR_ERR_ARG;
    if (a.pro == 3 && (!a.part || a.nsplit < 1 || a.nsplit > MAXSPLIT || a.H < 1 || a.hd < 1 || a.H * a.hd != a.K || a.H * a.nsplit > MAXHS || a.hd % (a.K / 64))) return OMR_ERR_ARG;
    const dim3 grid((unsigned)cdiv(a.N, NOUT), (unsigned)cdiv(a.M, RM)), block(256);
    const size_t shm = (size_t)RM * a.K * sizeof(float);
    if (a.dtype == OMR_BF16 && a.w8) hipLaunchKernelGGL((decode_linear_kernel<bf16, true>), grid, block, shm, (hipStream_t)stream, a);
    else if (a.dtype == OMR_F32 && a.w8) hipLaunchKernelGGL((decode_linear_kernel<float, true>), grid, block, shm, (hipStream_t)stream, a);
    else if (a.dtype == OMR_BF16) hipLaunchKernelGGL((decode_linear_kernel<bf16, false>), grid, block, shm, (hipStream_t)stream, a);
    else if (a.dtype == OMR_F32) hipLaunchKernelGGL((decode_linear_kernel<float, false>), grid, block, shm, (hipStream_t)stream, a);
    else return OMR_ERR_UNSUPPORTED;
    if (a.amax_idx) hipLaunchKernelGGL(decode_pick_kernel, dim3((unsigned)a.M), dim3(64), 0, (hipStream_t)stream, a.amax_part, (int)grid.x, a.amax_idx, a.amax_val);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" long omr_decode_workspace_bytes(const omr_decode_desc* d) {
    if (!d || d->B <= 0 || d->d <= 0 || d->ff <= 0 || d->ldv < d->V) return OMR_ERR_ARG;
    return (long)carve(*d, nullptr, nullptr);
}

extern "C" int omr_decode_steps(const omr_decode_desc* dp, long* tokens, int t0, int n_steps, long* out_tokens, float* out_top1,
                                float* last_logits, void* stream) {
    if (!dp || !tokens || n_steps < 1 || t0 < 0) return OMR_ERR_ARG;
    const omr_decode_desc& d = *dp;
    if (d.B <= 0 || d.L <= 0 || d.d <= 0 || d.d % d.nhead || d.V <= 0 || d.ldv < d.V || d.ldv % 8) return OMR_ERR_ARG;
    if (t0 + n_steps > d.max_len) return OMR_ERR_ARG;                   // positional-encoding table / cache exhausted (decoder.py:31)
    if (n_steps > 1 && !out_tokens) return OMR_ERR_ARG;                 // several steps need the token feedback
    if (!d.emb || !d.pe || !d.layer_w || !d.head_w || !d.self_kv || !d.cross_kv || !d.ws) return OMR_ERR_ARG;
    if (d.ws_bytes < (long)carve(d, nullptr, nullptr)) return OMR_ERR_ARG;
    Ws w;
    carve(d, (char*)d.ws, &w);
    const int dt = d.dtype, B = d.B, dm = d.d, hd = dm / d.nhead;
    const size_t es = dt == OMR_BF16 ? 2 : 4;
    if (d.fp8 && (!d.layer_w8 || !d.layer_s8 || !d.head_w8 || !d.head_s8 || dm % 16 || d.ff % 16)) return OMR_ERR_ARG;
    // One linear: bf16 / fp32 GEMM on the layer's own weights, or (fp8 mode) quantise the B input rows per token and run the
    // fp8 MFMA GEMM on the pre-quantised weight `w8` (rows row0 .. row0+N of matrix `mat` of this layer; scales alongside).
    const void* const* W = nullptr;
    const unsigned char* const* W8 = nullptr;
    const float* const* S8 = nullptr;
    auto gemm = [&](const void* a, long lda, int widx, const float* bias, void* c, long ldc, int N, int K, int relu, int mat, int row0) -> int {
        if (!d.fp8)
            return omr_gemm(dt, dt, 0, 0, B, N, K, a, lda, (const char*)W[widx] + (size_t)row0 * K * es, K, c, ldc, bias, relu, 0, 1, nullptr, 0.f, 0, 0, 0, 0, 0, stream);
        int rc = omr_quantize_rows_fp8(dt, a, lda, w.a8, (K + 15) / 16 * 16, w.sa8, B, K, stream);
        if (rc != OMR_OK) return rc;
        return omr_gemm_fp8(dt, B, N, K, w.a8, (K + 15) / 16 * 16, w.sa8, W8[mat] + (size_t)row0 * K, K, S8[mat] + row0, c, ldc, bias, relu, stream);
    };
    // ---- 8 launches per layer (bf16 / fp32 weights, or e4m3 weights dequantised on load by the row kernel).  Every element-wise step between two linears (embedding + positional row, the
    // three add + LayerNorm, the merge of the key-split attention) is folded into the loading of the NEXT linear's input rows
    // (omr_decode_linear prologues); the residual stream alternates between two buffers because the workgroup that stores a
    // freshly normalised row runs beside workgroups still reading the previous one.
    const int smax = d.S > d.max_len ? d.S : d.max_len, splits_max = (smax + 255) / 256 < MAXSPLIT ? (smax + 255) / 256 : MAXSPLIT;
    if ((dm == 128 || dm == 256 || dm == 512) && d.ff <= 2048 && d.ff % (16 * ((dt == OMR_BF16 || d.fp8) ? 8 : 4)) == 0 && d.nhead * splits_max <= MAXHS) {
        const long* tok_in = tokens;
        for (int s = 0; s < n_steps; ++s) {
            const int t = t0 + s;
            const int lo = (d.window > 0 && t - d.window > 0) ? t - d.window : 0;  // banded causal mask = a key range (decoder.py:213-214)
            char* xa = w.x; char* xb = w.x2;                                        // xa: residual stream entering the sub-layer
            auto lin = [&](int pro, const void* x, const void* res, const float* g, const float* bt, void* xn_out, const float* part, int nsplit,
                           const void* wmat, const float* bias, int N, int K, int relu, void* out0, long ld0, int n0, void* out1, long ld1,
                           float* out32, long* amax_idx = nullptr, float* amax_val = nullptr, const unsigned char* w8 = nullptr,
                           const float* s8 = nullptr) -> int {
                omr_decode_linear_args a = {};
                a.w8 = d.fp8 ? w8 : nullptr; a.w8_scale = d.fp8 ? s8 : nullptr;
                a.amax_idx = amax_idx; a.amax_val = amax_val; a.amax_part = amax_idx ? w.apart : nullptr;
                a.dtype = dt; a.pro = pro; a.M = B; a.N = N; a.K = K; a.relu = relu; a.n0 = n0; a.nsplit = nsplit; a.H = d.nhead; a.hd = hd; a.vocab = d.V;
                a.eps = 1e-5f; a.x = x; a.ldx = K; a.res = res; a.ldres = K; a.gamma = g; a.beta = bt; a.xn_out = xn_out;
                a.tokens = tok_in; a.emb = d.emb; a.pe_row = d.pe + (size_t)t * dm; a.part = part;
                a.w = wmat; a.bias = bias; a.out0 = out0; a.ld0 = ld0; a.out1 = out1; a.ld1 = ld1; a.out32 = out32; a.ld32 = d.ldv;
                return omr_decode_linear(&a, stream);
            };
            const float *pg = nullptr, *pb = nullptr;                               // norm3 of the previous layer, still to be applied
            for (int l = 0; l < d.L; ++l) {
                const void* const* Wl = d.layer_w + (size_t)l * OMR_DECODE_LAYER_PTRS;
                const unsigned char* const* W8l = d.fp8 ? d.layer_w8 + (size_t)l * OMR_DECODE_LAYER_FP8 : nullptr;     // e4m3 rows + row scales of the
                const float* const* S8l = d.fp8 ? d.layer_s8 + (size_t)l * OMR_DECODE_LAYER_FP8 : nullptr;             // layer's six matrices
                char* cache_l = (char*)d.self_kv + ((size_t)l * B * d.max_len) * 2 * dm * es;
                char* kv_row = cache_l + (size_t)t * 2 * dm * es;
                // q | k|v projection of the position: q -> w.q, k|v straight into row t of the cache.  Its input is the
                // embedding (layer 0) or norm3(x + ffn) of the previous layer; either way the rows land in xb
                if (l == 0) TRY(lin(2, nullptr, nullptr, nullptr, nullptr, xb, nullptr, 0, Wl[0], (const float*)Wl[1], 3 * dm, dm, 0, w.q, dm, dm, kv_row, (long)d.max_len * 2 * dm, nullptr, nullptr, nullptr, W8l ? W8l[0] : nullptr, S8l ? S8l[0] : nullptr));
                else TRY(lin(1, w.proj, xa, pg, pb, xb, nullptr, 0, Wl[0], (const float*)Wl[1], 3 * dm, dm, 0, w.q, dm, dm, kv_row, (long)d.max_len * 2 * dm, nullptr, nullptr, nullptr, W8l ? W8l[0] : nullptr, S8l ? S8l[0] : nullptr));
                { char* tsw = xa; xa = xb; xb = tsw; }
                const char* k0 = cache_l + (size_t)lo * 2 * dm * es;
                int ns = 1;
                TRY(omr_attn_fwd_split_partials(dt, w.q, k0, k0 + (size_t)dm * es, w.o, w.lse, dm, 2 * dm, 2 * dm, dm, dm, (long)d.max_len * 2 * dm,
                                                (long)d.max_len * 2 * dm, dm, B, d.nhead, 1, t + 1 - lo, hd, w.split, w.split_floats, &ns, stream));
                TRY(lin(ns > 1 ? 3 : 0, w.o, nullptr, nullptr, nullptr, nullptr, w.split, ns, Wl[2], (const float*)Wl[3], dm, dm, 0, w.proj, dm, dm, nullptr, 0, nullptr, nullptr, nullptr, W8l ? W8l[1] : nullptr, S8l ? S8l[1] : nullptr));
                // cross-attention query from norm1(x + self-attention)
                TRY(lin(1, w.proj, xa, (const float*)Wl[4], (const float*)Wl[5], xb, nullptr, 0, Wl[6], (const float*)Wl[7], dm, dm, 0, w.q, dm, dm, nullptr, 0, nullptr, nullptr, nullptr, W8l ? W8l[2] : nullptr, S8l ? S8l[2] : nullptr));
                { char* tsw = xa; xa = xb; xb = tsw; }
                const char* ck = (const char*)d.cross_kv + (size_t)l * 2 * dm * es;
                TRY(omr_attn_fwd_split_partials(dt, w.q, ck, ck + (size_t)dm * es, w.o, w.lse, dm, d.cross_ld, d.cross_ld, dm, dm, d.cross_bs, d.cross_bs, dm,
                                                B, d.nhead, 1, d.S, hd, w.split, w.split_floats, &ns, stream));
                TRY(lin(ns > 1 ? 3 : 0, w.o, nullptr, nullptr, nullptr, nullptr, w.split, ns, Wl[8], (const float*)Wl[9], dm, dm, 0, w.proj, dm, dm, nullptr, 0, nullptr, nullptr, nullptr, W8l ? W8l[3] : nullptr, S8l ? S8l[3] : nullptr));
                // feed-forward from norm2(x + cross-attention)
                TRY(lin(1, w.proj, xa, (const float*)Wl[10], (const float*)Wl[11], xb, nullptr, 0, Wl[12], (const float*)Wl[13], d.ff, dm, 1, w.h, d.ff, d.ff, nullptr, 0, nullptr, nullptr, nullptr, W8l ? W8l[4] : nullptr, S8l ? S8l[4] : nullptr));
                { char* tsw = xa; xa = xb; xb = tsw; }
                TRY(lin(0, w.h, nullptr, nullptr, nullptr, nullptr, nullptr, 0, Wl[14], (const float*)Wl[15], dm, d.ff, 0, w.proj, dm, dm, nullptr, 0, nullptr, nullptr, nullptr, W8l ? W8l[5] : nullptr, S8l ? S8l[5] : nullptr));
                pg = (const float*)Wl[16]; pb = (const float*)Wl[17];
            }
            // vocabulary head (Conv1d k=1, decoder.py:145-146) on norm3 of the last layer: logits rounded to the compute dtype like
            // the training forward, kept as fp32 rows
            // ... and the greedy pick (model.py:187,253): candidates from the head's workgroups, one small launch to reduce them; the
            // next position reads the token from where it was written
            TRY(lin(1, w.proj, xa, pg, pb, xb, nullptr, 0, d.head_w, d.head_b, d.V, dm, 0, w.logits, d.ldv, d.V, nullptr, 0, w.logits32,
                    out_tokens ? out_tokens + (size_t)s * B : nullptr, (out_tokens && out_top1) ? out_top1 + (size_t)s * B : nullptr, d.head_w8, d.head_s8));
            if (out_tokens) tok_in = out_tokens + (size_t)s * B;
        }
        if (out_tokens && hipMemcpyAsync(tokens, out_tokens + (size_t)(n_steps - 1) * B, (size_t)B * sizeof(long), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return OMR_ERR_LAUNCH;
        if (last_logits && hipMemcpyAsync(last_logits, w.logits32, (size_t)B * d.ldv * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return OMR_ERR_LAUNCH;
        return OMR_OK;
    }
    // ---- model widths the row kernel does not take: one GEMM / element-wise kernel per step of the layer (fp8 mode: activations
    //      quantised per token, fp8 MFMA GEMM)
    for (int s = 0; s < n_steps; ++s) {
        const int t = t0 + s;
        // embedding(tgt) + pe[t]  (decoder.py:124; T_len = 1 so every row of the batch takes the table row given)
        TRY(omr_embed_pe_fwd(dt, tokens, d.emb, d.pe + (size_t)t * dm, w.x, B, 1, dm, d.V, stream));
        const int lo = (d.window > 0 && t - d.window > 0) ? t - d.window : 0;      // banded causal mask = a key range (decoder.py:213-214)
        for (int l = 0; l < d.L; ++l) {
            W = d.layer_w + (size_t)l * OMR_DECODE_LAYER_PTRS;
            if (d.fp8) { W8 = d.layer_w8 + (size_t)l * OMR_DECODE_LAYER_FP8; S8 = d.layer_s8 + (size_t)l * OMR_DECODE_LAYER_FP8; }
            char* cache_l = (char*)d.self_kv + ((size_t)l * B * d.max_len) * 2 * dm * es;
            // self-attention: q rows of the packed in_proj; the k|v rows go straight into position t of the cache
            TRY(gemm(w.x, dm, 0, (const float*)W[1], w.q, dm, dm, dm, 0, 0, 0));
            TRY(gemm(w.x, dm, 0, (const float*)W[1] + dm, cache_l + (size_t)t * 2 * dm * es, (long)d.max_len * 2 * dm, 2 * dm, dm, 0, 0, dm));
            const char* k0 = cache_l + (size_t)lo * 2 * dm * es;
            TRY(omr_attn_fwd_split(dt, w.q, k0, k0 + (size_t)dm * es, w.o, w.lse, dm, 2 * dm, 2 * dm, dm, dm, (long)d.max_len * 2 * dm, (long)d.max_len * 2 * dm, dm,
                                   B, d.nhead, 1, t + 1 - lo, hd, nullptr, w.split, w.split_floats, stream));
            TRY(gemm(w.o, dm, 2, (const float*)W[3], w.proj, dm, dm, dm, 0, 1, 0));
            TRY(omr_add_layernorm_fwd(dt, w.proj, w.x, (const float*)W[4], (const float*)W[5], w.x, w.mean, w.rstd, B, dm, 1e-5f, 0.f, 0, stream));
            // cross-attention over the memory K|V projected once (init): layer l's block of the [B][S][L*2d] buffer
            TRY(gemm(w.x, dm, 6, (const float*)W[7], w.q, dm, dm, dm, 0, 2, 0));
            const char* ck = (const char*)d.cross_kv + (size_t)l * 2 * dm * es;
            TRY(omr_attn_fwd_split(dt, w.q, ck, ck + (size_t)dm * es, w.o, w.lse, dm, d.cross_ld, d.cross_ld, dm, dm, d.cross_bs, d.cross_bs, dm,
                                   B, d.nhead, 1, d.S, hd, nullptr, w.split, w.split_floats, stream));
            TRY(gemm(w.o, dm, 8, (const float*)W[9], w.proj, dm, dm, dm, 0, 3, 0));
            TRY(omr_add_layernorm_fwd(dt, w.proj, w.x, (const float*)W[10], (const float*)W[11], w.x, w.mean, w.rstd, B, dm, 1e-5f, 0.f, 0, stream));
            // feed-forward
            TRY(gemm(w.x, dm, 12, (const float*)W[13], w.h, d.ff, d.ff, dm, 1, 4, 0));
            TRY(gemm(w.h, d.ff, 14, (const float*)W[15], w.proj, dm, dm, d.ff, 0, 5, 0));
            TRY(omr_add_layernorm_fwd(dt, w.proj, w.x, (const float*)W[16], (const float*)W[17], w.x, w.mean, w.rstd, B, dm, 1e-5f, 0.f, 0, stream));
        }
        // vocabulary head (Conv1d k=1, decoder.py:145-146) in the compute dtype like the training forward, then fp32 rows
        if (!d.fp8) {
            TRY(omr_gemm(dt, dt, 0, 0, B, d.V, dm, w.x, dm, d.head_w, dm, w.logits, d.ldv, d.head_b, 0, 0, 1, nullptr, 0.f, 0, 0, 0, 0, 0, stream));
        } else {
            TRY(omr_quantize_rows_fp8(dt, w.x, dm, w.a8, dm, w.sa8, B, dm, stream));
            TRY(omr_gemm_fp8(dt, B, d.V, dm, w.a8, dm, w.sa8, d.head_w8, dm, d.head_s8, w.logits, d.ldv, d.head_b, 0, stream));
        }
        float* l32 = w.logits32;
        if (dt == OMR_F32) l32 = (float*)w.logits;
        else TRY(omr_cast(w.logits, dt, w.logits32, OMR_F32, (long)B * d.ldv, stream));
        if (out_tokens) {
            // greedy pick (model.py:187,253); the token is fed back to the next step through device memory
            TRY(omr_argmax(l32, B, d.V, d.ldv, out_tokens + (size_t)s * B, out_top1 ? out_top1 + (size_t)s * B : nullptr, stream));
            if (hipMemcpyAsync(tokens, out_tokens + (size_t)s * B, (size_t)B * sizeof(long), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                return OMR_ERR_LAUNCH;
        }
        if (last_logits && s == n_steps - 1) {
            if (hipMemcpyAsync(last_logits, l32, (size_t)B * d.ldv * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                return OMR_ERR_LAUNCH;
        }
    }
    return OMR_OK;
}

/* Weighted late fusion (src/multimodal/weighted_multimodal/test.py:21-70) as ONE host call per run of tokens: two unimodal
 * models with their own KV caches decode the same prefix in lock-step; per position both descriptors run their step
 * (omr_decode_steps without a pick: fp32 logits only), omr_weighted_argmax mixes the two softmaxes and picks, and the token
 * reaches BOTH models' next position through device memory.  bs = 1 like the reference (test.py:27). */
extern "C" int omr_weighted_decode_steps(const omr_decode_desc* da, const omr_decode_desc* db, float alpha, long* tokens, int t0, int n_steps,
                                         long* out_tokens, float* out_prob, float* logits_a, float* logits_b, void* stream) {
    if (!da || !db || !tokens || !out_tokens || !logits_a || !logits_b || n_steps < 1 || t0 < 0) return OMR_ERR_ARG;
    if (da->B != 1 || db->B != 1 || da->V != db->V) return OMR_ERR_ARG;
    for (int s = 0; s < n_steps; ++s) {
        TRY(omr_decode_steps(da, tokens, t0 + s, 1, nullptr, nullptr, logits_a, stream));
        TRY(omr_decode_steps(db, tokens, t0 + s, 1, nullptr, nullptr, logits_b, stream));
        TRY(omr_weighted_argmax(logits_a, logits_b, da->V, alpha, out_tokens + s, out_prob ? out_prob + s : nullptr, stream));
        if (hipMemcpyAsync(tokens, out_tokens + s, sizeof(long), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return OMR_ERR_LAUNCH;
    }
    return OMR_OK;
}
