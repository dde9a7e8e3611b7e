// Fused multi-head attention (flash style: scores never leave the chip) for gfx950.
// Reference math: torch nn/functional.py multi_head_attention_forward as used by
//   nn.TransformerDecoderLayer self/cross attention (decoder.py:86-95) and CrossAttention (model.py:289-355):
//   P = softmax(Q K^T / sqrt(hd) + bias), O = dropout(P) V, heads = contiguous hd-wide channel slices.
// Mask semantics reproduced exactly:
//   * key_bias[b][key]  : ADDED to the score.  Float padding masks add +1.0 (quirk 1); bool masks arrive as -inf.
//   * causal / window   : key <= q, and key >= q - window when 0 < window < T (decoder.py:191-217).
//   * block mask        : score = -inf where q >= lq[b'] and key >= lkv[b'] with b' = (b*H + h) % B (quirk 2, model.py:349-354).
//
// Data flow per wave (32 query rows), keys on the MFMA M dimension so that each LANE owns one query row:
//   S^T = K . Q^T        (A = K tile from LDS, B = Q fragments held in registers)      -> softmax is lane-local
//   O^T += V^T . P^T     (A = V^T tile from LDS, B = P^T straight from the accumulator registers, permuted-k order)
// Backward: dQ kernel (same orientation) and dK/dV kernel (queries on the register axis, keys on the lanes).
#include <cstdlib>
#include <type_traits>

#include "omr_common.h"
#include "omr_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o;
    long ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;      // row / batch strides in elements
    float* lse;                                        // [B][H][T]
    const float* key_bias;                             // [B][S] or null
    const int* blk_lq; const int* blk_lkv;             // [B] or null
    int B, H, T, S; float scale; int causal; int window;
    uint32_t drop_thresh; float drop_scale; uint64_t seed;
    const uint64_t* dmask;                             // dropout keep bits (attn_dropout_words_kernel); passed to the kernels as a
                                                       // separate __restrict__ parameter so that its loads become scalar loads
    // single-query-block forward split over the keys (decode): blockIdx.x = split, partials [B][H][nsplit][T][HD + 2] floats
    int nsplit, split_len; float* part;
    // backward only
    const void* dout; long lddo, bsdo;
    const float* delta;                                // [B][H][T]
    void* dq; void* dk; void* dv; long lddq, lddk, lddv, bsdq, bsdk, bsdv;
};

template <typename T> struct ACfg {
    static constexpr int MPI = std::is_same<T, bf16>::value ? 1 : 4;      // MFMA instructions per mma32 (sched_group_barrier counts)
    static constexpr int RPF = std::is_same<T, bf16>::value ? 2 : 1;      // LDS reads per permuted-k fragment (kperm_frag)
    static constexpr int VEC = Frag<T>::N;
    static constexpr int NFR = 16 / VEC;   // operand fragments per 32-wide accumulator block (2 bf16 / 4 fp32)
};

// Fragment of a k-contiguous LDS row whose k order matches accumulator registers s*VEC .. s*VEC+VEC-1 of a
// 32-row block: element j  <->  k = kb + acc_row(s*VEC + j, lane).
template <typename T> __device__ __forceinline__ typename Frag<T>::type load_kperm_frag(const T* row, int kb, int s, int h);
template <> __device__ __forceinline__ bf16x8 load_kperm_frag<bf16>(const bf16* row, int kb, int s, int h) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row + kb + 16 * s + 4 * h);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(row + kb + 16 * s + 4 * h + 8);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
template <> __device__ __forceinline__ f32x4 load_kperm_frag<float>(const float* row, int kb, int s, int h) {
    return *reinterpret_cast<const f32x4*>(row + kb + 8 * s + 4 * h);
}

// The same permuted-k fragment, dtype dispatched:
//   bf16: straight from the ROW-MAJOR tile [k][cols] with two ds_read_b64_tr_b16 (each returns 4 consecutive k rows of this
//         lane's column) -- no transposed copy of the tile is ever staged;
//   fp32: from a transposed tile [col][k] (staged with element-wise LDS stores; parity path only).
template <typename T>
__device__ __forceinline__ typename Frag<T>::type kperm_frag(const T* rowmajor, int prow, const T* transposed, int ptr_, int kb, int s,
                                                              int col0, int lane) {
    if constexpr (std::is_same<T, bf16>::value) {
        typedef __attribute__((address_space(3))) bf16x4 LdsV4;
        const int q = (lane & 15) >> 2, col = col0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4, k = kb + 16 * s + 4 * (lane >> 5) + q;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(rowmajor + k * prow + col));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(rowmajor + (k + 8) * prow + col));
        const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    } else {
        return load_kperm_frag<T>(transposed + (col0 + (lane & 31)) * ptr_, kb, s, lane >> 5);
    }
}

template <typename T> __device__ __forceinline__ typename Frag<T>::type acc_to_frag(const f32x16& acc, int s) {
    typename Frag<T>::type f;
#pragma unroll
    for (int j = 0; j < Frag<T>::N; ++j) f[j] = from_f32<T>(acc[s * Frag<T>::N + j]);
    return f;
}

// Register-staged tile of NR rows x HD: load() issues the global reads, store() / store_t() commit them to LDS row-major /
// transposed.  The kernels load tile t+1 right after the barrier that publishes tile t, so the HBM latency of the next tile
// is hidden behind the MFMA / softmax work on the current one.  Rows at or beyond nrows read the LAST VALID row instead
// (finite data; every consumer masks those rows' scores): the loads carry no per-lane condition -- a conditional load
// compiles to an exec-masked branch per chunk and pessimistic waits behind it.
template <typename T, int HD, int NR> struct RowTile {
    typedef typename Frag<T>::type F;
    static constexpr int VEC = Frag<T>::N, CPR = HD / VEC, NCH = (NR * CPR) / 256;
    static_assert((NR * CPR) % 256 == 0, "every thread owns the same number of 16-byte chunks");
    F r[NCH];
    __device__ __forceinline__ void load(const T* src, long ld, int r0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256, row = c / CPR, kc = (c % CPR) * VEC;
            r[i] = *reinterpret_cast<const F*>(src + (long)min(r0 + row, nrows - 1) * ld + kc);
        }
    }
    template <int P> __device__ __forceinline__ void store(T* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256, row = c / CPR, kc = (c % CPR) * VEC;
            *reinterpret_cast<F*>(lds + row * P + kc) = r[i];
        }
    }
    template <int P> __device__ __forceinline__ void store_t(T* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256, row = c / CPR, kc = (c % CPR) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) lds[(kc + e) * P + row] = r[i][e];
        }
    }
};

// Visibility (length limits, causal / window band, CrossAttention's block mask) as a per-lane RANGE, computed once per kernel:
// the keys a query row sees are [lo, lo + span) (lane = query: forward, dQ); the queries that see a key are such a range too
// (lane = key: dK/dV).  A boundary tile then tests  (unsigned)(index - lo) < span  per score: no branches.
__device__ __forceinline__ void visible_keys(const AttnArgs& a, int q, int lq, int lkv, int& lo, unsigned& span) {
    int hi = q < a.T ? a.S : 0;
    lo = 0;
    if (a.causal) {
        hi = min(hi, q + 1);
        if (a.window > 0 && a.window < a.T) lo = max(0, q - a.window);
    }
    if (lq >= 0 && q >= lq) hi = min(hi, lkv);
    span = (unsigned)max(hi - lo, 0);
}
__device__ __forceinline__ void visible_queries(const AttnArgs& a, int key, int lq, int lkv, int& lo, unsigned& span) {
    int hi = key < a.S ? a.T : 0;
    lo = 0;
    if (a.causal) {
        lo = key;
        if (a.window > 0 && a.window < a.T) hi = min(hi, key + a.window + 1);
    }
    if (lq >= 0 && key >= lkv) hi = min(hi, lq);
    span = (unsigned)max(hi - lo, 0);
}
// Attention-probability dropout mask (nn.MultiheadAttention dropout, decoder.py:91): ONE 7-op multiply-xorshift hash of the
// pair index (q, key >> 1), keyed per (seed, b, h), decides two adjacent keys with 16 bits each (keep iff bits >= p * 2^16).
// The bits are a pure function of (seed, b, h, q, key); attn_dropout_words_kernel evaluates it once per (layer, step) into
// the word layout the three kernels consume (1 bit per score).  The 1/(1-p) rescale is folded out of the per-score code
// (applied to O / dV / dQ / dK).
__device__ __forceinline__ uint32_t attn_bh_key(const AttnArgs& a, int b, int h) {
    return hash32((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)(b * a.H + h));
}
// 32 random bits for the key pair (key & ~1, key | 1) of query q: low half = even key, high half = odd key
__device__ __forceinline__ uint32_t attn_rand2(uint32_t bh_key, uint32_t pair_idx) {
    uint32_t x = pair_idx ^ bh_key;
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 16;
    return x;
}
// thr32 = threshold << 16.  High half: (x >> 16) >= t  <=>  x >= t << 16; low half: shift it up first.
__device__ __forceinline__ bool attn_keep_lo(uint32_t x, uint32_t thr32) { return (x << 16) >= thr32; }
__device__ __forceinline__ bool attn_keep_hi(uint32_t x, uint32_t thr32) { return x >= thr32; }

// ------------------------------------------------------------------------------------------------
// Vector-instruction budget.  The kernels below are bound by the VALU issue rate, not by the matrix pipe (hd = 64: a lane owns 2
// scores per MFMA), so everything that can leave the per-score vector code does:
//   * the softmax scale * log2(e) is folded into the Q (forward, dQ) / K (dK, dV) fragments once per kernel;
//   * the additive terms of a score -- key bias, minus the row's reference maximum (forward) or log-sum-exp (backward), minus
//     delta / c for dP -- ride on ONE extra k-step of the score's MFMA chain ("augmented k-step"): side X carries a value in
//     two bf16 slots (hi + lo = the fp32 value to 2^-17; fp32 mode: one exact slot) against unit slots of side Y and vice versa,
//     so the accumulator leaves the chain as  s * scale * log2 e + bias - reference  and the only vector work on a score is the exp2;
//   * forward: the reference maximum is LAGGED (it moves only when a tile's maximum exceeds it by more than 2^THR, a
//     wave-uniform rare branch), so no subtraction and no accumulator rescale in the common tile;
//   * attention-probability dropout: the keep bits are generated ONCE per (layer, step) by attn_dropout_words_kernel in the
//     accumulator's own lane layout -- one 64-bit word per (32 queries, register) -- and the kernels apply them with one
//     v_cndmask per score whose mask operand is that word in an SGPR pair (forward, dQ: scalar loads) or, in the key-per-lane
//     dK/dV kernel, from the same words read as one 32-bit column per lane (v_bfe + v_and / v_bfi).
template <typename T> struct Aug;
template <> struct Aug<bf16> {
    static __device__ __forceinline__ void split(float v, bf16& hi, bf16& lo) {
        hi = (bf16)v;
        const float r = v - (float)hi;
        lo = (r == r) ? (bf16)r : (bf16)0.f;                  // v = +-inf: hi carries it, inf - inf = NaN is dropped
    }
    static __device__ __forceinline__ bf16x8 x(float v) {
        bf16 hi, lo; split(v, hi, lo);
        const bf16 one = (bf16)1.f, z = (bf16)0.f;
        const bf16x8 f = {hi, lo, one, one, z, z, z, z};
        return f;
    }
    static __device__ __forceinline__ bf16x8 y(float v) {
        bf16 hi, lo; split(v, hi, lo);
        const bf16 one = (bf16)1.f, z = (bf16)0.f;
        const bf16x8 f = {one, one, hi, lo, z, z, z, z};
        return f;
    }
};
template <> struct Aug<float> {
    static __device__ __forceinline__ f32x4 x(float v) { const f32x4 f = {v, 1.f, 0.f, 0.f}; return f; }
    static __device__ __forceinline__ f32x4 y(float v) { const f32x4 f = {1.f, v, 0.f, 0.f}; return f; }
};
// x . y over the augmented k-step = x's value + y's value; only the lanes of the lower k half (lane < 32) carry the slots.

// Both halves of the wave (lanes i and i + 32 hold the two k-halves of one row): v_permlane32_swap instead of a ds_bpermute
// shuffle -- an LDS-pipe instruction would make the kernel wait on lgkmcnt, i.e. on the scalar dropout-word loads in flight.
__device__ __forceinline__ float max_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const unsigned lo = r[0], hi = r[1];                        // the lower half's value / the upper half's value, in every lane
    return fmaxf(__uint_as_float(lo), __uint_as_float(hi));
}

// One score under its dropout bit: mask = the 64-bit word of this accumulator register (bit = lane)
__device__ __forceinline__ float keep_or_zero(float x, uint64_t mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}
__device__ __forceinline__ float keep_or(float x, float alt, uint64_t mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(alt), "v"(x), "s"(mask));
    return r;
}
// Dropout words: [B*H][ceil(T/32)][ceil(S/64)][32] 64-bit words; word (mb, r) of a (32-query, 64-key) tile holds, at bit
// (query & 31) + 32 * hh, the keep bit of key  tile*64 + mb*32 + acc_row(r, hh).
// (indices are clamped to the last block: waves whose rows lie beyond T / S read valid words and discard the result)
__device__ __forceinline__ long drop_word_base(const AttnArgs& a, int bh, int qb32, int kt) {
    const int nqb = (a.T + 31) >> 5, nkt = (a.S + 63) >> 6;
    return (((long)bh * nqb + min(qb32, nqb - 1)) * nkt + min(kt, nkt - 1)) * 32;
}

// ------------------------------------------------------------------------------------------------
// Forward.  grid = (ceil(T/128), H, B); wave w owns query rows q0 + 32w .. +31; KV tiles of 64 keys.
// SPLITW (T <= 32: KV-cached greedy decode, one query row): the four waves would own the same 32 rows, so they split the KEYS
// instead -- 256 keys are staged per step, wave w takes keys [64w, 64w+64) of them -- and their (max, sum, O) partials are
// merged through LDS at the end.  Same arithmetic per score, a quarter of the serial tile walk.
template <typename T, int HD, bool SPLITW, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a, const uint64_t* __restrict__ dmask) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = ACfg<T>::VEC, NFR = ACfg<T>::NFR, KS = KStep<T>::value;
    constexpr int NKS = HD / KS, NDB = HD / 32, BKV = 64;
    constexpr int NTL = SPLITW ? 4 : 1, BST = BKV * NTL;        // keys staged per step
    constexpr int PK = HD + VEC;          // Ks pitch
    constexpr int PV = BST + 4;   // Vt pitch: 136 B (bf16) / 272 B (fp32) -> conflict-free permuted reads
    constexpr float THR = 8.f;            // the reference maximum of a row moves when a tile exceeds it by 2^THR
    constexpr bool TRD = std::is_same<T, bf16>::value;          // bf16: V is staged row-major and read with tr reads
    // bf16, query-per-wave form: the staged tiles are DOUBLE-BUFFERED -- tile t+1 is committed to the other buffer before the math on
    // tile t, so one workgroup barrier per tile orders both "t+1 is complete" and "everybody is done with t" (was two)
    constexpr int NBUF = (TRD && !SPLITW) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) T Ks_[NBUF][BST * PK];
    __shared__ __attribute__((aligned(16))) T Vt[TRD ? 8 : HD * PV];
    __shared__ __attribute__((aligned(16))) T Vs_[NBUF][TRD ? BST * PK : 8];
    __shared__ __attribute__((aligned(16))) F Ka_[NBUF][BST + 1];      // augmented k-step, key side: the key bias; entry BST = zeros

    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx.x = key split (SPLITW: one query block) or query block + nqb * key split
    const int nqb = SPLITW ? 1 : (a.T + 127) / 128;
    const int ksplit = blockIdx.x / nqb, q0 = (blockIdx.x % nqb) * 128;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qw0 = q0 + (SPLITW ? 0 : wave * 32);              // first query row of this wave
    const int q = qw0 + (lane & 31);
    const int kboff = SPLITW ? wave * BKV : 0;                  // this wave's keys inside the staged block
    const T* Q = (const T*)a.q + (long)b * a.bsq + h * HD;
    const T* K = (const T*)a.k + (long)b * a.bsk + h * HD;
    const T* V = (const T*)a.v + (long)b * a.bsv + h * HD;

    const float sc2 = a.scale * LOG2E;
    F qf[NKS];                                                  // Q * scale * log2 e
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const F raw = q < a.T ? *reinterpret_cast<const F*>(Q + (long)q * a.ldq + ks * KS + hh * VEC) : frag_zero<T>();
#pragma unroll
        for (int e = 0; e < VEC; ++e) qf[ks][e] = from_f32<T>(to_f32(raw[e]) * sc2);
    }
    int lq = -1, lkv = 0;
    if (a.blk_lq) { const int bb = (b * a.H + h) % a.B; lq = a.blk_lq[bb]; lkv = a.blk_lkv[bb]; }
    int vis_lo; unsigned vis_span;
    visible_keys(a, q, lq, lkv, vis_lo, vis_span);

    f32x16 acc_o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    // m_ref: the row's reference maximum (log2 domain; exactly representable in T, so that one slot of the augmented k-step
    // subtracts it without rounding); meaningless until m_set
    float m_ref = 0.f, l_run = 0.f;
    bool m_set = false;
    F qa = hh ? frag_zero<T>() : Aug<T>::y(0.f);
    const bool win_on = a.window > 0 && a.window < a.T;

    int kv_beg = 0, kv_end = a.S;
    if (a.causal) {
        kv_end = min(a.S, q0 + 128);
        if (a.window > 0 && a.window < a.T) kv_beg = max(0, q0 - a.window) / BKV * BKV;
    }
    if (a.nsplit > 1) { kv_beg = max(kv_beg, ksplit * a.split_len); kv_end = min(kv_end, (ksplit + 1) * a.split_len); }
    RowTile<T, HD, BST> kt, vt;
    float bias_r = 0.f;
    auto prefetch = [&](int kvb) {
        kt.load(K, a.ldk, kvb, a.S, tid);
        vt.load(V, a.ldv, kvb, a.S, tid);
        if (tid < BST) bias_r = a.key_bias ? a.key_bias[(long)b * a.S + min(kvb + tid, a.S - 1)] * LOG2E : 0.f;     // keys >= S: masked
    };
    if (tid < NBUF) Ka_[tid][BST] = frag_zero<T>();
    auto commit = [&](int buf) {
        kt.template store<PK>(Ks_[buf], tid);
        if constexpr (TRD) vt.template store<PK>(Vs_[buf], tid);
        else vt.template store_t<PV>(Vt, tid);
        if (tid < BST) Ka_[buf][tid] = Aug<T>::x(bias_r);
    };
    if (kv_beg < kv_end) prefetch(kv_beg);
    if constexpr (NBUF == 2) {
        if (kv_beg < kv_end) {
            commit(0);
            if (kv_beg + BST < kv_end) prefetch(kv_beg + BST);
        }
        __syncthreads();
    }
    int buf = 0;
    for (int kvb = kv_beg; kvb < kv_end; kvb += BST) {
        if constexpr (NBUF == 2) {
            // tile kvb sits complete in buffer `buf` (the barrier that ended the previous iteration); the next tile's registers go into
            // the other buffer, which everybody left at that barrier, and the loads of the tile after it start
            if (kvb + BST < kv_end) {
                commit(buf ^ 1);
                if (kvb + 2 * BST < kv_end) prefetch(kvb + 2 * BST);
            }
        } else {
            __syncthreads();
            commit(0);
            __syncthreads();
            if (kvb + BST < kv_end) prefetch(kvb + BST);
        }
        T* const Ks = Ks_[buf];
        T* const Vs = Vs_[buf];
        F* const Ka = Ka_[buf];
        const int kv0 = kvb + kboff;                            // first key of this wave's 64-key tile
        const uint64_t* wp = DROP ? dmask + drop_word_base(a, b * a.H + h, qw0 >> 5, min(kv0, a.S - 1) >> 6) : nullptr;

        // scores in the log2 domain, already relative to the row's reference:  s * scale * log2 e + bias - m_ref
        f32x16 st[2];
        {   // every LDS operand of the two score chains is requested before the first MFMA (one LDS latency, counted waits)
            F kfr[2][NKS + 1];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[mb][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    kfr[mb][ks] = *reinterpret_cast<const F*>(&Ks[(kboff + mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                kfr[mb][NKS] = Ka[hh ? BST : kboff + mb * 32 + (lane & 31)];
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NKS + 1), 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) mma32(st[mb], kfr[mb][ks], qf[ks]);
                mma32(st[mb], kfr[mb][NKS], qa);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * (NKS + 1) * ACfg<T>::MPI, 0);
        }
        // The tile's dropout words are requested HERE: scalar loads share lgkmcnt with the LDS reads and return out of order, so
        // the next wait on an LDS operand also waits for them -- behind the score chain the next LDS read is the first V
        // fragment, a whole softmax (~450 issue cycles) away, and the words land under the max / exp2 / sum code.
        uint64_t w0[16], w1[16];
        if constexpr (DROP) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) { w0[r] = wp[r]; w1[r] = wp[16 + r]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        // Tiles that are entirely visible for this wave's 32 query rows (the common case) skip every per-element mask test.
        const bool full = (kv0 + BKV <= a.S) && (qw0 + 32 <= a.T) && lq < 0 &&
                          (!a.causal || (kv0 + BKV - 1 <= qw0 && (!win_on || kv0 >= qw0 + 31 - a.window)));
        if (!full) {
            const int rel = kv0 + 4 * hh - vis_lo;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[mb][r] = (unsigned)(rel + mb * 32 + acc_row(r, 0)) < vis_span ? st[mb][r] : -INFINITY;
        }
        float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, st[0][r]), st[1][r]);
        mx = max_halves(mx);
        const bool fix = m_set ? (mx > THR) : (mx > -INFINITY);   // the same for both lanes of a row
        if (__builtin_amdgcn_ballot_w64(fix) != 0) {              // rare: first tile of a row, or its maximum grew past the threshold
            const float m2 = fix ? to_f32(from_f32<T>(m_ref + mx)) : m_ref;
            const float delta = m2 - m_ref;
            const float alpha = (fix && m_set) ? __builtin_amdgcn_exp2f(-delta) : 1.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[mb][r] -= delta;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            m_ref = m2;
            m_set = m_set || fix;
            qa = hh ? frag_zero<T>() : Aug<T>::y(-m_ref);
        }
        float psum = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(st[mb][r]);          // exp2(-inf) = 0 for masked keys
                psum += pv;
                st[mb][r] = pv;
            }
        l_run += psum;
        if constexpr (DROP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[0][r] = keep_or_zero(st[0][r], w0[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) st[1][r] = keep_or_zero(st[1][r], w1[r]);
        }
        // O^T += V^T . P^T
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int s = 0; s < NFR; ++s) {
                const F pf = acc_to_frag<T>(st[mb], s);
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const F vf = kperm_frag<T>(Vs, PK, Vt, PV, kboff + mb * 32, s, d * 32, lane);
                    mma32(acc_o[d], vf, pf);
                }
            }
        if constexpr (NBUF == 2) { __syncthreads(); buf ^= 1; }
    }
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float m_fin = l_tot > 0.f ? m_ref : -INFINITY;              // a row that saw no finite score has no reference
    if constexpr (SPLITW) {
        // merge the four waves' partial softmaxes of the same 32 query rows (log2 domain): the staging tiles are dead
        __syncthreads();
        float* o_s = reinterpret_cast<float*>(Ks_[0]);              // [4][HD][32]
        float* m_s = o_s + 4 * HD * 32;                         // [4][32]
        float* l_s = m_s + 4 * 32;
        static_assert((4 * HD * 32 + 8 * 32) * sizeof(float) <= sizeof(T) * BST * PK, "merge scratch fits in the K tile");
        const int qi = lane & 31;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_s[(wave * HD + d * 32 + acc_row(r, lane)) * 32 + qi] = acc_o[d][r];
        if (hh == 0) { m_s[wave * 32 + qi] = m_fin; l_s[wave * 32 + qi] = l_tot; }
        __syncthreads();
        if (wave != 0) return;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, m_s[w * 32 + qi]);
        float wsc[4];
        l_tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = m_s[w * 32 + qi];
            wsc[w] = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mm);
            l_tot += l_s[w * 32 + qi] * wsc[w];
        }
        m_fin = mm;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += o_s[(w * HD + d * 32 + acc_row(r, lane)) * 32 + qi] * wsc[w];
                acc_o[d][r] = v;
            }
    }
    if (a.nsplit > 1) {          // partial softmax of this key split: un-normalised O, reference maximum (log2 domain) and sum
        if (q < a.T) {
            float* P = a.part + ((((long)b * a.H + h) * a.nsplit + ksplit) * a.T + q) * (HD + 2);
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) P[d * 32 + acc_row(r, lane)] = acc_o[d][r];
            if (hh == 0) { P[HD] = m_fin; P[HD + 1] = l_tot; }
        }
        return;
    }
    const float inv = l_tot > 0.f ? a.drop_scale / l_tot : 0.f;      // dropout rescale folded in (1 when p = 0)
    if (q < a.T) {
        T* O = (T*)a.o + (long)b * a.bso + (long)q * a.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[d * 32 + acc_row(r, lane)] = from_f32<T>(acc_o[d][r] * inv);
        if (hh == 0 && a.lse) a.lse[((long)b * a.H + h) * a.T + q] = (l_tot > 0.f) ? (m_fin + log2f(l_tot)) * LN2 : -INFINITY;
    }
}

// Merge of the key-split partials (same arithmetic as the in-kernel merge of the four waves): one 64-thread block per
// (b, h, q), thread = output channel.
template <typename T, int HD>
__global__ __launch_bounds__(64) void attn_split_merge_kernel(AttnArgs a) {
    const int q = blockIdx.x % a.T, bh = blockIdx.x / a.T, h = bh % a.H, b = bh / a.H, dch = threadIdx.x;
    const float* P = a.part + (((long)bh * a.nsplit) * a.T + q) * (HD + 2);
    const long pstride = (long)a.T * (HD + 2);
    float mm = -INFINITY;
    for (int j = 0; j < a.nsplit; ++j) mm = fmaxf(mm, P[j * pstride + HD]);
    float l_tot = 0.f, o = 0.f;
    for (int j = 0; j < a.nsplit; ++j) {
        const float mj = P[j * pstride + HD];
        const float wj = mj == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mj - mm);
        l_tot += P[j * pstride + HD + 1] * wj;
        if (dch < HD) o += P[j * pstride + dch] * wj;
    }
    const float inv = l_tot > 0.f ? a.drop_scale / l_tot : 0.f;
    if (dch < HD) ((T*)a.o)[(long)b * a.bso + (long)q * a.ldo + h * HD + dch] = from_f32<T>(o * inv);
    if (dch == 0 && a.lse) a.lse[((long)b * a.H + h) * a.T + q] = (l_tot > 0.f) ? (mm + log2f(l_tot)) * LN2 : -INFINITY;
}

// ------------------------------------------------------------------------------------------------
// Backward, dQ.  Same orientation as the forward: lane = query row.  c = 1 / (1 - p_drop), M = keep mask:
//   S'^T = K Q~^T + bias - lse  (augmented k-step) ; P^T = exp2(S'^T) ; dP'^T = V dO^T - delta / c  (augmented k-step)
//   dS^T / c = P^T o (M ? dP'^T : -delta / c) ; dQ^T += K^T dS^T / c ; dQ = scale * c * dQ^T
template <typename T, int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a, const uint64_t* __restrict__ dmask) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = ACfg<T>::VEC, NFR = ACfg<T>::NFR, KS = KStep<T>::value;
    constexpr int NKS = HD / KS, NDB = HD / 32, BKV = 64;
    constexpr int PK = HD + VEC;
    constexpr int PV = BKV + 4;
    constexpr bool TRD = std::is_same<T, bf16>::value;          // bf16: K^T fragments are tr reads of the row-major Ks tile
    constexpr int NBUF = TRD ? 2 : 1;                           // bf16: double-buffered tiles, one workgroup barrier per tile (see the forward)
    __shared__ __attribute__((aligned(16))) T Ks_[NBUF][BKV * PK];
    __shared__ __attribute__((aligned(16))) T Vs_[NBUF][BKV * PK];
    __shared__ __attribute__((aligned(16))) T Kt[TRD ? 8 : HD * PV];
    __shared__ __attribute__((aligned(16))) F Ka_[NBUF][BKV + 1];      // augmented k-step, key side: the key bias; entry BKV = zeros

    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (a.T + 127) / 128;                          // blockIdx.x = query block + nqb * key split
    const int ksplit = blockIdx.x / nqb, q0 = (blockIdx.x % nqb) * 128;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qw0 = q0 + wave * 32;
    const int q = qw0 + (lane & 31);
    const T* Q = (const T*)a.q + (long)b * a.bsq + h * HD;
    const T* K = (const T*)a.k + (long)b * a.bsk + h * HD;
    const T* V = (const T*)a.v + (long)b * a.bsv + h * HD;
    const T* DO = (const T*)a.dout + (long)b * a.bsdo + h * HD;

    const float sc2 = a.scale * LOG2E;
    F qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const F raw = q < a.T ? *reinterpret_cast<const F*>(Q + (long)q * a.ldq + ks * KS + hh * VEC) : frag_zero<T>();
#pragma unroll
        for (int e = 0; e < VEC; ++e) qf[ks][e] = from_f32<T>(to_f32(raw[e]) * sc2);
        dof[ks] = q < a.T ? *reinterpret_cast<const F*>(DO + (long)q * a.lddo + ks * KS + hh * VEC) : frag_zero<T>();
    }
    const long sidx = ((long)b * a.H + h) * a.T + q;
    float lse2 = q < a.T ? a.lse[sidx] * LOG2E : 0.f;
    if (!(lse2 > -INFINITY)) lse2 = 0.f;                        // a row without a visible key: every P is zeroed by its mask below
    // delta[q] = sum_d dO[q][d] * O[q][d] is formed HERE (the two lanes of a row hold the two halves of its d values) and stored for
    // the dK/dV kernel that follows on the stream -- a separate pass over O and dO was a launch of its own per layer
    float dlt = 0.f;
    {
        const T* O = (const T*)a.o + (long)b * a.bso + h * HD;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const F of = q < a.T ? *reinterpret_cast<const F*>(O + (long)q * a.ldo + ks * KS + hh * VEC) : frag_zero<T>();
#pragma unroll
            for (int e = 0; e < VEC; ++e) dlt += to_f32(of[e]) * to_f32(dof[ks][e]);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (hh == 0 && q < a.T && ksplit == 0) const_cast<float*>(a.delta)[sidx] = dlt;
    }
    const float ndc = -dlt / a.drop_scale;                      // -delta / c
    const bool win_on = a.window > 0 && a.window < a.T;
    int lq = -1, lkv = 0;
    if (a.blk_lq) { const int bb = (b * a.H + h) % a.B; lq = a.blk_lq[bb]; lkv = a.blk_lkv[bb]; }
    int vis_lo; unsigned vis_span;
    visible_keys(a, q, lq, lkv, vis_lo, vis_span);
    const F qa = hh ? frag_zero<T>() : Aug<T>::y(-lse2);        // query side of the score chain: minus the row's log-sum-exp
    const F da = hh ? frag_zero<T>() : Aug<T>::y(ndc);          // dO side of the dP chain: minus delta / c
    const F va = hh ? frag_zero<T>() : Aug<T>::x(0.f);          // V side of the dP chain: the unit slots

    f32x16 acc_q[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_q[d][r] = 0.f;

    int kv_beg = 0, kv_end = a.S;
    if (a.causal) {
        kv_end = min(a.S, q0 + 128);
        if (a.window > 0 && a.window < a.T) kv_beg = max(0, q0 - a.window) / BKV * BKV;
    }
    if (a.nsplit > 1) { kv_beg = max(kv_beg, ksplit * a.split_len); kv_end = min(kv_end, (ksplit + 1) * a.split_len); }
    RowTile<T, HD, BKV> kt, vt;
    float bias_r = 0.f;
    auto prefetch = [&](int kv0) {
        kt.load(K, a.ldk, kv0, a.S, tid);
        vt.load(V, a.ldv, kv0, a.S, tid);
        if (tid < BKV) bias_r = a.key_bias ? a.key_bias[(long)b * a.S + min(kv0 + tid, a.S - 1)] * LOG2E : 0.f;     // keys >= S: masked
    };
    if (tid < NBUF) Ka_[tid][BKV] = frag_zero<T>();
    auto commit = [&](int bi) {
        kt.template store<PK>(Ks_[bi], tid);
        vt.template store<PK>(Vs_[bi], tid);
        if constexpr (!TRD) kt.template store_t<PV>(Kt, tid);
        if (tid < BKV) Ka_[bi][tid] = Aug<T>::x(bias_r);
    };
    if (kv_beg < kv_end) prefetch(kv_beg);
    if constexpr (NBUF == 2) {
        if (kv_beg < kv_end) {
            commit(0);
            if (kv_beg + BKV < kv_end) prefetch(kv_beg + BKV);
        }
        __syncthreads();
    }
    int buf = 0;
    for (int kv0 = kv_beg; kv0 < kv_end; kv0 += BKV) {
        if constexpr (NBUF == 2) {
            if (kv0 + BKV < kv_end) {
                commit(buf ^ 1);
                if (kv0 + 2 * BKV < kv_end) prefetch(kv0 + 2 * BKV);
            }
        } else {
            __syncthreads();
            commit(0);
            __syncthreads();
            if (kv0 + BKV < kv_end) prefetch(kv0 + BKV);
        }
        T* const Ks = Ks_[buf];
        T* const Vs = Vs_[buf];
        F* const Ka = Ka_[buf];
        const bool full = (kv0 + BKV <= a.S) && (qw0 + 32 <= a.T) && lq < 0 &&
                          (!a.causal || (kv0 + BKV - 1 <= qw0 && (!win_on || kv0 >= qw0 + 31 - a.window)));
        const uint64_t* wp = DROP ? dmask + drop_word_base(a, b * a.H + h, qw0 >> 5, kv0 >> 6) : nullptr;
        f32x16 st[2], dp[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[mb][r] = 0.f; dp[mb][r] = 0.f; }
            if constexpr (TRD) {   // the block's LDS operands first, then its two chains (one LDS latency per block, counted waits)
                F kfr[NKS + 1], vfr[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    kfr[ks] = *reinterpret_cast<const F*>(&Ks[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                    vfr[ks] = *reinterpret_cast<const F*>(&Vs[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                }
                kfr[NKS] = Ka[hh ? BKV : mb * 32 + (lane & 31)];
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NKS + 1, 0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    mma32(st[mb], kfr[ks], qf[ks]);
                    mma32(dp[mb], vfr[ks], dof[ks]);
                }
                mma32(st[mb], kfr[NKS], qa);
                mma32(dp[mb], va, da);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NKS + 2, 0);
            } else {               // fp32 (parity mode): the registers do not hold twice the fragments
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const F kf = *reinterpret_cast<const F*>(&Ks[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                    mma32(st[mb], kf, qf[ks]);
                    const F vf = *reinterpret_cast<const F*>(&Vs[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                    mma32(dp[mb], vf, dof[ks]);
                }
                mma32(st[mb], Ka[hh ? BKV : mb * 32 + (lane & 31)], qa);
                mma32(dp[mb], va, da);
            }
        }
        // dropout words: requested behind the last LDS operand of the score / dP chains (see the forward kernel), consumed
        // behind the 32 exp2
        uint64_t w0[16], w1[16];
        if constexpr (DROP) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) { w0[r] = wp[r]; w1[r] = wp[16 + r]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        // P^T = exp2(score - lse) (0 where masked); wave-uniform branches keep the common tile free of mask tests
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[mb][r] = __builtin_amdgcn_exp2f(st[mb][r]);
        if (!full) {
            const int rel = kv0 + 4 * hh - vis_lo;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[mb][r] = (unsigned)(rel + mb * 32 + acc_row(r, 0)) < vis_span ? st[mb][r] : 0.f;
        }
        if constexpr (DROP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[0][r] = keep_or(dp[0][r], ndc, w0[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[1][r] = keep_or(dp[1][r], ndc, w1[r]);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[mb][r] *= dp[mb][r];                 // dS^T / c
            if constexpr (TRD) {
                F ktf[NFR][NDB];
#pragma unroll
                for (int s = 0; s < NFR; ++s)
#pragma unroll
                    for (int d = 0; d < NDB; ++d) ktf[s][d] = kperm_frag<T>(Ks, PK, Kt, PV, mb * 32, s, d * 32, lane);
                __builtin_amdgcn_sched_group_barrier(0x100, NFR * NDB * ACfg<T>::RPF, 0);
#pragma unroll
                for (int s = 0; s < NFR; ++s) {
                    const F sf = acc_to_frag<T>(st[mb], s);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) mma32(acc_q[d], ktf[s][d], sf);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NFR * NDB, 0);
            } else {
#pragma unroll
                for (int s = 0; s < NFR; ++s) {
                    const F sf = acc_to_frag<T>(st[mb], s);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        const F kf = kperm_frag<T>(Ks, PK, Kt, PV, mb * 32, s, d * 32, lane);
                        mma32(acc_q[d], kf, sf);
                    }
                }
            }
        }
        if constexpr (NBUF == 2) { __syncthreads(); buf ^= 1; }
    }
    const float osc = a.scale * a.drop_scale;
    if (q < a.T) {
        if (a.nsplit > 1) {          // partial dQ of this key split, fp32 [nsplit][B][T][H*HD]: summed by attn_dq_sum_kernel
            float* PQ = a.part + ((((long)ksplit * a.B + b) * a.T + q) * a.H + h) * HD;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) PQ[d * 32 + acc_row(r, lane)] = acc_q[d][r] * osc;
        } else {
            T* DQ = (T*)a.dq + (long)b * a.bsdq + (long)q * a.lddq + h * HD;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) DQ[d * 32 + acc_row(r, lane)] = from_f32<T>(acc_q[d][r] * osc);
        }
    }
}

// dq[b][q][c] = sum over key splits of the fp32 partials (fixed order), c over the H*HD channels
template <typename T>
__global__ void attn_dq_sum_kernel(AttnArgs a, int hd) {
    const long per = (long)a.B * a.T * a.H * hd, cols = (long)a.H * hd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < a.nsplit; ++j) s += a.part[j * per + i];
        const long c = i % cols, bq = i / cols, qq = bq % a.T, bb = bq / a.T;
        ((T*)a.dq)[bb * a.bsdq + qq * a.lddq + c] = from_f32<T>(s);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, dK and dV.  grid = (ceil(S/128), H, B); wave w owns keys k0 + 32w .. +31 (lane = key), queries
// on the register axis:  S' = Q K~^T + bias - lse ; P = exp2(S') ; dP' = dO V^T - delta / c  (both by augmented k-steps)
//   dV += (M o P)^T dO      dK += (dS / c)^T Q,  dS / c = P o (M ? dP' : -delta / c)        (A operand straight from accumulator registers)
// K~ = K * scale * log2 e is rounded to T here while the forward rounds Q * scale * log2 e: in bf16 the recomputed P differs from
// the forward's by the two roundings (a few 1e-3 relative, the size of P's own bf16 rounding); exact in fp32.
template <typename T, int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs a, const uint64_t* __restrict__ dmask) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = ACfg<T>::VEC, NFR = ACfg<T>::NFR, KS = KStep<T>::value;
    constexpr int NKS = HD / KS, NDB = HD / 32, BQ = 64;
    constexpr int PK = HD + VEC;
    constexpr int PT = BQ + 4;   // transposed tiles [d][q]
    constexpr bool TRD = std::is_same<T, bf16>::value;          // bf16: Q^T / dO^T fragments are tr reads of Qs / Ds
    constexpr int NBUF = TRD ? 2 : 1;                           // bf16: double-buffered tiles, one workgroup barrier per tile (see the forward)
    __shared__ __attribute__((aligned(16))) T Qs_[NBUF][BQ * PK];
    __shared__ __attribute__((aligned(16))) T Ds_[NBUF][BQ * PK];
    __shared__ __attribute__((aligned(16))) T Qt[TRD ? 8 : HD * PT];
    __shared__ __attribute__((aligned(16))) T Dt[TRD ? 8 : HD * PT];
    __shared__ __attribute__((aligned(16))) F Qa_[NBUF][BQ + 1];       // augmented k-step, query side of the score chain: -lse; entry BQ = zeros
    __shared__ __attribute__((aligned(16))) F Da_[NBUF][BQ + 1];       // ... of the dP chain: -delta / c
    __shared__ __attribute__((aligned(16))) float ndc_s_[NBUF][BQ];    // -delta / c per query row (the value a dropped score takes)

    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 128;
    const int kw0 = k0 + wave * 32;
    const int key = kw0 + (lane & 31);
    const T* Q = (const T*)a.q + (long)b * a.bsq + h * HD;
    const T* K = (const T*)a.k + (long)b * a.bsk + h * HD;
    const T* V = (const T*)a.v + (long)b * a.bsv + h * HD;
    const T* DO = (const T*)a.dout + (long)b * a.bsdo + h * HD;

    const float sc2 = a.scale * LOG2E;
    F kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const F raw = key < a.S ? *reinterpret_cast<const F*>(K + (long)key * a.ldk + ks * KS + hh * VEC) : frag_zero<T>();
#pragma unroll
        for (int e = 0; e < VEC; ++e) kf[ks][e] = from_f32<T>(to_f32(raw[e]) * sc2);
        vf[ks] = key < a.S ? *reinterpret_cast<const F*>(V + (long)key * a.ldv + ks * KS + hh * VEC) : frag_zero<T>();
    }
    const float kb2 = (a.key_bias && key < a.S) ? a.key_bias[(long)b * a.S + key] * LOG2E : 0.f;
    const F kya = hh ? frag_zero<T>() : Aug<T>::y(kb2);         // key side of the score chain: the key bias
    const F vya = hh ? frag_zero<T>() : Aug<T>::y(0.f);         // V side of the dP chain: the unit slots
    const bool win_on = a.window > 0 && a.window < a.T;
    int lq = -1, lkv = 0;
    if (a.blk_lq) { const int bb = (b * a.H + h) % a.B; lq = a.blk_lq[bb]; lkv = a.blk_lkv[bb]; }
    int vis_lo; unsigned vis_span;
    visible_queries(a, key, lq, lkv, vis_lo, vis_span);
    // dropout bits of this lane's key: one 32-bit column (bit = query & 31) of the (32-query, 64-key) tile's words
    const int ko = lane & 31, nqb32 = (a.T + 31) >> 5;
    const uint32_t* wcol = reinterpret_cast<const uint32_t*>(dmask) +
                           2 * (drop_word_base(a, b * a.H + h, 0, min(kw0, a.S - 1) >> 6) + ((kw0 >> 5) & 1) * 16 + (ko & 3) + 4 * (ko >> 3)) + ((ko >> 2) & 1);
    const long wq_stride = 2L * ((a.S + 63) >> 6) * 32;         // dwords between consecutive 32-query blocks

    f32x16 acc_k[NDB], acc_v[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_k[d][r] = 0.f; acc_v[d][r] = 0.f; }

    int q_beg = 0, q_end = a.T;
    if (a.causal) {
        q_beg = k0 / BQ * BQ;                                   // rows q < k0 never see these keys
        if (a.window > 0 && a.window < a.T) q_end = min(a.T, k0 + 128 + a.window);
    }
    const long sbase = ((long)b * a.H + h) * a.T;
    RowTile<T, HD, BQ> qt, dt;
    float lse_r = 0.f, ndc_r = 0.f;
    auto prefetch = [&](int q0) {
        qt.load(Q, a.ldq, q0, a.T, tid);
        dt.load(DO, a.lddo, q0, a.T, tid);
        if (tid < BQ) {                                         // rows >= T: masked
            const int qq = min(q0 + tid, a.T - 1);
            lse_r = a.lse[sbase + qq] * LOG2E;
            if (!(lse_r > -INFINITY)) lse_r = 0.f;              // a row without a visible key: every P is zeroed by its mask below
            ndc_r = -a.delta[sbase + qq] / a.drop_scale;
        }
    };
    if (tid < NBUF) { Qa_[tid][BQ] = frag_zero<T>(); Da_[tid][BQ] = frag_zero<T>(); }
    auto commit = [&](int bi) {
        qt.template store<PK>(Qs_[bi], tid);
        dt.template store<PK>(Ds_[bi], tid);
        if constexpr (!TRD) {
            qt.template store_t<PT>(Qt, tid);
            dt.template store_t<PT>(Dt, tid);
        }
        if (tid < BQ) { Qa_[bi][tid] = Aug<T>::x(-lse_r); Da_[bi][tid] = Aug<T>::x(ndc_r); ndc_s_[bi][tid] = ndc_r; }
    };
    if (q_beg < q_end) prefetch(q_beg);
    if constexpr (NBUF == 2) {
        if (q_beg < q_end) {
            commit(0);
            if (q_beg + BQ < q_end) prefetch(q_beg + BQ);
        }
        __syncthreads();
    }
    int buf = 0;
    for (int q0 = q_beg; q0 < q_end; q0 += BQ) {
        uint32_t wbits[BQ / 32];
        auto load_bits = [&]() {
#pragma unroll
            for (int mb = 0; mb < BQ / 32; ++mb) {              // requested ahead of the next tile's rows: vmcnt is in order
                const int qb32 = min((q0 >> 5) + mb, nqb32 - 1);
                wbits[mb] = DROP ? wcol[qb32 * wq_stride] >> (4 * hh) : 0u;
            }
        };
        if constexpr (NBUF == 2) {
            if (q0 + BQ < q_end) commit(buf ^ 1);
            load_bits();
            if (q0 + 2 * BQ < q_end) prefetch(q0 + 2 * BQ);
        } else {
            __syncthreads();
            commit(0);
            __syncthreads();
            load_bits();
            if (q0 + BQ < q_end) prefetch(q0 + BQ);
        }
        T* const Qs = Qs_[buf];
        T* const Ds = Ds_[buf];
        F* const Qa = Qa_[buf];
        F* const Da = Da_[buf];
        float* const ndc_s = ndc_s_[buf];
#pragma unroll
        for (int mb = 0; mb < BQ / 32; ++mb) {
            const int qb = q0 + mb * 32;
            if (qb >= q_end) break;                                     // block-uniform
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
            if constexpr (TRD) {   // every LDS operand of the two chains is requested before the first MFMA: one LDS latency per block, not one per MFMA
                F qfr[NKS + 1], dfr[NKS + 1];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    qfr[ks] = *reinterpret_cast<const F*>(&Qs[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                    dfr[ks] = *reinterpret_cast<const F*>(&Ds[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                }
                qfr[NKS] = Qa[hh ? BQ : mb * 32 + (lane & 31)];
                dfr[NKS] = Da[hh ? BQ : mb * 32 + (lane & 31)];
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NKS + 2, 0);      // the DS reads ...
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    mma32(st, qfr[ks], kf[ks]);
                    mma32(dp, dfr[ks], vf[ks]);
                }
                mma32(st, qfr[NKS], kya);
                mma32(dp, dfr[NKS], vya);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NKS + 2, 0);      // ... then the MFMAs
            } else {               // fp32 (parity mode): twice the fragments; the registers do not hold them all
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const F qf = *reinterpret_cast<const F*>(&Qs[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                    mma32(st, qf, kf[ks]);
                    const F df = *reinterpret_cast<const F*>(&Ds[(mb * 32 + (lane & 31)) * PK + ks * KS + hh * VEC]);
                    mma32(dp, df, vf[ks]);
                }
                mma32(st, Qa[hh ? BQ : mb * 32 + (lane & 31)], kya);
                mma32(dp, Da[hh ? BQ : mb * 32 + (lane & 31)], vya);
            }
            f32x16 pd;  // dropped probabilities (for dV)
            const bool full = (kw0 + 32 <= a.S) && (qb + 32 <= a.T) && lq < 0 &&
                              (!a.causal || (kw0 + 31 <= qb && (!win_on || kw0 >= qb + 31 - a.window)));
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r]);            // P
            if (!full) {
                const int rel = qb + 4 * hh - vis_lo;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = (unsigned)(rel + acc_row(r, 0)) < vis_span ? st[r] : 0.f;
            }
            if constexpr (DROP) {
                const uint32_t wb = wbits[mb];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 n4 = *reinterpret_cast<const f32x4*>(&ndc_s[mb * 32 + 8 * g + 4 * hh]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        // m = all ones where kept (v_bfe_i32); pd = P & m; t = kept ? dP' : -delta / c (v_bfi_b32); written as
                        // instructions: the compiler's own lowering of the same expressions took twice as many
                        const float pv = st[r], dv = dp[r], nv = n4[e];
                        uint32_t m;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(wb), "n"(8 * g + e));
                        pd[r] = __uint_as_float(__float_as_uint(pv) & m);
                        float t;
                        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(t) : "v"(m), "v"(dv), "v"(nv));
                        st[r] *= t;                                                                      // dS / c
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { pd[r] = st[r]; st[r] *= dp[r]; }
            }
            if constexpr (TRD) {
                F dtf[NFR][NDB], qtf[NFR][NDB];
#pragma unroll
                for (int s = 0; s < NFR; ++s)
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        dtf[s][d] = kperm_frag<T>(Ds, PK, Dt, PT, mb * 32, s, d * 32, lane);
                        qtf[s][d] = kperm_frag<T>(Qs, PK, Qt, PT, mb * 32, s, d * 32, lane);
                    }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NFR * NDB * ACfg<T>::RPF, 0);
#pragma unroll
                for (int s = 0; s < NFR; ++s) {
                    const F pf = acc_to_frag<T>(pd, s);
                    const F sf = acc_to_frag<T>(st, s);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        mma32(acc_v[d], pf, dtf[s][d]);
                        mma32(acc_k[d], sf, qtf[s][d]);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NFR * NDB, 0);
            } else {
#pragma unroll
                for (int s = 0; s < NFR; ++s) {
                    const F pf = acc_to_frag<T>(pd, s);
                    const F sf = acc_to_frag<T>(st, s);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        const F dtf = kperm_frag<T>(Ds, PK, Dt, PT, mb * 32, s, d * 32, lane);
                        mma32(acc_v[d], pf, dtf);
                        const F qtf = kperm_frag<T>(Qs, PK, Qt, PT, mb * 32, s, d * 32, lane);
                        mma32(acc_k[d], sf, qtf);
                    }
                }
            }
        }
        if constexpr (NBUF == 2) { __syncthreads(); buf ^= 1; }
    }
    // accumulators: column = d (lane & 31), row = key (register axis)
    const float ksc = a.scale * a.drop_scale;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + wave * 32 + acc_row(r, lane);
            if (kk >= a.S) continue;
            const int col = h * HD + d * 32 + (lane & 31);
            ((T*)a.dk)[(long)b * a.bsdk + (long)kk * a.lddk + col] = from_f32<T>(acc_k[d][r] * ksc);
            ((T*)a.dv)[(long)b * a.bsdv + (long)kk * a.lddv + col] = from_f32<T>(acc_v[d][r] * a.drop_scale);
        }
}

// Debug / test entry: materialise the keep-mask the three kernels above regenerate on the fly (1 = kept), one byte per score.
__global__ void attn_dropout_mask_kernel(unsigned char* __restrict__ out, AttnArgs a) {
    const long n = (long)a.B * a.H * a.T * a.S;
    const uint32_t s2 = (uint32_t)(a.S + 1) >> 1, thr32 = a.drop_thresh << 16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int key = (int)(i % a.S); long r = i / a.S;
        const int q = (int)(r % a.T); r /= a.T;
        const int h = (int)(r % a.H), b = (int)(r / a.H);
        const uint32_t x = attn_rand2(attn_bh_key(a, b, h), (uint32_t)q * s2 + (uint32_t)(key >> 1));
        out[i] = (a.drop_thresh == 0 || ((key & 1) ? attn_keep_hi(x, thr32) : attn_keep_lo(x, thr32))) ? 1 : 0;
    }
}

// The keep bits in the kernels' word layout (drop_word_base): one wave per (b, h, 32-query block, 64-key tile) evaluates the
// pair hash in the forward kernel's lane layout (lane = query + 32 * k-half, register = key) and turns each register's
// comparison into its 64-bit lane mask -- v_cmp writes exactly that word.
__global__ __launch_bounds__(256) void attn_dropout_words_kernel(uint64_t* __restrict__ out, AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5;
    const int nkt = (a.S + 63) >> 6, kt = blockIdx.x * 4 + wave;
    if (kt >= nkt) return;                                       // wave-uniform
    const int qb32 = blockIdx.y, bh = blockIdx.z;
    const uint32_t bh_key = hash32((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)bh);
    const uint32_t s2 = (uint32_t)(a.S + 1) >> 1, thr32 = a.drop_thresh << 16;
    const uint32_t pbase = (uint32_t)(qb32 * 32 + (lane & 31)) * s2 + (uint32_t)((kt * 64 + 4 * hh) >> 1);
    uint64_t mine = 0;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {     // registers 2j, 2j+1 of a block are adjacent keys: one hash per pair
            const uint32_t x = attn_rand2(bh_key, pbase + (uint32_t)((mb * 32 + acc_row(2 * j, 0)) >> 1));
            const uint64_t w0 = __builtin_amdgcn_ballot_w64(attn_keep_lo(x, thr32));
            const uint64_t w1 = __builtin_amdgcn_ballot_w64(attn_keep_hi(x, thr32));
            if (lane == mb * 16 + 2 * j) mine = w0;
            if (lane == mb * 16 + 2 * j + 1) mine = w1;
        }
    if (lane < 32) out[drop_word_base(a, bh, qb32, kt) + lane] = mine;
}

template <typename T, int HD> int run_fwd(const AttnArgs& a, hipStream_t s, bool merge = true) {
    // a single 32-row query block (KV-cached decode): split the keys over the waves (and, with a workspace, over workgroups).
    // Inference only: a training forward with dropout and T <= 32 takes the query-per-wave kernel below.
    if (a.T <= 32 && a.S > 64 && !a.drop_thresh) {
        const dim3 g(a.nsplit > 1 ? a.nsplit : 1, a.H, a.B);
        hipLaunchKernelGGL((attn_fwd_kernel<T, HD, true, false>), g, dim3(256), 0, s, a, a.dmask);
        if (a.nsplit > 1 && merge) hipLaunchKernelGGL((attn_split_merge_kernel<T, HD>), dim3(a.B * a.H * a.T), dim3(64), 0, s, a);
        OMR_CHECK_LAUNCH();
        return OMR_OK;
    }
    dim3 grid(cdiv(a.T, 128) * (a.nsplit > 1 ? a.nsplit : 1), a.H, a.B);
    if (a.drop_thresh) hipLaunchKernelGGL((attn_fwd_kernel<T, HD, false, true>), grid, dim3(256), 0, s, a, a.dmask);
    else hipLaunchKernelGGL((attn_fwd_kernel<T, HD, false, false>), grid, dim3(256), 0, s, a, a.dmask);
    if (a.nsplit > 1) hipLaunchKernelGGL((attn_split_merge_kernel<T, HD>), dim3(a.B * a.H * a.T), dim3(64), 0, s, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
template <typename T, int HD> int run_bwd(const AttnArgs& a, hipStream_t s) {
    const dim3 gq(cdiv(a.T, 128) * (a.nsplit > 1 ? a.nsplit : 1), a.H, a.B);
    if (a.drop_thresh) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, true>), gq, dim3(256), 0, s, a, a.dmask);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, false>), gq, dim3(256), 0, s, a, a.dmask);
    if (a.nsplit > 1) {
        long nsum = (long)a.B * a.T * a.H * HD, gs = (nsum + 255) / 256;
        hipLaunchKernelGGL((attn_dq_sum_kernel<T>), dim3((unsigned)(gs > 4096 ? 4096 : gs)), dim3(256), 0, s, a, HD);
    }
    const dim3 gk(cdiv(a.S, 128), a.H, a.B);
    if (a.drop_thresh) hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, true>), gk, dim3(256), 0, s, a, a.dmask);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, false>), gk, dim3(256), 0, s, a, a.dmask);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

int fill_common(AttnArgs& a, int B, int H, int T, int S, int hd, float dropout_p, unsigned long long seed, int causal, int window,
                const float* key_bias, const int* blk_lq, const int* blk_lkv, const unsigned long long* drop_words = nullptr, bool need_words = false) {
    if (B <= 0 || H <= 0 || T <= 0 || S <= 0) return OMR_ERR_ARG;
    if (hd != 32 && hd != 64) return OMR_ERR_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return OMR_ERR_ARG;
    if ((blk_lq == nullptr) != (blk_lkv == nullptr)) return OMR_ERR_ARG;
    a.B = B; a.H = H; a.T = T; a.S = S; a.scale = 1.0f / sqrtf((float)hd); a.causal = causal; a.window = window;
    a.key_bias = key_bias; a.blk_lq = blk_lq; a.blk_lkv = blk_lkv;
    a.drop_thresh = (uint32_t)((double)dropout_p * 65536.0 + 0.5);      // 16-bit threshold (attn_rand2); 0 = dropout off
    a.drop_scale = 1.f / (1.f - dropout_p);
    a.seed = seed;
    a.dmask = reinterpret_cast<const uint64_t*>(drop_words);
    if (need_words && a.drop_thresh != 0 && !drop_words) return OMR_ERR_ARG;      // the kernels read the keep bits, they do not hash
    return OMR_OK;
}

}  // namespace

extern "C" int omr_attn_dropout_mask(unsigned char* mask, int B, int H, int T, int S, float dropout_p, unsigned long long seed, void* stream) {
    AttnArgs a = {};
    int rc = fill_common(a, B, H, T, S, 64, dropout_p, seed, 0, -1, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (!mask) return OMR_ERR_ARG;
    const long n = (long)B * H * T * S;
    long g = (n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, mask, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" long omr_attn_dropout_words_count(int B, int H, int T, int S) {
    if (B <= 0 || H <= 0 || T <= 0 || S <= 0) return 0;
    return (long)B * H * ((T + 31) / 32) * ((S + 63) / 64) * 32;
}

extern "C" int omr_attn_dropout_words(unsigned long long* words, int B, int H, int T, int S, float dropout_p, unsigned long long seed, void* stream) {
    AttnArgs a = {};
    int rc = fill_common(a, B, H, T, S, 64, dropout_p, seed, 0, -1, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (!words || a.drop_thresh == 0) return OMR_ERR_ARG;
    const dim3 grid((unsigned)(((S + 63) / 64 + 3) / 4), (unsigned)((T + 31) / 32), (unsigned)(B * H));
    hipLaunchKernelGGL(attn_dropout_words_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint64_t*>(words), a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

// Key split of the lane-per-query kernels (forward, dQ).  They own 32 query rows per wave, so B*H*T/32 waves exist whatever
// the key count: 2 per SIMD at the benchmark's cross-attention (B 32, H 4, T 512, S 4096) -- too few to hide the LDS / barrier /
// load latencies of a 64-key tile (measured: 4 000 SIMD cycles per wave-tile against ~1 900 of issue).  Splitting the KEYS of
// a (batch, head, query block) over several workgroups multiplies the resident waves; the forward's partial softmaxes are
// merged by attn_split_merge_kernel, the partial dQ sums by attn_dq_sum_kernel (both fixed-order, no atomics).  Decode
// (T <= 32): one 256-key block per workgroup.  Causal attention is not split (its key range depends on the query block).
static void choose_split(int B, int H, int T, int S, int causal, int* nsplit, int* split_len) {
    *nsplit = 1; *split_len = 0;
    if (causal || S <= 256) return;
    int want;
    if (T <= 32) { want = (S + 255) / 256; if (want > 64) want = 64; }      // the merge prologue of omr_decode_linear takes <= 64 splits
    else {
        const long blocks = (long)B * H * ((T + 127) / 128);
        static const int min_wg = getenv("OMR_ATTN_MIN_WG") ? atoi(getenv("OMR_ATTN_MIN_WG")) : 512;      // experiment knob
        want = (int)((min_wg + blocks - 1) / blocks);               // at least ~512 workgroups (2 per CU); more buys nothing: the
                                                                  // kernels are VALU-issue bound, not latency bound (measured)
        const int maxs = S / 512;                                 // at least 512 keys per split
        if (want > maxs) want = maxs;
    }
    if (want <= 1) return;
    const int len = ((S + want - 1) / want + 255) / 256 * 256;    // whole 256-key staging blocks
    *split_len = len; *nsplit = (S + len - 1) / len;
    if (*nsplit <= 1) { *nsplit = 1; *split_len = 0; }
}

/* floats of scratch omr_attn_fwd_ws / omr_attn_bwd_ws want for a shape (0: the shape is not split) */
extern "C" long omr_attn_workspace_floats(int B, int H, int T, int S, int head_dim, int causal, int backward) {
    if (B <= 0 || H <= 0 || T <= 0 || S <= 0) return 0;
    int nsplit, len;
    choose_split(B, H, T, S, causal, &nsplit, &len);
    if (nsplit <= 1) return 0;
    return backward ? (long)nsplit * B * T * H * head_dim : (long)B * H * nsplit * T * (head_dim + 2);
}

static int attn_fwd_impl(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                         long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, int causal, int window,
                         const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p, unsigned long long seed,
                         const unsigned long long* drop_words, float* split_ws, long split_ws_floats, void* stream, int* nsplit_out = nullptr);

/* omr_attn_fwd with caller-provided scratch for the key split (omr_attn_workspace_floats(..., backward = 0) floats) */
extern "C" int omr_attn_fwd_ws(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                               long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, int causal, int window,
                               const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p, unsigned long long seed,
                               const unsigned long long* drop_words, float* ws, long ws_floats, void* stream) {
    return attn_fwd_impl(dtype, q, k, v, o, lse, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, T, S, head_dim, causal, window, key_bias, blk_lq, blk_lkv,
                         dropout_p, seed, drop_words, ws, ws_floats, stream);
}


extern "C" int omr_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                            long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, int causal, int window,
                            const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p, unsigned long long seed,
                            const unsigned long long* drop_words, void* stream) {
    return attn_fwd_impl(dtype, q, k, v, o, lse, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, T, S, head_dim, causal, window, key_bias, blk_lq, blk_lkv,
                         dropout_p, seed, drop_words, nullptr, 0, stream);
}

/* floats of split workspace omr_attn_fwd_split wants for (B, H, T <= 32, S): partial softmaxes of the key splits */
extern "C" long omr_attn_split_workspace_floats(int B, int H, int T, int S, int head_dim) {
    if (T > 32) return 0;
    return omr_attn_workspace_floats(B, H, T, S, head_dim, 0, 0);
}

/* omr_attn_fwd for a single block of at most 32 query rows (KV-cached decode) with the KEYS split over workgroups of 256 keys
 * each (flash-decoding): one query row otherwise keeps a (batch, head) pair on ONE workgroup that walks all S keys. */
extern "C" int omr_attn_fwd_split(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                                  long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, const float* key_bias,
                                  float* split_ws, long split_ws_floats, void* stream) {
    if (T > 32) return OMR_ERR_ARG;
    return attn_fwd_impl(dtype, q, k, v, o, lse, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, T, S, head_dim, 0, -1, key_bias, nullptr, nullptr, 0.f, 0,
                         nullptr, split_ws, split_ws_floats, stream);
}

static int attn_fwd_impl(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                         long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, int causal, int window,
                         const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p, unsigned long long seed,
                         const unsigned long long* drop_words, float* split_ws, long split_ws_floats, void* stream, int* nsplit_out) {
    AttnArgs a = {};
    int rc = fill_common(a, B, H, T, S, head_dim, dropout_p, seed, causal, window, key_bias, blk_lq, blk_lkv, drop_words, true);
    if (rc) return rc;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (ldq % vec || ldk % vec || ldv % vec || ldo % 4) return OMR_ERR_ARG;
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.bsq = bsq; a.bsk = bsk; a.bsv = bsv; a.bso = bso;
    a.nsplit = 1; a.split_len = 0; a.part = nullptr;
    if (split_ws) {
        int nsplit, len;
        choose_split(B, H, T, S, causal, &nsplit, &len);
        if (nsplit > 1) {
            if (split_ws_floats < (long)B * H * nsplit * T * (head_dim + 2)) return OMR_ERR_ARG;
            a.nsplit = nsplit; a.split_len = len; a.part = split_ws;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const bool merge = nsplit_out == nullptr;           // a caller that asks for the split count merges the partials itself
    if (nsplit_out) *nsplit_out = (T <= 32 && S > 64) ? a.nsplit : 1;
    if (!merge && !(T <= 32 && S > 64)) { a.nsplit = 1; a.split_len = 0; a.part = nullptr; }
    if (dtype == OMR_BF16) return head_dim == 64 ? run_fwd<bf16, 64>(a, s, merge) : run_fwd<bf16, 32>(a, s, merge);
    if (dtype == OMR_F32) return head_dim == 64 ? run_fwd<float, 64>(a, s, merge) : run_fwd<float, 32>(a, s, merge);
    return OMR_ERR_UNSUPPORTED;
}

/* omr_attn_fwd_split WITHOUT the merge pass: when the keys were split (*nsplit > 1) `o` is not written and split_ws holds, per
 * (b, h, split j), head_dim un-normalised outputs + running max (log2 domain) + sum at ((b*H + h)*nsplit + j)*T*(head_dim+2);
 * the consumer merges them (omr_decode_linear, prologue 3).  *nsplit = 1: `o` holds the finished rows. */
extern "C" int omr_attn_fwd_split_partials(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv,
                                           long ldo, long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim,
                                           float* split_ws, long split_ws_floats, int* nsplit, void* stream) {
    if (T > 32 || !nsplit) return OMR_ERR_ARG;
    return attn_fwd_impl(dtype, q, k, v, o, lse, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, T, S, head_dim, 0, -1, nullptr, nullptr, nullptr, 0.f, 0,
                         nullptr, split_ws, split_ws_floats, stream, nsplit);
}

extern "C" int omr_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                            float* delta_ws, void* dq, void* dk, void* dv, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq,
                            long lddk, long lddv, long bsq, long bsk, long bsv, long bso, long bsdo, long bsdq, long bsdk, long bsdv, int B,
                            int H, int T, int S, int head_dim, int causal, int window, const float* key_bias, const int* blk_lq,
                            const int* blk_lkv, float dropout_p, unsigned long long seed, const unsigned long long* drop_words, void* stream) {
    return omr_attn_bwd_ws(dtype, q, k, v, o, dout, lse, delta_ws, dq, dk, dv, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, bsq, bsk, bsv, bso, bsdo, bsdq,
                           bsdk, bsdv, B, H, T, S, head_dim, causal, window, key_bias, blk_lq, blk_lkv, dropout_p, seed, drop_words, nullptr, 0, stream);
}

/* omr_attn_bwd with caller-provided scratch for the key split of the dQ kernel (omr_attn_workspace_floats(..., backward = 1)) */
extern "C" int omr_attn_bwd_ws(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                               float* delta_ws, void* dq, void* dk, void* dv, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq,
                               long lddk, long lddv, long bsq, long bsk, long bsv, long bso, long bsdo, long bsdq, long bsdk, long bsdv, int B,
                               int H, int T, int S, int head_dim, int causal, int window, const float* key_bias, const int* blk_lq,
                               const int* blk_lkv, float dropout_p, unsigned long long seed, const unsigned long long* drop_words, float* ws,
                               long ws_floats, void* stream) {
    AttnArgs a = {};
    int rc = fill_common(a, B, H, T, S, head_dim, dropout_p, seed, causal, window, key_bias, blk_lq, blk_lkv, drop_words, true);
    if (rc) return rc;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (ldq % vec || ldk % vec || ldv % vec || lddo % vec) return OMR_ERR_ARG;
    if (!delta_ws || !lse) return OMR_ERR_ARG;
    a.q = q; a.k = k; a.v = v; a.o = (void*)o; a.lse = (float*)lse; a.dout = dout; a.delta = delta_ws; a.dq = dq; a.dk = dk; a.dv = dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.bsq = bsq; a.bsk = bsk; a.bsv = bsv; a.bso = bso; a.bsdo = bsdo; a.bsdq = bsdq; a.bsdk = bsdk; a.bsdv = bsdv;
    a.nsplit = 1; a.split_len = 0; a.part = nullptr;
    if (ws && T > 32) {
        int nsplit, len;
        choose_split(B, H, T, S, causal, &nsplit, &len);
        if (nsplit > 1) {
            if (ws_floats < (long)nsplit * B * T * H * head_dim) return OMR_ERR_ARG;
            a.nsplit = nsplit; a.split_len = len; a.part = ws;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMR_BF16) return head_dim == 64 ? run_bwd<bf16, 64>(a, s) : run_bwd<bf16, 32>(a, s);
    if (dtype == OMR_F32) return head_dim == 64 ? run_bwd<float, 64>(a, s) : run_bwd<float, 32>(a, s);
    return OMR_ERR_UNSUPPORTED;
}
