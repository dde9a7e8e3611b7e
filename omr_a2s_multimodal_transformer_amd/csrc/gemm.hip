// Generic MFMA GEMM for gfx950:  C[M,N] (+)= act( opA(A) . opB(B)^T + bias[N] )
//
// Canonical ("NT") operand form: A is [M][K] and B is [N][K], both k-contiguous.  transA / transB mean the operand is
// stored reduction-major instead ([K][M] / [K][N]).
//   linear forward      Y  = X  . W^T + b      (NT)            torch nn/modules/transformer.py:1158-1199
//   linear input grad   dX = dY . W            (transB)
//   linear weight grad  dW = dY^T . X          (transA+transB, split-K, fp32 atomic accumulate; db = column sums of dY fused)
//   1x1 "point_conv" of DepthSepConv2D (encoder.py:65-70) on NHWC activations = the same GEMMs
//   Conv1d(k=1) head (decoder.py:98-102)       = the same GEMM with N = vocabulary
// Block tile 128x128, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 blocks; BK = 4 k-steps (64 bf16 / 32 fp32).
// Operand staging: global -> registers (prefetch of tile t+1 issued before the MFMAs of tile t) -> LDS.
//   k-contiguous operands : [rows][BK] with a 16-byte odd-multiple pitch, fragments = ds_read_b128
//   reduction-major bf16   : staged untransposed [BK][rows], fragments = ds_read_b64_tr_b16 (no transposing stores)
//   reduction-major fp32   : transposed while stored (parity path only)
// Epilogue (bf16 C): the MFMA is issued with swapped operands so a lane holds runs of 4 consecutive columns, the wave's
// 64x64 tile goes through LDS (8-byte stores) and leaves as 16-byte row-contiguous global stores.  The fp32 / atomic
// (dW) epilogue keeps columns on lanes: one wave instruction then covers 128 contiguous bytes per row, the shape global
// float atomics want.
#include <type_traits>

#include "omr_common.h"
#include "omr_hip.h"

#include "gemm_args.h"

namespace {

// Tile of a workgroup.  Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2), so the linear
// tile order (split-major, then M tile, N tile fastest) is cut into chunks of `chunk` consecutive tiles and chunk c goes
// to XCD label c % 8: the N tiles that re-read one A panel -- or, with split-K, all tiles that re-read one K slice of both
// operands -- hit the same L2 instead of pulling the panel over the fabric once per XCD.  Pure speed: any placement is
// correct.  The grid is padded to whole rounds of 8 chunks; padded workgroups exit.
struct TileId { int m, n, split; bool valid; };
__device__ __forceinline__ TileId tile_of_block(const GemmArgs& g, int L) {      // L = block index inside this problem's grid
    const int slot = L >> 3;
    const long J = ((long)(slot / g.chunk) * 8 + (L & 7)) * g.chunk + slot % g.chunk;
    TileId t;
    t.valid = J < g.total;
    t.n = (int)(J % g.nt);
    t.m = (int)((J / g.nt) % g.mt);
    t.split = (int)(J / ((long)g.nt * g.mt));
    return t;
}

constexpr int BM = 128, BN = 128;

}  // namespace

// ---- fp8 (OCP e4m3) operands: 16 values per 16-byte fragment, two v_mfma_f32_32x32x16_fp8_fp8 per fragment pair -------------
typedef unsigned char fp8;
typedef __attribute__((ext_vector_type(16))) unsigned char u8x16;
template <> struct Frag<fp8> {
    typedef u8x16 type;
    static constexpr int N = 16;
};
template <> __device__ __forceinline__ fp8 from_f32<fp8>(float x) { return (fp8)(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xff); }
__device__ __forceinline__ float to_f32(fp8 x) { return __builtin_amdgcn_cvt_f32_fp8((int)x, 0); }
__device__ __forceinline__ void mma32(f32x16& acc, u8x16 a, u8x16 b) {
    // lane half h holds k = 16h .. 16h+15 of the 32-wide k-step; instruction i takes bytes [8i, 8i+8) of both operands: A and B
    // use the same (arbitrary) k order, so the sum is the dot product over all 32 k
    long a2[2], b2[2];
    __builtin_memcpy(a2, &a, 16);
    __builtin_memcpy(b2, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2[0], b2[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2[1], b2[1], acc, 0, 0, 0);
}

int omr_gemm_panel_bf16(const GemmArgs& g, hipStream_t s);      // gemm_panel.hip
int omr_gemm_tall_bf16(const GemmArgs& g, hipStream_t s);       // gemm_tall.hip

namespace {

// physical - logical row offset of the tile that starts at logical row i (tiles never straddle a group: host-checked)
__device__ __forceinline__ long grp_delta(const GemmArgs& g, int operand, int i) {
    return g.grp_operand == operand ? (long)(i / g.grp) * (g.grp_stride - g.grp) + g.grp_base : 0;
}

template <typename T> struct GemmCfg {
    static constexpr int VEC = Frag<T>::N;
    static constexpr int BK = 4 * KStep<T>::value;           // 64 (bf16) / 32 (fp32): 128 B per row
    static constexpr int PITCH = BK + VEC;                   // 144 B pitch: conflict-free b128 reads
    static constexpr int CHUNKS_PER_ROW = BK / VEC;          // 8
    static constexpr int NCH = BM * BK / VEC / 256;          // 16-byte chunks per thread per operand tile (4)
    // reduction-major bf16 operands stay [k][128 rows]; pitch = 16 dwords mod 64 so the 4 k-rows of a transposing read
    // tile all 64 banks
    static constexpr int TPITCH = 128 + 32;
    static constexpr int OP_ELEMS = (BM * PITCH > BK * TPITCH) ? BM * PITCH : BK * TPITCH;
    static constexpr int CPITCH = 64 + 8;                    // per-wave 64x64 output staging tile (bf16 C)
    static constexpr int LDS_ELEMS = (2 * OP_ELEMS > 4 * 64 * CPITCH) ? 2 * OP_ELEMS : 4 * 64 * CPITCH;
};
template <typename T, bool TR> struct UseTrRead { static constexpr bool value = TR && std::is_same<T, bf16>::value; };

template <typename T> __device__ __forceinline__ typename Frag<T>::type load_chunk_guard(
    const T* p, int valid)  // valid = number of leading elements that are in range (<= 0: none)
{
    typedef typename Frag<T>::type F;
    if (valid <= 0) return frag_zero<T>();
    F v = *reinterpret_cast<const F*>(p);
    if (valid < Frag<T>::N) {
#pragma unroll
        for (int i = 0; i < Frag<T>::N; ++i)
            if (i >= valid) v[i] = from_f32<T>(0.f);
    }
    return v;
}

// Stage one operand tile (128 rows x BK) into LDS.  TR=false: src is [rows][K] k-contiguous; TR=true: src is [K][rows].
template <typename T, bool TR> struct Stager {
    typedef GemmCfg<T> Cfg;
    typedef typename Frag<T>::type F;
    F r[Cfg::NCH];
    __device__ __forceinline__ void load(const T* src, long ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < Cfg::NCH; ++i) {
            const int c = tid + i * 256;
            if (!TR) {
                const int row = c / Cfg::CHUNKS_PER_ROW, kc = (c % Cfg::CHUNKS_PER_ROW) * Cfg::VEC;
                const int gr = row0 + row, gk = k0 + kc;
                r[i] = (gr < nrows) ? load_chunk_guard<T>(src + (long)gr * ld + gk, kend - gk) : frag_zero<T>();
            } else {
                constexpr int CPR = 128 / Cfg::VEC;  // chunks along the 128 tile rows, per k
                const int k = c / CPR, rc = (c % CPR) * Cfg::VEC;
                const int gk = k0 + k, gr = row0 + rc;
                r[i] = (gk < kend) ? load_chunk_guard<T>(src + (long)gk * ld + gr, nrows - gr) : frag_zero<T>();
            }
        }
    }
    __device__ __forceinline__ void store(T* lds, int tid) {
#pragma unroll
        for (int i = 0; i < Cfg::NCH; ++i) {
            const int c = tid + i * 256;
            if (!TR) {
                const int row = c / Cfg::CHUNKS_PER_ROW, kc = (c % Cfg::CHUNKS_PER_ROW) * Cfg::VEC;
                *reinterpret_cast<F*>(lds + row * Cfg::PITCH + kc) = r[i];
            } else {
                constexpr int CPR = 128 / Cfg::VEC;
                const int k = c / CPR, rc = (c % CPR) * Cfg::VEC;
                if constexpr (UseTrRead<T, TR>::value) {
                    *reinterpret_cast<F*>(lds + k * Cfg::TPITCH + rc) = r[i];     // untransposed; fragments come from tr reads
                } else {
#pragma unroll
                    for (int e = 0; e < Cfg::VEC; ++e) lds[(rc + e) * Cfg::PITCH + k] = r[i][e];
                }
            }
        }
    }
    // per-thread column sums of the staged chunks (TR only): every chunk of this thread covers the same VEC tile rows
    __device__ __forceinline__ void add_colsum(float (&cs)[Cfg::VEC]) {
#pragma unroll
        for (int i = 0; i < Cfg::NCH; ++i)
#pragma unroll
            for (int e = 0; e < Cfg::VEC; ++e) cs[e] += to_f32(r[i][e]);
    }
};

// Operand fragment for the 32 output rows starting at `row0` and k-step `ks` of the current tile.
template <typename T, bool TR>
__device__ __forceinline__ typename Frag<T>::type load_frag(const T* lds, int row0, int ks, int lane) {
    typedef GemmCfg<T> Cfg;
    typedef typename Frag<T>::type F;
    if constexpr (UseTrRead<T, TR>::value) {
        typedef __attribute__((address_space(3))) bf16x4 LdsV4;
        const int q = (lane & 15) >> 2, col = row0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4, k = ks + 8 * (lane >> 5) + q;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(lds + k * Cfg::TPITCH + col));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(lds + (k + 4) * Cfg::TPITCH + col));
        const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    } else {
        return *reinterpret_cast<const F*>(&lds[(row0 + (lane & 31)) * Cfg::PITCH + ks + (lane >> 5) * Cfg::VEC]);
    }
}

// FULLK (forward-type GEMMs with K <= 4 BK, i.e. d_model = 256 in bf16): all k-slabs of both operands are requested up front
// (one global round trip instead of four dependent ones -- these GEMMs are latency chains, not throughput problems), at
// the price of 96 more staging registers.
template <typename T, typename TC, bool TA, bool TB, bool FULLK>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int block_index) {
    typedef GemmCfg<T> Cfg;
    typedef typename Frag<T>::type F;
    constexpr bool SWAP = std::is_same<TC, bf16>::value && std::is_same<T, bf16>::value;     // bf16 in/out: transposed accumulators + LDS-staged stores
    __shared__ __attribute__((aligned(16))) T smem[Cfg::LDS_ELEMS];
    T* As = smem;
    T* Bs = smem + Cfg::OP_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const TileId tile = tile_of_block(g, block_index);
    if (!tile.valid) return;
    const int m0 = tile.m * BM, n0 = tile.n * BN;
    const int kbeg = tile.split * g.ksplit_len;
    const int kend = min(g.K, kbeg + g.ksplit_len);
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B + grp_delta(g, 1, n0) * g.ldb;

    // bias of the tile's 128 columns -> LDS (the bf16 epilogue reads it as float4 runs; 32 gathered global loads per lane otherwise)
    __shared__ __attribute__((aligned(16))) float sbias[BN];
    if constexpr (SWAP) {
        if (tid < BN) {
            const float* bp = g.bias ? g.bias + grp_delta(g, 1, n0) : nullptr;
            sbias[tid] = (bp != nullptr && tile.split == 0 && n0 + tid < g.N) ? bp[n0 + tid] : 0.f;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stager<T, TA> sa;
    Stager<T, TB> sb;
    float cs[Cfg::VEC];
#pragma unroll
    for (int e = 0; e < Cfg::VEC; ++e) cs[e] = 0.f;
    const bool do_cs = TA && g.colsum_a != nullptr && tile.n == 0;
    auto mma_slab = [&]() {
#pragma unroll
        for (int ks = 0; ks < Cfg::BK; ks += KStep<T>::value) {
            F a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = load_frag<T, TA>(As, wm * 64 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = load_frag<T, TB>(Bs, wn * 64 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (SWAP) mma32(acc[i][j], b[j], a[i]);     // D[n][m]: lane = row m, registers = columns n
                    else mma32(acc[i][j], a[i], b[j]);                     // D[m][n]: lane = column n
                }
        }
    };
    if constexpr (FULLK) {
        Stager<T, TA> fa[4];
        Stager<T, TB> fb[4];
        const int nk = (kend - kbeg + Cfg::BK - 1) / Cfg::BK;        // host guarantees nk <= 4, one split
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (sl < nk) {
                fa[sl].load(A, g.lda, m0, g.M, kbeg + sl * Cfg::BK, kend, tid);
                fb[sl].load(B + grp_delta(g, 2, kbeg + sl * Cfg::BK) * g.ldb, g.ldb, n0, g.N, kbeg + sl * Cfg::BK, kend, tid);
            }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (sl < nk) {
                if (sl) __syncthreads();
                fa[sl].store(As, tid);
                fb[sl].store(Bs, tid);
                __syncthreads();
                mma_slab();
            }
        __syncthreads();
    } else {
    sa.load(A, g.lda, m0, g.M, kbeg, kend, tid);
    sb.load(B + grp_delta(g, 2, kbeg) * g.ldb, g.ldb, n0, g.N, kbeg, kend, tid);
    if (do_cs) sa.add_colsum(cs);
    sa.store(As, tid);
    sb.store(Bs, tid);
    __syncthreads();

    for (int k0 = kbeg; k0 < kend; k0 += Cfg::BK) {
        const bool more = (k0 + Cfg::BK) < kend;
        if (more) {
            sa.load(A, g.lda, m0, g.M, k0 + Cfg::BK, kend, tid);
            sb.load(B + grp_delta(g, 2, k0 + Cfg::BK) * g.ldb, g.ldb, n0, g.N, k0 + Cfg::BK, kend, tid);
        }
        mma_slab();
        __syncthreads();
        if (more) {
            if (do_cs) sa.add_colsum(cs);   // here the prefetched registers are needed anyway (no extra wait)
            sa.store(As, tid);
            sb.store(Bs, tid);
            __syncthreads();
        }
    }
    }

    if (do_cs) {   // combine the 256 per-thread partials in LDS (operand tiles are dead), then ONE global atomic per row per block
        constexpr int CPR = 128 / Cfg::VEC;
        float* red = reinterpret_cast<float*>(smem);
        if (tid < 128) red[tid] = 0.f;
        __syncthreads();
        const int ml = (tid % CPR) * Cfg::VEC;
#pragma unroll
        for (int e = 0; e < Cfg::VEC; ++e) atomicAdd(&red[ml + e], cs[e]);
        __syncthreads();
        if (tid < 128 && m0 + tid < g.M) atomicAdd(&g.colsum_a[m0 + tid + grp_delta(g, 3, m0)], red[tid]);
        __syncthreads();
    }

    TC* C = (TC*)g.C + grp_delta(g, 3, m0) * g.ldc;
    const float* biasp = g.bias ? g.bias + grp_delta(g, 1, n0) : nullptr;
    if constexpr (SWAP) {
        // ---- bf16 C: lane = row (lane & 31) of M-block i; register r = column (r&3) + 8(r>>2) + 4(lane>>5) of N-block j
        typedef __attribute__((ext_vector_type(4))) bf16 B4;
        T* Cs = smem + wave * (64 * Cfg::CPITCH);          // all waves passed the last barrier: operand tiles are dead
        const int hsel = 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float bv[16];          // bias of this lane's 16 columns: four 16-byte reads of the block's LDS copy (staged at kernel start)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sbias[wn * 64 + j * 32 + 8 * q + hsel]);
                bv[4 * q] = b4[0]; bv[4 * q + 1] = b4[1]; bv[4 * q + 2] = b4[2]; bv[4 * q + 3] = b4[3];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                T* crow = Cs + (i * 32 + (lane & 31)) * Cfg::CPITCH + j * 32 + hsel;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    B4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * q + e] + bv[4 * q + e];
                        if (g.relu) v = fmaxf(v, 0.f);
                        o[e] = (bf16)v;
                    }
                    *reinterpret_cast<B4*>(crow + 8 * q) = o;
                }
            }
        }
        // same-wave LDS writes are ordered before the reads below (in-order LDS queue); no block barrier needed
        const bool vec_ok = (g.ldc % 8) == 0 && (((uintptr_t)C) & 15) == 0;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = lane + it * 64, row = c >> 3, kc = (c & 7) * 8;
            const int gr = m0 + wm * 64 + row, gc = n0 + wn * 64 + kc;
            if (gr >= g.M || gc >= g.N) continue;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(Cs + row * Cfg::CPITCH + kc);
            if (g.drop_thresh) {      // nn.Dropout after the activation: omr_dropout's mask over the flat [M][ldc] index
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = drop_keep(g.drop_seed, (uint64_t)((long)gr * g.ldc + gc + e), g.drop_thresh) ? (bf16)((float)v[e] * g.drop_scale) : (bf16)0.f;
            }
            TC* dst = C + (long)gr * g.ldc + gc;
            if (vec_ok && gc + 8 <= g.N && !g.accum) {
                *reinterpret_cast<bf16x8*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (gc + e < g.N) dst[e] = g.accum ? (bf16)((float)dst[e] + (float)v[e]) : v[e];
            }
        }
    } else {
        const int col_l = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + col_l;
            if (col >= g.N) continue;
            const float bv = (biasp != nullptr && tile.split == 0) ? biasp[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                    if (row >= g.M) continue;
                    float v = acc[i][j][r];
                    if constexpr (std::is_same<T, fp8>::value) v *= g.scale_a[row] * g.scale_b[col];      // de-quantise: per-row scales of both operands
                    v += bv;
                    if (g.relu) v = fmaxf(v, 0.f);
                    if constexpr (!TA) {      // forward-type GEMMs only: the weight-gradient instantiations keep their lean atomic loop
                        if (g.drop_thresh) v = drop_keep(g.drop_seed, (uint64_t)((long)row * g.ldc + col), g.drop_thresh) ? to_f32(from_f32<TC>(v)) * g.drop_scale : 0.f;
                    }
                    TC* dst = C + (long)row * g.ldc + col;
                    if (g.atomic) {
                        if constexpr (sizeof(TC) == 4) atomicAdd((float*)dst, v);
                    } else if (g.accum) {
                        *dst = from_f32<TC>(to_f32(*dst) + v);
                    } else {
                        *dst = from_f32<TC>(v);
                    }
                }
            }
        }
    }
}

template <typename T, typename TC, bool TA, bool TB, bool FULLK>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    gemm_body<T, TC, TA, TB, FULLK>(g, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Grouped weight-gradient GEMM: dW_p[n_out][n_in] += dY_p^T . X_p (+ db_p += column sums of dY_p) for up to DW_MAX linear
// layers in ONE launch.  A decoder layer's dW products are 256 x 256 outputs reduced over 16 384 rows: launched one by one
// they are a handful of workgroups each -- latency chains that leave most of the chip idle (50 launches, 4.2 ms per step at
// C2).  Collected over the backward pass and launched together they are ~1 600 workgroups of 2 048-row reductions.  The
// problem table travels in the kernel arguments (no descriptor buffer to keep alive); a workgroup finds its problem by its
// block index (uniform scalar scan) and runs the ordinary transA/transB split-K body.
struct DwProb {
    const void* A; const void* B; float* C; float* colsum;
    int M, N, K, ksplit_len; int lda, ldb, ldc, block0; int nt, mt, chunk, total; int grp, grp_stride, grp_base, pad;
};
constexpr int DW_MAX = 24;
struct DwGroup { int nprob, pad; DwProb p[DW_MAX]; };

template <typename T>
__global__ __launch_bounds__(256) void gemm_dw_grouped_kernel(DwGroup grp) {
    int pi = 0;
    for (int i = 1; i < grp.nprob; ++i) pi = ((int)blockIdx.x >= grp.p[i].block0) ? i : pi;
    const DwProb& q = grp.p[pi];
    GemmArgs g;
    g.A = q.A; g.B = q.B; g.C = q.C; g.bias = nullptr; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc;
    g.relu = 0; g.accum = 1; g.atomic = 1; g.ksplit_len = q.ksplit_len; g.colsum_a = q.colsum;
    g.mt = q.mt; g.nt = q.nt; g.chunk = q.chunk; g.total = q.total;
    g.drop_thresh = 0; g.drop_scale = 1.f; g.drop_seed = 0;
    g.grp = q.grp; g.grp_stride = q.grp_stride; g.grp_base = q.grp_base; g.grp_operand = q.grp ? 3 : 0;
    g.scale_a = nullptr; g.scale_b = nullptr;
    gemm_body<T, float, true, true, false>(g, (int)blockIdx.x - q.block0);
}

template <typename T, typename TC> int launch(GemmArgs g, int ta, int tb, int splits, hipStream_t s) {
    g.nt = cdiv(g.N, BN); g.mt = cdiv(g.M, BM);
    // FULLK keeps all four k-slabs of both operands in staging registers: fine for 2- and 4-byte elements, 42-75 spilled
    // registers with fp8's 16-element chunks -- the fp8 GEMMs take the pipelined loop.
    constexpr bool CAN_FULLK = !std::is_same<T, fp8>::value;
    const bool fullk = CAN_FULLK && !ta && splits == 1 && g.K <= 4 * GemmCfg<T>::BK;
    g.total = g.nt * g.mt * splits;
    g.chunk = splits > 1 ? g.nt * g.mt : g.nt;
    const int nchunks = g.total / g.chunk;
    dim3 grid((unsigned)(cdiv(nchunks, 8) * 8 * g.chunk)), block(256);
    bool done = false;
    if constexpr (CAN_FULLK) {
        if (!ta && !tb && fullk) { hipLaunchKernelGGL((gemm_kernel<T, TC, false, false, true>), grid, block, 0, s, g); done = true; }
        else if (!ta && tb && fullk) { hipLaunchKernelGGL((gemm_kernel<T, TC, false, true, true>), grid, block, 0, s, g); done = true; }
    }
    if (done) {}
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, TC, false, false, false>), grid, block, 0, s, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<T, TC, false, true, false>), grid, block, 0, s, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, TC, true, false, false>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<T, TC, true, true, false>), grid, block, 0, s, g);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

}  // namespace

extern "C" int omr_gemm(int dtype, int c_dtype, int transA, int transB, int M, int N, int K, const void* A, long lda,
                        const void* B, long ldb, void* C, long ldc, const float* bias, int relu, int accumulate,
                        int split_k, float* colsum_a, float drop_p, unsigned long long drop_seed, int row_group, int row_group_stride,
                        int row_group_base, int row_group_operand, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return OMR_ERR_ARG;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (lda % vec || ldb % vec) return OMR_ERR_ARG;                       // 16-byte aligned rows
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return OMR_ERR_ARG;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && c_dtype != OMR_F32) return OMR_ERR_ARG;            // split-K accumulates with fp32 atomics
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.relu = relu; g.accum = accumulate; g.atomic = split_k > 1; g.colsum_a = colsum_a;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (split_k > 1 || accumulate || transA))) return OMR_ERR_ARG;      // dropout needs the final value
    g.grp = g.grp_stride = g.grp_base = g.grp_operand = 0;
    g.scale_a = g.scale_b = nullptr;
    if (row_group_operand) {
        if (row_group_operand < 1 || row_group_operand > 3 || row_group <= 0 || row_group % 128 || row_group_stride < row_group || row_group_base < 0) return OMR_ERR_ARG;
        if ((row_group_operand == 2 && !transB) || (row_group_operand == 1 && transB)) return OMR_ERR_ARG;
        g.grp = row_group; g.grp_stride = row_group_stride; g.grp_base = row_group_base; g.grp_operand = row_group_operand;
    }
    g.drop_thresh = OMR_DROP_THRESH16(drop_p); g.drop_scale = 1.f / (1.f - drop_p); g.drop_seed = drop_seed;
    if (colsum_a && !transA) return OMR_ERR_ARG;
    const int bk = dtype == OMR_BF16 ? 64 : 32;
    int len = cdiv(cdiv(K, split_k), bk) * bk;
    g.ksplit_len = len;
    int splits = cdiv(K, len);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMR_BF16 && c_dtype == OMR_BF16 && !transA && !transB && !relu && !accumulate && splits == 1 && !colsum_a && drop_p == 0.f &&
        !(lda % 8) && !(ldb % 8)) {
        // tall-and-wide output with a short reduction (the all-layer K|V projection of the memory): panel kernel, gemm_panel.hip
        const int rc = omr_gemm_panel_bf16(g, s);
        if (rc != OMR_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == OMR_BF16 && c_dtype == OMR_BF16 && !transA && transB && splits == 1) {
        // tall output of 256 columns with a long reduction (the data gradient of the all-layer K|V projection): DMA-fed kernel, gemm_tall.hip
        const int rc = omr_gemm_tall_bf16(g, s);
        if (rc != OMR_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == OMR_BF16) {
        if (c_dtype == OMR_BF16) return launch<bf16, bf16>(g, transA, transB, splits, s);
        return launch<bf16, float>(g, transA, transB, splits, s);
    } else if (dtype == OMR_F32) {
        if (c_dtype != OMR_F32) return OMR_ERR_UNSUPPORTED;
        return launch<float, float>(g, transA, transB, splits, s);
    }
    return OMR_ERR_UNSUPPORTED;
}

/* Weight gradients of several linear layers in one launch (see gemm_dw_grouped_kernel).  rows = reduction length. */
extern "C" int omr_linear_wgrad_grouped(int dtype, int nprob, const omr_dw_problem* probs, void* stream) {
    if (nprob <= 0 || !probs) return OMR_ERR_ARG;
    if (dtype != OMR_BF16 && dtype != OMR_F32) return OMR_ERR_UNSUPPORTED;
    const int vec = dtype == OMR_BF16 ? 8 : 4, bk = dtype == OMR_BF16 ? 64 : 32;
    // Split factor: all problems of the call together should put ~8 workgroups on every CU, no workgroup reducing fewer than
    // 512 rows (below that the prologue / atomic epilogue dominate) -- chosen per problem from its own tile count.
    long tiles_total = 0;
    for (int i = 0; i < nprob; ++i) tiles_total += (long)cdiv(probs[i].n_out, BM) * cdiv(probs[i].n_in, BN);
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < nprob; base += DW_MAX) {
        DwGroup grp;
        grp.nprob = nprob - base < DW_MAX ? nprob - base : DW_MAX; grp.pad = 0;
        int block0 = 0;
        for (int i = 0; i < grp.nprob; ++i) {
            const omr_dw_problem& q = probs[base + i];
            if (!q.dy || !q.x || !q.dw || q.rows <= 0 || q.n_out <= 0 || q.n_in <= 0) return OMR_ERR_ARG;
            if (q.ld_dy % vec || q.ld_x % vec || ((uintptr_t)q.dy & 15) || ((uintptr_t)q.x & 15)) return OMR_ERR_ARG;
            if (q.row_group && (q.row_group % 128 || q.row_group_stride < q.row_group || q.row_group_base < 0)) return OMR_ERR_ARG;
            DwProb& d = grp.p[i];
            d.A = q.dy; d.B = q.x; d.C = q.dw; d.colsum = q.db;
            d.M = q.n_out; d.N = q.n_in; d.K = q.rows; d.lda = (int)q.ld_dy; d.ldb = (int)q.ld_x; d.ldc = (int)q.ld_dw;
            d.mt = cdiv(d.M, BM); d.nt = cdiv(d.N, BN);
            long want = (2048L + tiles_total - 1) / tiles_total;            // splits so that the call totals ~2048 workgroups
            long maxs = (q.rows + 511) / 512;
            if (want > maxs) want = maxs;
            if (want < 1) want = 1;
            int len = cdiv(cdiv(d.K, (int)want), bk) * bk;
            d.ksplit_len = len;
            const int splits = cdiv(d.K, len);
            d.total = d.nt * d.mt * splits;
            d.chunk = splits > 1 ? d.nt * d.mt : d.nt;
            const int nchunks = d.total / d.chunk;
            d.block0 = block0;
            block0 += cdiv(nchunks, 8) * 8 * d.chunk;
            d.grp = q.row_group; d.grp_stride = q.row_group_stride; d.grp_base = q.row_group_base; d.pad = 0;
        }
        if (dtype == OMR_BF16) hipLaunchKernelGGL((gemm_dw_grouped_kernel<bf16>), dim3((unsigned)block0), dim3(256), 0, s, grp);
        else hipLaunchKernelGGL((gemm_dw_grouped_kernel<float>), dim3((unsigned)block0), dim3(256), 0, s, grp);
        OMR_CHECK_LAUNCH();
    }
    return OMR_OK;
}

// ------------------------------------------------------------------------------------------------
// fp8 path (BASELINE config "fp8 MFMA weights"; an extension: the reference has no fp8).  Both operands are OCP e4m3 with one
// fp32 scale per ROW (weights: per output feature, quantised once; activations: per token, quantised on the fly by
// omr_quantize_rows_fp8), the products run on the fp8 MFMA, accumulation and the de-quantising epilogue are fp32.
namespace {
// one wave per row: absmax -> scale = absmax / 448 (e4m3 max) -> q = x / scale rounded to fp8 (RNE)
template <typename T>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const T* __restrict__ x, long ldx, fp8* __restrict__ q, long ldq, float* __restrict__ scale,
                                                                int M, int K) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * ldx;
    float mx = 0.f;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(to_f32(xr[k])));
    mx = wave_max(mx);
    const float sc = mx > 0.f ? mx / 448.f : 1.f;
    for (int k = lane; k < K; k += 64) q[row * ldq + k] = from_f32<fp8>(to_f32(xr[k]) / sc);      // true divisions: the codes equal torch's (x / scale).to(float8_e4m3fn)
    if (lane == 0) scale[row] = sc;
}
}  // namespace

extern "C" int omr_quantize_rows_fp8(int dtype, const void* x, long ldx, unsigned char* q, long ldq, float* scale, int M, int K, void* stream) {
    if (M <= 0 || K <= 0 || !x || !q || !scale || ldx < K || ldq < K) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMR_BF16) hipLaunchKernelGGL((quantize_rows_fp8_kernel<bf16>), cdiv(M, 4), 256, 0, s, (const bf16*)x, ldx, q, ldq, scale, M, K);
    else if (dtype == OMR_F32) hipLaunchKernelGGL((quantize_rows_fp8_kernel<float>), cdiv(M, 4), 256, 0, s, (const float*)x, ldx, q, ldq, scale, M, K);
    else return OMR_ERR_UNSUPPORTED;
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_gemm_fp8(int c_dtype, int M, int N, int K, const unsigned char* A8, long lda, const float* scale_a, const unsigned char* W8, long ldb,
                            const float* scale_w, void* C, long ldc, const float* bias, int relu, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A8 || !W8 || !C || !scale_a || !scale_w) return OMR_ERR_ARG;
    if (lda % 16 || ldb % 16 || ((uintptr_t)A8 & 15) || ((uintptr_t)W8 & 15)) return OMR_ERR_ARG;      // 16-byte aligned rows
    GemmArgs g;
    g.A = A8; g.B = W8; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.relu = relu; g.accum = 0; g.atomic = 0; g.colsum_a = nullptr; g.ksplit_len = cdiv(K, 128) * 128;
    g.drop_thresh = 0; g.drop_scale = 1.f; g.drop_seed = 0;
    g.grp = g.grp_stride = g.grp_base = g.grp_operand = 0;
    g.scale_a = scale_a; g.scale_b = scale_w;
    hipStream_t s = (hipStream_t)stream;
    if (c_dtype == OMR_BF16) return launch<fp8, bf16>(g, 0, 0, 1, s);
    if (c_dtype == OMR_F32) return launch<fp8, float>(g, 0, 0, 1, s);
    return OMR_ERR_UNSUPPORTED;
}
