// Tall GEMM with a narrow output and a LONG reduction:  C[M][256] = A[M][K] . B[K][256], bf16, B reduction-major.
// The case: the data gradient of the key|value projection of the encoder memory for ALL decoder layers
// (functional.FusedCrossKVFn.backward; nn.MultiheadAttention in_proj rows [d, 3d) of every layer, decoder.py:86-95):
// M = B*S = 131 072 rows, K = L*2d = 3 072, N = d = 256 at the benchmark -- 206 GFLOP against 805 MB of gradient to READ, so the
// A stream (160 us at 5 TB/s) is the real work.  The 128 x 128 tile kernel of gemm.hip stages every A tile twice (once per N
// tile) through registers at one or two workgroups per CU and reaches 2.2 TB/s on it (368 us).
//
// Here one 16-wave workgroup per CU owns 256 rows for the whole width:
//   * both operands travel global -> LDS asynchronously (global_load_lds_dwordx4, dma_common.h) in k-tiles of 64, two stages
//     of 32 KB (A: 256 rows x 128 B) + 32 KB (B: 64 reduction rows x 512 B): the next 64 KB are in flight behind the current
//     tile's MFMAs, no staging registers, and every wave issues exactly four copies per k-tile, so "my pieces have landed" is an
//     exact s_waitcnt and there is ONE workgroup barrier per k-tile;
//   * a DMA instruction fills 1 KB of contiguous LDS, so rows cannot be padded: bank conflicts are removed by permuting the
//     SOURCE chunk instead -- A: 16-byte chunk c of row m lands at position c ^ ((m >> 1) & 7) (the 16 rows a ds_read_b128
//     group touches cover all 64 banks); B: chunk c of reduction row k lands at c ^ ((k & 3) << 2) (the 4 rows x 64 B half a
//     wave touches in one ds_read_b64_tr_b16 cover all 64 banks).  k & 3 and (m >> 1) & 7 are lane constants, so the addresses
//     of the MFMA loop are per-lane bases + compile-time offsets;
//   * waves 4 (M) x 4 (N), 64 x 64 each; the product is taken as C^T so a lane owns one output row and runs of four
//     consecutive columns: the 256 x 256 tile is packed to bf16 into the (dead) stage memory and leaves as whole 512-byte rows.
// Row groups on the reduction side (GemmArgs grp_operand 2: the K|V rows of the layers' packed in_proj matrices) are resolved
// per k-tile (a tile never straddles a group: grp % 64 == 0, host-checked).
#include <atomic>
#include "omr_common.h"
#include "omr_hip.h"
#include "gemm_args.h"
#include "dma_common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 LdsV4;
constexpr int TM = 256, TN = 256;
#ifndef OMR_TALL_BK
#define OMR_TALL_BK 64
#endif
constexpr int BK = OMR_TALL_BK, NST = 128 / BK;       // k-tile and ring depth: 128 KB of stages either way.  64 x 2 stages: 221 us on the
                                                      // K|V data gradient; 32 x 4 stages (64-byte row pieces, twice the barriers): 234 us
constexpr int KS = BK / 16, ROWB = BK * 2, CPR = BK / 8;                       // k-steps per tile; bytes / 16-byte chunks of an A row
constexpr int A_ROWS_PER_COPY = 1024 / ROWB, A_COPIES = TM / A_ROWS_PER_COPY / 16, B_COPIES = BK / 2 / 16;      // per wave and k-tile
constexpr int A_BYTES = TM * BK * 2, B_BYTES = BK * TN * 2, STAGE = A_BYTES + B_BYTES;
__device__ __forceinline__ int a_swz(int m) { return BK == 64 ? (m >> 1) & 7 : (m >> 2) & 3; }
constexpr int CP = TN * 2 + 16;                       // byte pitch of the output image (rows 16 B apart in banks: 2-way on the 8-byte writes)
constexpr int SMEM = TM * CP > NST * STAGE ? TM * CP : NST * STAGE;

__global__ __launch_bounds__(1024) void gemm_tall_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const bf16* A = (const bf16*)g.A;
    const bf16* B = (const bf16*)g.B;
    bf16* C = (bf16*)g.C;
    const unsigned lds0 = lds_address(smem);
    const int nk = g.K / BK;

    // MFMA-loop addresses (bytes from the stage base).  A fragment of row block i, k-step ks: chunk 2 ks + hh of row mrow;
    // B fragment of column block j: two transposing reads at reduction rows 16 ks + 8 hh + q (+ 4)
    int a_off[KS];
    {
        const int mrow = wm * 64 + (lane & 31), sw = a_swz(mrow);             // + 32 i leaves the swizzle bits alone
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_off[ks] = mrow * ROWB + (((2 * ks + hh) ^ sw) << 4);
    }
    const int q = (lane & 15) >> 2, bcol = wn * 64 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    int b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = bcol + j * 32;
        b_off[j] = A_BYTES + (8 * hh + q) * 512 + ((((col >> 3) ^ (q << 2))) << 4) + (col & 7) * 2;
    }

    for (int tile = blockIdx.x; tile < g.mt; tile += gridDim.x) {
        const int m0 = tile * TM;
        // DMA sources of this wave's copies per k-tile (1 KB each): A_COPIES pieces of A (8 rows each at BK = 64), B_COPIES pieces of B (2 reduction rows each)
        const bf16* a_src[A_COPIES];
        long b_rel[B_COPIES];
#pragma unroll
        for (int u = 0; u < A_COPIES; ++u) {
            const int r = (A_COPIES * wave + u) * A_ROWS_PER_COPY + lane / CPR, c = (lane % CPR) ^ a_swz(r);
            a_src[u] = A + (long)min(m0 + r, g.M - 1) * g.lda + c * 8;          // rows beyond M: a valid row, computed, never stored
        }
#pragma unroll
        for (int u = 0; u < B_COPIES; ++u) {
            const int kr = (B_COPIES * wave + u) * 2 + (lane >> 5), cb = (lane & 31) ^ ((kr & 3) << 2);
            b_rel[u] = (long)kr * g.ldb + cb * 8;
        }
        auto issue = [&](int kt) {
            const unsigned st = lds0 + (kt % NST) * STAGE;
            const int k0 = kt * BK;
            const long krow = g.grp_operand == 2 ? (long)(k0 / g.grp) * g.grp_stride + g.grp_base + k0 % g.grp : k0;
            const bf16* bt = B + krow * g.ldb;
#pragma unroll
            for (int u = 0; u < A_COPIES; ++u) dma16(a_src[u] + k0, st + (A_COPIES * wave + u) * 1024);
#pragma unroll
            for (int u = 0; u < B_COPIES; ++u) dma16(bt + b_rel[u], st + A_BYTES + (B_COPIES * wave + u) * 1024);
        };
        constexpr int PER_TILE = A_COPIES + B_COPIES;

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
            if (p < nk) issue(p);
        for (int kt = 0; kt < nk; ++kt) {
            // my copies of k-tile kt have landed (the NST - 2 younger tiles may still fly; at the tail, where fewer were issued, wait for
            // all); behind the barrier: everyone's have, and everyone is done with kt - 1, whose stage the next issue overwrites
            if (kt + NST - 2 < nk) dma_wait_barrier<(NST - 2) * PER_TILE>();
            else dma_wait_barrier<0>();
            if (kt + NST - 1 < nk) issue(kt + NST - 1);
            const unsigned char* st = smem + (kt % NST) * STAGE;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[ks] + i * 32 * ROWB);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned char* p = st + b_off[j] + ks * 16 * 512;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)p);
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(p + 4 * 512));
                    const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bfr[j] = f;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma32(acc[i][j], bfr[j], af[i]);       // D[n][m]: lane = row m, registers = columns n
            }
        }
        lds_barrier();                             // every wave is done with the last stage: the output image takes its place
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned char* row = smem + (wm * 64 + i * 32 + (lane & 31)) * CP + (wn * 64 + j * 32 + 4 * hh) * 2;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * gq + e];
                    *reinterpret_cast<bf16x4*>(row + gq * 16) = o;
                }
            }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < TM * 32 / 1024; ++it) {
            const int c = tid + it * 1024, row = c >> 5, ch = c & 31;
            if (m0 + row < g.M) *reinterpret_cast<uint4*>(C + (long)(m0 + row) * g.ldc + ch * 8) = *reinterpret_cast<const uint4*>(smem + row * CP + ch * 16);
        }
        lds_barrier();                             // the image is read: the next tile's copies may land
    }
}

}  // namespace

// OMR_ERR_UNSUPPORTED = "not my shape": the caller falls through to the tile kernel.
int omr_gemm_tall_bf16(const GemmArgs& g0, hipStream_t s) {
    GemmArgs g = g0;
    if (g.N != TN || g.K % BK || g.K < BK || g.bias || g.relu || g.accum || g.atomic || g.colsum_a || g.drop_thresh) return OMR_ERR_UNSUPPORTED;
    if (g.lda % 8 || g.ldb % 8 || g.ldc % 8 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return OMR_ERR_UNSUPPORTED;
    if (g.grp_operand != 0 && (g.grp_operand != 2 || g.grp % BK)) return OMR_ERR_UNSUPPORTED;
    g.mt = (g.M + TM - 1) / TM;
    if (g.mt < 192) return OMR_ERR_UNSUPPORTED;          // fewer row tiles than CUs: the tile kernel spreads better
    static std::atomic<int> ready{0};
    if (!ready.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)gemm_tall_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return OMR_ERR_LAUNCH;
        ready.store(1, std::memory_order_release);
    }
    const int grid = g.mt < 256 ? g.mt : 256;
    hipLaunchKernelGGL(gemm_tall_kernel, dim3(grid), dim3(1024), SMEM, s, g);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
