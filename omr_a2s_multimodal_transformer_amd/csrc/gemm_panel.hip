// Panel GEMM for the tall-and-wide linears with a SHORT reduction:  C[M][N] = A[M][K] . W[N][K]^T + bias, bf16, K = 128 / 256.
// The case: the key|value projection of the encoder memory for ALL decoder layers (nn.MultiheadAttention in_proj rows [d, 3d) of
// every layer, decoder.py:86-95; functional.FusedCrossKVFn): M = B*S = 131 072 rows, N = L*2d = 3 072, K = d = 256 at the
// benchmark -- 206 GFLOP against 805 MB of output, so the output write (134 us at 6 TB/s) is the real work and the tile kernel
// of gemm.hip, which re-stages both operands for every 128 x 128 output tile, spends ~500 us on it (this kernel: ~345 us --
// 0.9 PFLOP/s in its MFMA + epilogue loop alone, two lock-stepped waves per SIMD; the rest is store / fetch latency that one
// workgroup per CU cannot hide).
//
// Here a 512-thread workgroup owns a PANEL of 256 rows for the whole width N:
//   * its A rows never touch LDS: each wave keeps the 16-byte MFMA fragments of its 64 rows x K in registers for the whole
//     sweep (2 row blocks x K/16 fragments = 128 registers at K = 256);
//   * W travels in 64-column tiles global -> registers -> LDS (double buffered; the loads of tile t+1 are issued before the
//     MFMAs of tile t and committed after them), each fragment read from LDS feeds two MFMAs (the wave's two row blocks);
//   * the product is taken as C^T (W tile on the MFMA row axis, activations on the column axis), so a lane owns one output row
//     and four consecutive columns per accumulator group: + bias, packed to bf16, written to an LDS image of the 256 x 64 tile
//     with conflict-free 8-byte stores, then streamed out as whole 128-byte rows (16 bytes per lane);
//   * one workgroup barrier per 64-column step (W tiles and the C image are both double buffered).
#include "omr_common.h"
#include "omr_hip.h"

#include "gemm_args.h"

namespace {

constexpr int PM = 256, PN = 64;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;

__device__ __forceinline__ long pgrp_delta(const GemmArgs& g, int n) {
    return g.grp_operand == 1 ? (long)(n / g.grp) * (g.grp_stride - g.grp) + g.grp_base : 0;
}

__device__ float g_zero_bias[4096];      // stands in for a missing bias: the per-tile bias load stays unconditional (see fetch)

template <int NKS>
__global__ __launch_bounds__(512, 2) void gemm_panel_kernel(GemmArgs g) {
    constexpr int K = NKS * 16, PB = K + 8, PC = PN + 8;        // LDS pitches (elements): 16-byte fragments at an odd multiple of 16 B
    constexpr int CPR = K / 8, NLD = PN * CPR / 512;            // 16-byte chunks per W row; chunks per thread and tile
    static_assert(PN * CPR % 512 == 0, "whole chunks per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Bs = reinterpret_cast<bf16*>(smem);                   // [2][PN][PB]
    bf16* Cs = Bs + 2 * PN * PB;                                // [2][PM][PC]
    float* bias_s = reinterpret_cast<float*>(Cs + 2 * PM * PC); // [2][PN]

    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * PM;
    const bf16* A = (const bf16*)g.A;
    const bf16* W = (const bf16*)g.B;
    bf16* C = (bf16*)g.C;

    // this wave's activation rows as MFMA column fragments, resident for the whole sweep
    bf16x8 af[2][NKS];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = min(m0 + wm * 64 + rb * 32 + (lane & 31), g.M - 1);          // rows beyond M: computed, never stored
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) af[rb][ks] = *reinterpret_cast<const bf16x8*>(A + (long)row * g.lda + ks * 16 + hh * 8);
    }
    const int ntiles = g.N / PN;
    bf16x8 wreg[NLD];
    float breg = 0.f;
    // Every thread loads a bias entry, unconditionally: a load under `if (tid < PN)` is an exec-masked branch whose register
    // the compiler then protects with a vmcnt wait that also covers the previous step's C stores -- one wave stalls a full
    // store round trip per step and the barrier spreads it to the workgroup.
    const float* biasp = g.bias ? g.bias : g_zero_bias;
    const long bias_mask = g.bias ? ~0L : 0L;                   // no bias: every tile reads entries [0, 64) of the zero array
    // per-thread element offsets inside a W tile / a C tile are constants of the sweep; a step only moves the (scalar) tile base
    unsigned woff[NLD], coff[PM * PN / 8 / 512];
#pragma unroll
    for (int i = 0; i < NLD; ++i) { const int c = tid + i * 512; woff[i] = (unsigned)((c / CPR) * (int)g.ldb + (c % CPR) * 8); }
#pragma unroll
    for (int i = 0; i < PM * PN / 8 / 512; ++i) { const int c = tid + i * 512; coff[i] = (unsigned)((c >> 3) * (int)g.ldc + (c & 7) * 8); }
    auto fetch = [&](int nt) {
        const int n0 = nt * PN;
        const long dn = pgrp_delta(g, n0);
        const bf16* wt = W + (n0 + dn) * g.ldb;
#pragma unroll
        for (int i = 0; i < NLD; ++i) wreg[i] = *reinterpret_cast<const bf16x8*>(wt + woff[i]);
        breg = biasp[((n0 + dn) & bias_mask) + (tid & (PN - 1))];
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * 512, row = c / CPR, kc = (c % CPR) * 8;
            *reinterpret_cast<bf16x8*>(Bs + (buf * PN + row) * PB + kc) = wreg[i];
        }
        if (tid < PN) bias_s[buf * PN + tid] = breg;
    };
    fetch(0);
    commit(0);
    __syncthreads();
    if (ntiles > 1) fetch(1);
    for (int nt = 0; nt < ntiles; ++nt) {
        const int cur = nt & 1;
        f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        const bf16* brow = Bs + (cur * PN + wn * 32 + (lane & 31)) * PB + hh * 8;
        // W fragments in groups of 8 LDS reads ahead of their 16 MFMAs (the next group's reads ride under this group's MFMAs)
        constexpr int GRP = NKS < 8 ? NKS : 8;
#pragma unroll
        for (int k0 = 0; k0 < NKS; k0 += GRP) {
            bf16x8 wf[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(brow + (k0 + j) * 16);
            __builtin_amdgcn_sched_group_barrier(0x100, GRP, 0);
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                mma32(acc[0], wf[j], af[0][k0 + j]);            // C^T: rows = the tile's 32 columns n, columns = 32 activation rows
                mma32(acc[1], wf[j], af[1][k0 + j]);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * GRP, 0);
        }
        // + bias, bf16, into the C image: lane = output row, four consecutive columns per accumulator group
        bf16* cimg = Cs + cur * PM * PC;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            bf16* crow = cimg + (wm * 64 + rb * 32 + (lane & 31)) * PC + wn * 32 + 4 * hh;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_s[cur * PN + wn * 32 + 8 * gq + 4 * hh]);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[rb][4 * gq + e] + b4[e]);
                *reinterpret_cast<bf16x4*>(crow + 8 * gq) = o;
            }
        }
        if (nt + 1 < ntiles) commit(cur ^ 1);
        __syncthreads();
        // The loads of tile nt + 2 are issued BEFORE this tile's stores: vmcnt retires in order, so the wait in front of the
        // next commit would otherwise also wait for these stores' full round trip to HBM.
        if (nt + 2 < ntiles) fetch(nt + 2);
        // the finished 256 x 64 tile as whole 128-byte rows
        const int n0 = nt * PN;
        constexpr int NST = PM * PN / 8 / 512;
        bf16x8 cv[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = tid + i * 512, row = c >> 3, ch = c & 7;
            cv[i] = *reinterpret_cast<const bf16x8*>(cimg + row * PC + ch * 8);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NST, 0);
        bf16* ct = C + (long)m0 * g.ldc + n0;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int row = (tid + i * 512) >> 3;
            if (m0 + row < g.M) *reinterpret_cast<bf16x8*>(ct + coff[i]) = cv[i];
        }
    }
}

}  // namespace

// Takes the shapes it is built for (see the dispatch in omr_gemm); OMR_ERR_UNSUPPORTED otherwise.
int omr_gemm_panel_bf16(const GemmArgs& g, hipStream_t s) {
    if (g.K != 128 && g.K != 256) return OMR_ERR_UNSUPPORTED;
    if (g.N % PN || g.M < 200 * PM || g.N < 8 * PN || g.ldc % 8 || ((uintptr_t)g.C & 15)) return OMR_ERR_UNSUPPORTED;      // at least ~one panel per CU
    if (g.grp_operand != 0 && g.grp_operand != 1) return OMR_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)cdiv(g.M, PM)), block(512);
    if (g.K == 256) {
        constexpr size_t shm = (2 * PN * (256 + 8) + 2 * PM * (PN + 8)) * sizeof(bf16) + 2 * PN * sizeof(float);
        static bool attr = false;
        if (!attr) { if (hipFuncSetAttribute((const void*)gemm_panel_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return OMR_ERR_LAUNCH; attr = true; }
        hipLaunchKernelGGL((gemm_panel_kernel<16>), grid, block, shm, s, g);
    } else {
        constexpr size_t shm = (2 * PN * (128 + 8) + 2 * PM * (PN + 8)) * sizeof(bf16) + 2 * PN * sizeof(float);
        static bool attr = false;
        if (!attr) { if (hipFuncSetAttribute((const void*)gemm_panel_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return OMR_ERR_LAUNCH; attr = true; }
        hipLaunchKernelGGL((gemm_panel_kernel<8>), grid, block, shm, s, g);
    }
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
