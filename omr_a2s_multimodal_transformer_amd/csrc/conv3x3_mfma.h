// conv3x3 implicit-GEMM kernel (forward and data gradient) -- shared by conv_bf16.hip / conv_f32.hip so the two dtypes
// compile in parallel.  See conv.hip for the overview.
#pragma once
#include <atomic>
#include <type_traits>

#include "omr_common.h"

namespace omr_conv {

constexpr int NORM_MAX = 128;   // most input channels a normalise-on-load conv may have (LDS copy of the statistics)
constexpr int TW = 32;  // output tile width = one MFMA M-block (32 pixels of one output row)

// Walk tile pixels pix = pix0 + k*DP (k = 0, 1, ...) keeping an incremental (row, col) inside a tile of width IW, in
// batches of G: all G global loads are issued before the first LDS store so their latencies overlap (a plain
// load->store loop serialises on s_waitcnt vmcnt(0) every iteration).
template <int G, typename F, typename LoadFn, typename StoreFn>
__device__ __forceinline__ void staged_walk(int pix0, int npix, int DP, int il0, int jl0, int di, int dj, int IW, LoadFn load, StoreFn store) {
    int il = il0, jl = jl0;
    for (int base = pix0; base < npix; base += G * DP) {
        F v[G];
        int pix = base;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (pix < npix) v[g] = load(il, jl);
            jl += dj; il += di;
            if (jl >= IW) { jl -= IW; ++il; }
            pix += DP;
        }
        pix = base;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (pix < npix) store(pix, v[g]);
            pix += DP;
        }
    }
}

// halo tile geometry, shared by the kernel and its launcher
__host__ __device__ constexpr bool conv_subpix(int SH, int SW, int DH, int DW, int RPW) { return DH == 2 && DW == 2 && SH == 1 && SW == 1 && RPW == 2; }
__host__ __device__ constexpr int conv_halo_h(int TH, int SH, bool subpix) { return subpix ? ((TH - 1) * SH + 3 + 1) / 2 : (TH - 1) * SH + 3; }
__host__ __device__ constexpr int conv_halo_w(int SW, bool subpix) { return subpix ? ((TW - 1) * SW + 3) / 2 : (TW - 1) * SW + 3; }

struct ConvArgs {
    const void* x; const void* w; const float* bias; void* y;
    const float* mean; const float* rstd;        // [B][CIN] fused InstanceNorm apply on load (or null)
    const void* mask; float mask_scale;          // epilogue: y = mask > 0 ? y * scale : 0   (ReLU/dropout backward of the consumer side)
    int B, Hr, Wr, CIN, Ho, Wo, COUT;
    int sh, sw, dh, dw, relu, tiles_w, tiles_h;
    // fused dropout on the output (after ReLU): counter-based mask keyed by the flat NHWC index (or (b, channel))
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed; int drop_channel;
    // fused per-(image, channel) reductions over the STORED output tile (fp64 accumulators stat_ws[B][COUT][2]):
    //   mode 1: {sum y, sum y^2}            -> InstanceNorm statistics of this conv's output (encoder.py:174)
    //   mode 2: {sum g, sum g * xhat}       -> InstanceNorm backward sums, xhat = (stat_x - mean) * rstd at the same position
    // The reduction is DETERMINISTIC: per-thread fp32 partials (fixed tile walk) -> fixed-order fp64 sum over the block's
    // threads -> plain store into the block's own slot stat_ws[b][blockIdx.x][COUT][2]; the consumer (omr_instnorm_finalize /
    // omr_instnorm_bwd_apply) adds the stat_slots slots of an image in index order.  No atomics anywhere.
    //   mode 4: the sums of mode 2 WITHOUT storing the output; mode 5: the output of mode 2 with the InstanceNorm backward
    //           applied in the epilogue,  dx = rstd * (g - s1 - xhat * s2) [* (stat_x > 0) * stat_relu_scale],  s1 / s2 = the
    //           image's {sum g, sum g * xhat} / (Ho * Wo) read from the compact sums behind the slots (omr_instnorm_reduce_sums).
    //           Together: the data gradient of a normalise-on-load conv is taken TWICE (HBM-bound layers: the MFMAs are free)
    //           instead of written as g, read back, and written again by a stand-alone apply pass.
    int stat_mode; double* stat_ws; const void* stat_x; const float* stat_mean; const float* stat_rstd; int stat_slots;
    int stat_relu; float stat_relu_scale;
};

// ------------------------------------------------------------------------------------------------
// 3x3 conv as implicit GEMM.  Block = 256 threads = 4 waves; output tile = (4*RPW) rows x 32 cols x NT couts.
// Wave w owns tile rows [w*RPW, (w+1)*RPW) and all NT/32 cout blocks.  Blocks are PERSISTENT over output tiles
// (grid-stride): when the whole reduction fits one channel chunk (CIN == CK: every 16/32-channel layer, i.e. all the
// full-resolution ones) the weight tile is staged into LDS once per block instead of once per tile.
// MFMA orientation: D[cout][pixel] (weights = A operand) so each lane owns one output pixel and its 16 accumulator
// registers are 4 runs of 4 consecutive couts: bias/ReLU in registers, one 8-byte LDS store per run, then the tile
// leaves LDS as 16-byte row-contiguous global stores (direct 8-byte global stores were measured 15 % slower).
// EPI selects the fused-epilogue code that is compiled in (keeps the plain kernel's register footprint small):
//   0 plain | 1 forward: InstanceNorm statistics of the output | 2 backward: InstanceNorm-backward sums | 3 forward: MixDropout (+ statistics)
// (one kernel with the MixDropout code behind a run-time test was measured 7-10 % slower on the statistics-only launches, two steps in three)
template <typename T, int NT, int RPW, int CK, int SH, int SW, int DH, int DW, bool SINGLE, int EPI>
__global__ __launch_bounds__(256, (SINGLE && NT == 32) ? ((CK == 16 && EPI != 2) ? 4 : 3) : 2) void conv3x3_mfma_kernel(ConvArgs a) {
    typedef typename Frag<T>::type F;
    typedef __attribute__((ext_vector_type(4))) T T4;
    constexpr int VEC = Frag<T>::N;
    constexpr int CKP = CK + VEC;           // pitch (elements): 16-byte odd multiple -> conflict-free b128 reads
    constexpr int NB = NT / 32;
    constexpr int TH = 4 * RPW;
    // data gradient of a stride-(2,2) conv (input zero-dilated both ways): only the REAL halo pixels are staged -- rows / columns
    // of odd tile-local index, (IH, IW) = (5, 17) instead of (10, 34) -- and MFMA blocks are made of one pixel-parity class
    constexpr bool SUBPIX = conv_subpix(SH, SW, DH, DW, RPW);
    constexpr int IH = conv_halo_h(TH, SH, SUBPIX), IW = conv_halo_w(SW, SUBPIX), NPIX = IH * IW;
    constexpr int CPP = CK / VEC;           // 16-byte chunks per pixel per channel chunk
    constexpr int OP = NT + VEC;            // output staging pitch (elements)
    // LDS: [Xs | Ws].  The output staging tile Os aliases Xs; when the weights are re-staged per chunk anyway (!SINGLE) it
    // may also run over Ws, otherwise Xs is sized to hold it.
    constexpr int XS_ELEMS = (SINGLE && TH * TW * OP > NPIX * CKP) ? TH * TW * OP : NPIX * CKP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Xs = reinterpret_cast<T*>(smem_raw);
    T* Os = Xs;
    T* Ws = Xs + XS_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * NT;
    const int Hv = (a.Hr - 1) * DH + 1, Wv = (a.Wr - 1) * DW + 1;
    const T* W = (const T*)a.w;
    constexpr bool single = SINGLE;   // the whole reduction is one channel chunk: weights staged once per block
    const int frow = lane & 31, fk = (lane >> 5) * VEC, hsel = 4 * (lane >> 5);
    // SUBPIX: MFMA blocks of one pixel-parity class (see the tap loop)
    const int sp_rp = wave >> 1;                                                 // row parity class of the wave
    const int sp_row = (wave & 1) * 4 + sp_rp + 2 * (frow >> 4), sp_col = 2 * (frow & 15);   // this lane's tile row / first column

    auto stage_weights = [&](int c0) {      // Ws[n][tap][k] for channel chunk c0; 16-byte chunks, loads batched four at a time
        constexpr int NCH = NT * 9 * CPP, ROUNDS = (NCH + 255) / 256, GB = 4;
#pragma unroll
        for (int r0 = 0; r0 < ROUNDS; r0 += GB) {
            F v[GB];
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int c = tid + (r0 + g) * 256, row = c / CPP, kc = (c % CPP) * VEC, n = n0 + row / 9;
                v[g] = frag_zero<T>();
                if (r0 + g < ROUNDS && c < NCH && n < a.COUT) v[g] = *reinterpret_cast<const F*>(W + ((long)n * 9 + row % 9) * a.CIN + c0 + kc);
            }
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int c = tid + (r0 + g) * 256, row = c / CPP, kc = (c % CPP) * VEC;
                if (r0 + g < ROUNDS && c < NCH) *reinterpret_cast<F*>(Ws + (long)row * CKP + kc) = v[g];
            }
        }
    };

    __shared__ __attribute__((aligned(16))) float sbias[NT];     // bias of this block's couts (read back as float4 runs)
    for (int i = tid; i < NT; i += 256) sbias[i] = (a.bias && n0 + i < a.COUT) ? a.bias[n0 + i] : 0.f;

    // fused InstanceNorm-apply on load: the (mean, rstd) rows of this block's image, staged once (a block sees one image)
    __shared__ __attribute__((aligned(16))) float snorm[2 * NORM_MAX];
    if (a.mean)
        for (int i = tid; i < a.CIN; i += 256) {
            snorm[i] = a.mean[(long)blockIdx.z * a.CIN + i];
            snorm[NORM_MAX + i] = a.rstd[(long)blockIdx.z * a.CIN + i];
        }

    if (single) stage_weights(0);

    // fused reductions: this thread always stores the same VEC-channel chunk, so it keeps fp32 partials in registers; when the
    // block is done with its (single) image the partials are combined in a FIXED order through LDS (the staging tiles are
    // dead by then) and stored into the block's own slot of the fp64 workspace -- bit-identical from run to run.
    // (Summing in the accumulator-to-LDS pass instead -- a lane holds 16 couts of its pixels there, as fp32 -- was measured:
    // 32 partial-sum registers per lane push the 16/32-channel variants over their occupancy budget, 591 -> 682 us.)
    constexpr int CPO = NT / VEC;
    constexpr int NSV = EPI ? VEC : 1;
    float ssum[NSV], ssq[NSV], smu[EPI == 2 ? VEC : 1], srs[EPI == 2 ? VEC : 1];
#pragma unroll
    for (int e = 0; e < NSV; ++e) ssum[e] = ssq[e] = 0.f;
#pragma unroll
    for (int e = 0; e < (EPI == 2 ? VEC : 1); ++e) { smu[e] = 0.f; srs[e] = 1.f; }
    constexpr bool FWD = EPI == 1 || EPI == 3, DROP = EPI == 3;     // forward extras; MixDropout code compiled in
    const bool stat1 = FWD && a.stat_mode == 1, stat2 = EPI == 2 && (a.stat_mode == 2 || a.stat_mode == 4);
    const bool apply5 = EPI == 2 && a.stat_mode == 5, store_y = a.stat_mode != 4;
    auto flush_stats = [&](int bimg) {
        if constexpr (EPI != 0) {
            constexpr int NGRP = 256 / CPO;                       // threads that share a channel chunk
            constexpr int RP = NT + 1;                             // odd pitch: the lanes of a wave write one column
            static_assert((size_t)NGRP * RP * sizeof(float) <= ((size_t)XS_ELEMS + (size_t)NT * 9 * CKP) * sizeof(T), "reduction scratch fits in the (dead) halo + weight tiles");
            float* red = reinterpret_cast<float*>(smem_raw);      // [NGRP][RP], first the sums, then the second moments
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                __syncthreads();                                  // the last tile's store loop is done with Os / pass 0 has been read
                const int kcs = (tid % CPO) * VEC, grp = tid / CPO;
#pragma unroll
                for (int e = 0; e < NSV; ++e) red[grp * RP + kcs + e] = k ? ssq[e] : ssum[e];
                __syncthreads();
                for (int i = tid; i < NT; i += 256) {
                    const int n = n0 + i;
                    double acc2 = 0.0;
                    for (int g = 0; g < NGRP; ++g) acc2 += (double)red[g * RP + i];
                    if (n < a.COUT) {
                        a.stat_ws[(((long)bimg * a.stat_slots + blockIdx.x) * a.COUT + n) * 2 + k] = acc2;
                        if (blockIdx.x == 0)          // slots no block of this launch owns read as zero: the caller need not clear the workspace
                            for (int sl = gridDim.x; sl < a.stat_slots; ++sl) a.stat_ws[(((long)bimg * a.stat_slots + sl) * a.COUT + n) * 2 + k] = 0.0;
                    }
                }
            }
        }
    };

    // persistent schedule: blockIdx.z = image; the gridDim.x blocks of an image walk its tiles with stride gridDim.x, so
    // concurrently running blocks work on adjacent tiles (halo rows shared through L2) and a block only ever sees ONE image
    // (its fused-reduction partials are flushed once, at the end)
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int b = blockIdx.z;
    if constexpr (EPI == 2) {
        if (stat2 || apply5) {
            const double* compact = a.stat_ws + (long)a.B * a.stat_slots * a.COUT * 2;       // [B][COUT][2]: the image's finished sums (mode 5)
            const double inv_hw = 1.0 / ((double)a.Ho * a.Wo);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int n = n0 + (tid % CPO) * VEC + e;
                smu[e] = n < a.COUT ? a.stat_mean[(long)b * a.COUT + n] : 0.f;
                srs[e] = n < a.COUT ? a.stat_rstd[(long)b * a.COUT + n] : 1.f;
                if (apply5) {          // the partial-sum registers are free in this mode: they hold s1 = mean(g), s2 = mean(g * xhat)
                    ssum[e] = n < a.COUT ? (float)(compact[((long)b * a.COUT + n) * 2] * inv_hw) : 0.f;
                    ssq[e] = n < a.COUT ? (float)(compact[((long)b * a.COUT + n) * 2 + 1] * inv_hw) : 0.f;
                }
            }
        }
    }
    // EPI 3, channel-wise MixDropout (Dropout2d): the keep decision depends on (image, cout) only, so the 16 NB decisions of
    // this lane's couts are taken once per block -- bit r of ckeep[j] belongs to accumulator register r of cout block j
    // (same pair hash as omr_dropout: index (b * COUT + n), one hash per even/odd cout pair).
    uint32_t ckeep[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) ckeep[j] = 0xffffu;
    if constexpr (DROP) {
        if (a.drop_thresh && a.drop_channel) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                ckeep[j] = 0;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t hsh = drop_pair_bits(a.drop_seed, ((uint64_t)b * a.COUT + n0 + j * 32 + 8 * (r >> 2) + hsel + (r & 3)) >> 1);
                    ckeep[j] |= (uint32_t)((hsh & 0xffffu) >= a.drop_thresh) << r | (uint32_t)((hsh >> 16) >= a.drop_thresh) << (r + 1);
                }
            }
        }
    }
    // Staging registers.  PF (every multi-chunk variant: those are LDS-limited to two workgroups per CU, so the registers are
    // free): the halo and weight chunks of the NEXT (tile, channel chunk) are loaded into registers before the MFMAs of the
    // current one and written to LDS after them, so a chunk's global latency hides behind a chunk's math instead of a barrier.
    constexpr bool PF = true;
    constexpr int XR = (NPIX + 255) / 256;
    constexpr int WNCH = NT * 9 * CPP, WR = SINGLE ? 1 : (WNCH + 255) / 256;
    F xv[XR][CPP], wv[WR];
    unsigned xok = 0;                                 // bit r: halo pixel of round r is a real input pixel
    const T* X = (const T*)a.x + (long)b * a.Hr * a.Wr * a.CIN;
    auto load_x = [&](int rem, int c0) {
        const int th = rem / a.tiles_w, tw = rem - th * a.tiles_w;
        const int vh0 = th * TH * SH - 1, vw0 = tw * TW * SW - 1;   // virtual (dilated) input origin of the halo
        xok = 0;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int pix = tid + r * 256;
            const int il = pix / IW, jl = pix - il * IW;
            const int vh = vh0 + (SUBPIX ? 2 * il + 1 : il), vw = vw0 + (SUBPIX ? 2 * jl + 1 : jl);   // SUBPIX: tile origins are odd, real pixels sit at odd offsets
            const bool ok = pix < NPIX && vh >= 0 && vh < Hv && vw >= 0 && vw < Wv && (vh & (DH - 1)) == 0 && (vw & (DW - 1)) == 0;
            const T* src = X + ((long)(vh >> (DH >> 1)) * a.Wr + (vw >> (DW >> 1))) * a.CIN + c0;
#pragma unroll
            for (int k = 0; k < CPP; ++k) {
                xv[r][k] = frag_zero<T>();
                if (ok) xv[r][k] = *reinterpret_cast<const F*>(src + k * VEC);
            }
            xok |= (unsigned)ok << r;
        }
    };
    auto store_x = [&](int c0) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int pix = tid + r * 256;
            if (a.mean && ((xok >> r) & 1)) {      // statistics of this block's image from LDS (broadcast 16-byte reads)
#pragma unroll
                for (int k = 0; k < CPP; ++k)
#pragma unroll
                    for (int q4 = 0; q4 < VEC / 4; ++q4) {
                        const f32x4 mu = *reinterpret_cast<const f32x4*>(&snorm[c0 + k * VEC + 4 * q4]);
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(&snorm[NORM_MAX + c0 + k * VEC + 4 * q4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            xv[r][k][4 * q4 + e] = from_f32<T>((to_f32(xv[r][k][4 * q4 + e]) - mu[e]) * rs[e]);
                    }
            }
            if (pix < NPIX)
#pragma unroll
                for (int k = 0; k < CPP; ++k) *reinterpret_cast<F*>(Xs + (long)pix * CKP + k * VEC) = xv[r][k];
        }
    };
    auto load_w = [&](int c0) {
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int c = tid + r * 256, row = c / CPP, kc = (c % CPP) * VEC, n = n0 + row / 9;
            wv[r] = frag_zero<T>();
            if (c < WNCH && n < a.COUT) wv[r] = *reinterpret_cast<const F*>(W + ((long)n * 9 + row % 9) * a.CIN + c0 + kc);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int c = tid + r * 256, row = c / CPP, kc = (c % CPP) * VEC;
            if (c < WNCH) *reinterpret_cast<F*>(Ws + (long)row * CKP + kc) = wv[r];
        }
    };
    if constexpr (PF) {
        if ((int)blockIdx.x < tiles_per_img) { load_x(blockIdx.x, 0); load_w(0); }
    }

    for (int rem = blockIdx.x; rem < tiles_per_img; rem += gridDim.x) {
        const int th = rem / a.tiles_w, tw = rem - th * a.tiles_w;
        const int oh0 = th * TH, ow0 = tw * TW;
        const int vh0 = oh0 * SH - 1;

        f32x16 acc[RPW][NB];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int c0 = 0; c0 < a.CIN; c0 += CK) {
            __syncthreads();                  // previous tile's store loop / previous chunk's MFMAs are done with Xs (and Ws)
            // ---- stage the input halo: thread = one halo pixel per round (ONE bounds test + address for its CPP chunks:
            //      measured 30 % faster than chunk-granular staging on the 16/32-channel layers, which are VALU-limited);
            //      all loads of all rounds are issued before the first LDS store
            if constexpr (!PF) load_x(rem, c0);
            store_x(c0);
            if constexpr (!SINGLE) store_w();
            __syncthreads();
            if constexpr (PF) {
                int nc0 = c0 + CK, nrem = rem;
                if (nc0 >= a.CIN) { nc0 = 0; nrem = rem + gridDim.x; }
                if (nrem < tiles_per_img) { load_x(nrem, nc0); load_w(nc0); }
            }
            // ---- nine shifted GEMMs out of LDS
            // Zero-dilated input (DH = 2: data gradient of a stride-2 conv): a whole halo row is structural zeros whenever its
            // virtual row index is odd, so for output row i only the tap rows kh with (vh0 + i + kh) even contribute --
            // one of three for even output rows, two of three for odd ones.  The test is wave-uniform.
            bool row_live[RPW][3];
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) row_live[i][kh] = DH == 1 || (((vh0 + (wave * RPW + i) * SH + kh) & (DH - 1)) == 0);
            if constexpr (SUBPIX) {
                // Zero-dilated in BOTH directions: three of four halo pixels are structural zeros.  A 32-pixel MFMA block is made of
                // pixels of ONE (row parity, column parity) class -- the wave owns two tile rows of its parity, block cp takes their
                // 16 + 16 columns of parity cp -- so every lane of a block agrees on which taps meet real data, and the dead
                // (tap row, tap column) pairs are skipped for the whole block: 9 MFMAs per 128 pixels and k-step where the
                // row-contiguous blocks (which can only skip tap rows) issue 18.
                const int vw0 = ow0 * SW - 1;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap % 3;
                    if (((vh0 + sp_rp + kh) & 1) != 0) continue;                   // wave-uniform: this tap row is all zeros for the wave's rows
                    const bool lc[2] = {((vw0 + kw) & 1) == 0, ((vw0 + 1 + kw) & 1) == 0};
#pragma unroll
                    for (int kk = 0; kk < CK; kk += KStep<T>::value) {
                        F af[2], bf[NB];
#pragma unroll
                        for (int cp = 0; cp < 2; ++cp)
                            if (lc[cp]) af[cp] = *reinterpret_cast<const F*>(Xs + (long)(((sp_row + kh) >> 1) * IW + ((sp_col + cp + kw) >> 1)) * CKP + kk + fk);
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            bf[j] = *reinterpret_cast<const F*>(Ws + (long)((j * 32 + frow) * 9 + tap) * CKP + kk + fk);
#pragma unroll
                        for (int cp = 0; cp < 2; ++cp) {
                            if (!lc[cp]) continue;
#pragma unroll
                            for (int j = 0; j < NB; ++j) mma32(acc[cp][j], bf[j], af[cp]);
                        }
                    }
                }
            } else
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                bool any_live = false;
#pragma unroll
                for (int i = 0; i < RPW; ++i) any_live = any_live || row_live[i][kh];
                if (!any_live) continue;
#pragma unroll
                for (int kk = 0; kk < CK; kk += KStep<T>::value) {
                    F af[RPW], bf[NB];
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
                        const int pix = ((wave * RPW + i) * SH + kh) * IW + frow * SW + kw;
                        af[i] = *reinterpret_cast<const F*>(Xs + (long)pix * CKP + kk + fk);
                    }
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        bf[j] = *reinterpret_cast<const F*>(Ws + (long)((j * 32 + frow) * 9 + tap) * CKP + kk + fk);
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
                        if (!row_live[i][kh]) continue;
#pragma unroll
                        for (int j = 0; j < NB; ++j) mma32(acc[i][j], bf[j], af[i]);   // D[cout][pixel]
                    }
                }
            }
        }

        // ---- epilogue.  Channel-wise MixDropout is applied here from the block's ckeep bits; the elementwise kind hashes in
        //      the store loop, where the accumulators are dead (registers).  (Starting the accumulator chains from the bias
        //      instead of adding it here was measured: 483 -> 523 us on the plain 32-channel kernel.)
        const float oscale = a.mask ? a.mask_scale : 1.f;
        const bool drop_chan = DROP && a.drop_thresh != 0 && a.drop_channel;
        __syncthreads();                              // every wave is done reading Xs: Os aliases it
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                T* orow = Os + (long)(SUBPIX ? sp_row * TW + sp_col + i : (wave * RPW + i) * TW + frow) * OP + j * 32 + hsel;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    T4 o;
                    const uint32_t k4 = ckeep[j] >> (4 * g);
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(&sbias[j * 32 + 8 * g + hsel]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * g + e] + bq[e];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if constexpr (DROP) {
                            if (drop_chan) v = ((k4 >> e) & 1u) ? v * a.drop_scale : 0.f;
                        } else {
                            v *= oscale;                    // the epilogue mask's 1/(1-p) rides here in fp32: the store loop only selects
                        }
                        o[e] = from_f32<T>(v);
                    }
                    *reinterpret_cast<T4*>(orow + 8 * g) = o;
                }
            }
        __syncthreads();
        T* Y = (T*)a.y + (long)b * a.Ho * a.Wo * a.COUT;
        const T* Mk = a.mask ? (const T*)a.mask + (long)b * a.Ho * a.Wo * a.COUT : nullptr;
        const T* SX = (stat2 || apply5) ? (const T*)a.stat_x + (long)b * a.Ho * a.Wo * a.COUT : nullptr;
#pragma unroll(EPI == 0 ? 8 : 4)
        for (int c = tid; c < TH * TW * CPO; c += 256) {
            const int pl = c / CPO, kc = (c % CPO) * VEC;
            const int oh = oh0 + pl / TW, ow = ow0 + pl % TW, n = n0 + kc;
            if (oh >= a.Ho || ow >= a.Wo || n >= a.COUT) continue;
            F v = *reinterpret_cast<const F*>(Os + (long)pl * OP + kc);
            const long o = ((long)oh * a.Wo + ow) * a.COUT + n;
            if constexpr (DROP) {
                if (a.drop_thresh && !a.drop_channel) {   // elementwise MixDropout after the ReLU, keyed exactly like omr_dropout by the flat NHWC index:
                    // drop_keep() of the VEC consecutive indices base .. base + VEC - 1, ONE hash per element pair (16 random bits
                    // each); base is a multiple of VEC (COUT % VEC == 0), so the pair indices are (base >> 1) + e
                    const uint64_t pbase = ((uint64_t)b * a.Ho * a.Wo * a.COUT + (uint64_t)o) >> 1;      // a multiple of VEC / 2: "+ e" is "^ e"
                    const uint32_t h0 = (uint32_t)pbase ^ (uint32_t)(pbase >> 32) * 0x27D4EB2Fu;          // drop_pair_bits() with the index folding hoisted
                    const uint32_t slo = (uint32_t)a.drop_seed, shi = (uint32_t)(a.drop_seed >> 32);
#pragma unroll
                    for (int e = 0; e < VEC / 2; ++e) {
                        const uint32_t hsh = hash32(slo, shi, h0 ^ (uint32_t)e);
                        v[2 * e] = (hsh & 0xffffu) >= a.drop_thresh ? from_f32<T>(to_f32(v[2 * e]) * a.drop_scale) : from_f32<T>(0.f);
                        v[2 * e + 1] = (hsh >> 16) >= a.drop_thresh ? from_f32<T>(to_f32(v[2 * e + 1]) * a.drop_scale) : from_f32<T>(0.f);
                    }
                }
            }
            if (Mk) {      // keep where the saved activation is > 0: for bf16 and fp32 alike that is "bit pattern > 0 as a signed integer"
                typedef typename std::conditional<sizeof(T) == 2, short, int>::type I;
                typedef __attribute__((ext_vector_type(VEC))) I IV;
                const IV m = *reinterpret_cast<const IV*>(Mk + o);
                const IV keep = m > (I)0;                       // lanes of -1 / 0
                IV bits;
                __builtin_memcpy(&bits, &v, sizeof(bits));
                bits &= keep;
                __builtin_memcpy(&v, &bits, sizeof(bits));
            }
            if constexpr (EPI == 2) {
                if (apply5) {          // InstanceNorm backward on the way out (norm.hip instnorm_bwd_apply_kernel's arithmetic)
                    const F xv = *reinterpret_cast<const F*>(SX + o);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float xf = to_f32(xv[e]);
                        float d = srs[e] * (to_f32(v[e]) - ssum[e] - (xf - smu[e]) * srs[e] * ssq[e]);
                        if (a.stat_relu) d = xf > 0.f ? d * a.stat_relu_scale : 0.f;
                        v[e] = from_f32<T>(d);
                    }
                }
            }
            if (store_y) *reinterpret_cast<F*>(Y + o) = v;
            if constexpr (FWD) {
                if (stat1) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { const float f = to_f32(v[e]); ssum[e] += f; ssq[e] += f * f; }
                }
            } else if constexpr (EPI == 2) {
                if (stat2) {
                    const F xv = *reinterpret_cast<const F*>(SX + o);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { const float f = to_f32(v[e]); ssum[e] += f; ssq[e] += f * ((to_f32(xv[e]) - smu[e]) * srs[e]); }
                }
            }
        }
    }
    if (stat1 || stat2) flush_stats(b);
}

template <typename T, int NT, int RPW, int CK, int SH, int SW, int DH, int DW, bool SINGLE, int EPI> int launch_conv3(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int TH = 4 * RPW;
    constexpr int CKP = CK + Frag<T>::N;
    constexpr bool SUBPIX = conv_subpix(SH, SW, DH, DW, RPW);
    constexpr int IH = conv_halo_h(TH, SH, SUBPIX), IW = conv_halo_w(SW, SUBPIX), NPIX = IH * IW, OP = NT + Frag<T>::N;
    constexpr int XS_ELEMS = (SINGLE && TH * TW * OP > NPIX * CKP) ? TH * TW * OP : NPIX * CKP;
    a.tiles_w = cdiv(a.Wo, TW);
    a.tiles_h = cdiv(a.Ho, TH);
    size_t shm = ((size_t)XS_ELEMS + (size_t)NT * 9 * CKP) * sizeof(T);
    if (!SINGLE && (size_t)TH * TW * OP * sizeof(T) > shm) shm = (size_t)TH * TW * OP * sizeof(T);
    if (shm > 160 * 1024) return OMR_ERR_UNSUPPORTED;
    if (a.COUT % Frag<T>::N) return OMR_ERR_UNSUPPORTED;
    if (a.mean && a.CIN > NORM_MAX) return OMR_ERR_UNSUPPORTED;
    auto kern = conv3x3_mfma_kernel<T, NT, RPW, CK, SH, SW, DH, DW, SINGLE, EPI>;
    const int ny = cdiv(a.COUT, NT);
    const long tiles_per_img = (long)a.tiles_w * a.tiles_h;
    // persistent grid = the block slots the chip really has for this kernel (256 CUs x resident blocks per CU), split
    // evenly over the images: every block is resident from the start, no tail of queued blocks.  The per-instantiation
    // launch constants (dynamic-LDS opt-in, resident blocks per CU) are looked up once; the cache is an atomic whose only
    // transition is 0 -> value, and racing first calls compute the same value (one process drives one GPU).
    static std::atomic<int> occ_cache{0};
    int occv = occ_cache.load(std::memory_order_acquire);
    if (occv == 0) {
        if (shm > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return OMR_ERR_LAUNCH;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, shm) != hipSuccess || occ < 1) occ = 2;
        occv = occ;
        occ_cache.store(occ, std::memory_order_release);
    }
    long gx = (256L * occv + (long)ny * a.B - 1) / ((long)ny * a.B);
    if (gx > tiles_per_img) gx = tiles_per_img;
    if (a.stat_mode) {                       // one workspace slot per block of an image
        if (a.stat_slots < 1) return OMR_ERR_ARG;
        if (gx > a.stat_slots) gx = a.stat_slots;
    }
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, a.B), dim3(256), shm, s, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

// Which epilogue variants exist: forward extras (1) only without dilation, backward extras (2) only with unit stride; the
// fp32 (parity) build has no plain variant at all (it takes 1 or 2 with the features switched off at run time).
template <typename T, int NT, int RPW, int CK, int SH, int SW, int DH, int DW, bool SINGLE> int launch_conv2(const ConvArgs& a, hipStream_t s) {
    constexpr bool can1 = (DH == 1 && DW == 1), can2 = (SH == 1 && SW == 1);
    constexpr bool is_f32 = std::is_same<T, float>::value;
    int epi = (a.stat_mode == 2 || a.stat_mode >= 4) ? 2 : (a.drop_thresh ? 3 : (a.stat_mode == 1 ? 1 : 0));
    if (is_f32 && epi == 0) epi = can1 ? 1 : 2;
    if (epi == 1) {
        if constexpr (can1) return launch_conv3<T, NT, RPW, CK, SH, SW, DH, DW, SINGLE, 1>(a, s);
        else return OMR_ERR_UNSUPPORTED;
    }
    if (epi == 3) {
        if constexpr (can1) return launch_conv3<T, NT, RPW, CK, SH, SW, DH, DW, SINGLE, 3>(a, s);
        else return OMR_ERR_UNSUPPORTED;
    }
    if (epi == 2) {
        if constexpr (can2) return launch_conv3<T, NT, RPW, CK, SH, SW, DH, DW, SINGLE, 2>(a, s);
        else return OMR_ERR_UNSUPPORTED;
    }
    if constexpr (!is_f32) return launch_conv3<T, NT, RPW, CK, SH, SW, DH, DW, SINGLE, 0>(a, s);
    else return OMR_ERR_UNSUPPORTED;
}
template <typename T, int NT, int RPW, int CK, int SH, int SW, int DH, int DW> int launch_conv(const ConvArgs& a, hipStream_t s) {
    if (a.CIN == CK) return launch_conv2<T, NT, RPW, CK, SH, SW, DH, DW, true>(a, s);
    return launch_conv2<T, NT, RPW, CK, SH, SW, DH, DW, false>(a, s);
}

// stride / dilation combinations the encoder needs: forward (1,1) (2,2) (2,1); data gradient = stride 1 with dilation (2,2) / (2,1)
template <typename T, int NT, int SH, int SW, int DH, int DW> int dispatch_conv_ck(const ConvArgs& a, hipStream_t s) {
    constexpr int KS = KStep<T>::value;
    constexpr bool strided = SH > 1 || SW > 1;
    if (a.CIN % KS) return OMR_ERR_UNSUPPORTED;
    if constexpr (strided) {
        return launch_conv<T, NT, 1, KS, SH, SW, DH, DW>(a, s);
    } else {
        if (a.CIN % (2 * KS) == 0) return launch_conv<T, NT, 2, 2 * KS, SH, SW, DH, DW>(a, s);
        return launch_conv<T, NT, 2, KS, SH, SW, DH, DW>(a, s);
    }
}
template <typename T, int NT> int dispatch_conv_nt(const ConvArgs& a, hipStream_t s) {
    if (a.dh == 1 && a.dw == 1) {
        if (a.sh == 1 && a.sw == 1) return dispatch_conv_ck<T, NT, 1, 1, 1, 1>(a, s);
        if (a.sh == 2 && a.sw == 2) return dispatch_conv_ck<T, NT, 2, 2, 1, 1>(a, s);
        if (a.sh == 2 && a.sw == 1) return dispatch_conv_ck<T, NT, 2, 1, 1, 1>(a, s);
        return OMR_ERR_UNSUPPORTED;
    }
    if (a.sh != 1 || a.sw != 1) return OMR_ERR_UNSUPPORTED;
    if (a.dh == 2 && a.dw == 2) return dispatch_conv_ck<T, NT, 1, 1, 2, 2>(a, s);
    if (a.dh == 2 && a.dw == 1) return dispatch_conv_ck<T, NT, 1, 1, 2, 1>(a, s);
    return OMR_ERR_UNSUPPORTED;
}
template <typename T> int dispatch_conv(const ConvArgs& a, hipStream_t s) {
    return a.COUT > 32 ? dispatch_conv_nt<T, 64>(a, s) : dispatch_conv_nt<T, 32>(a, s);
}


}  // namespace omr_conv
