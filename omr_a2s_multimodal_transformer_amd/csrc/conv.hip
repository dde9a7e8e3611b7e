// Convolution kernels for the CNN encoder (reference: src/transformer/encoder.py) on gfx950.
// Activations are NHWC ([B][H][W][C], channel fastest) so that
//   * the 3x3 convs are nine shifted [pixels x Cin] . [Cin x Cout] MFMA GEMMs fed from ONE LDS halo tile,
//   * DepthSepConv2D's 1x1 point_conv is a plain GEMM (gemm.hip) and
//   * the encoder output IS the decoder memory [B, S, C] (model.py:147 flatten+permute) with no copy.
// Weights are [Cout][3][3][Cin] (= torch channels_last storage of the reference's [Cout,Cin,3,3] tensor).
//
//   conv3x3_mfma     forward (stride s, pad 1, optional fused InstanceNorm-apply on the input, bias, ReLU) and,
//                    with flipped/transposed weights + input dilation, the data gradient (transposed conv).
//   conv3x3_wgrad    dW[n][tap][c] += sum_pix dY[pix][n] X[pix+tap][c]  (MFMA, K = pixels, fp32 atomics)
//   conv1_direct     the 1 -> 16 first layer (K = 9: HBM-bound, VALU) forward and weight gradient
//   dwconv3x3        depthwise 3x3 forward / data gradient / weight gradient (HBM-bound, VALU)
#include <atomic>
#include <type_traits>

#include "omr_common.h"
#include "omr_hip.h"

#include "conv3x3_mfma.h"
#include "conv_wgrad.h"

using omr_conv::ConvArgs;
using omr_conv::TW;
int omr_conv3x3_dispatch_bf16(const ConvArgs& a, hipStream_t s);
int omr_conv3x3_dispatch_f32(const ConvArgs& a, hipStream_t s);

namespace {

// ------------------------------------------------------------------------------------------------
// Weight re-layout for the data gradient: Wd[c][8 - tap][n] = W[n][tap][c]   (transposed conv = conv with
// flipped taps and swapped channel roles).
template <typename T>
__global__ void weight_flip_kernel(const T* __restrict__ w, T* __restrict__ wd, int COUT, int CIN) {
    long total = (long)COUT * 9 * CIN;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n = (int)(i % COUT); long q = i / COUT; int tapd = (int)(q % 9); int c = (int)(q / 9);
        wd[i] = w[((long)n * 9 + (8 - tapd)) * CIN + c];
    }
}

// The same for SEVERAL convs in one launch (blockIdx.y = conv): the flipped copies of all 3x3 weights of a model are refreshed
// once per optimizer step, right behind omr_adam, instead of one launch in front of every data gradient of the backward pass.
constexpr int FLIP_MAX = 32;
struct FlipTable { const void* w[FLIP_MAX]; void* wd[FLIP_MAX]; int cout[FLIP_MAX]; int cin[FLIP_MAX]; };
template <typename T>
__global__ void weight_flip_grouped_kernel(FlipTable t) {
    const int g = blockIdx.y, COUT = t.cout[g], CIN = t.cin[g];
    const T* __restrict__ w = (const T*)t.w[g];
    T* __restrict__ wd = (T*)t.wd[g];
    const long total = (long)COUT * 9 * CIN;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n = (int)(i % COUT); long q = i / COUT; int tapd = (int)(q % 9); int c = (int)(q / 9);
        wd[i] = w[((long)n * 9 + (8 - tapd)) * CIN + c];
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 conv.  Workgroup = 2x2 waves; wave (wn, wc) owns the 32 couts x 32 cins
// block (n0 + 32 wn, c0 + 32 wc) for all nine taps (9 x 16 accumulator registers).  K = output pixels:
// the block walks pixel tiles (grid-stride), stages dY[pix][64 n] and the X halo [pix][64 c] in LDS
// and reads k-strided operand fragments element-wise (dtype generic; bf16 tr-reads are a later step).
template <typename T, int TH, int CBN, int CBC, int SH, int SW>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradArgs a) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    constexpr int WN = CBN / 32, WC = CBC / 32, WK = 4 / (WN * WC);   // waves over couts, cins and pixel slices
    constexpr bool TR = std::is_same<T, bf16>::value;       // bf16: transposing LDS reads; fp32: element gathers
    // LDS pitches (elements).  tr path: 4 consecutive pixel rows x 16 dwords must tile the 64 banks -> 64 B rows for 32
    // channels, 192 B rows for 64 channels (conflict-free for unit pixel stride); scalar path: odd dword pitch.
    constexpr int NP = TR ? CBN + 8 : CBN + 4, CP = TR ? CBC + 8 : CBC + 4;   // +16 B: conflict-free 16-byte staging stores, 2-way at worst on the tr reads
    typedef __attribute__((address_space(3))) bf16x4 LdsV4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3, NPIX = IH * IW;   // compile-time tile geometry
    T* Ys = reinterpret_cast<T*>(smem_raw);            // [TH*TW][NP]
    T* Xs = Ys + (long)TH * TW * NP;                   // [IH*IW][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (WN * WC), wn = (wave % (WN * WC)) / WC, wc = wave % WC;
    const int ncb = cdiv(a.CIN, CBC);
    const int n0 = (blockIdx.y / ncb) * CBN, c0 = (blockIdx.y % ncb) * CBC;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ntiles = a.B * a.tiles_h * a.tiles_w;
    const bool do_bias = a.db != nullptr && (blockIdx.y % ncb) == 0;   // one cin-block column of the grid owns the bias sums
    float bsum = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (a.tiles_h * a.tiles_w);
        const int rem = tile % (a.tiles_h * a.tiles_w);
        const int oh0 = (rem / a.tiles_w) * TH, ow0 = (rem % a.tiles_w) * TW;
        const int ih0 = oh0 * SH - 1, iw0 = ow0 * SW - 1;
        const T* X = (const T*)a.x + (long)b * a.Hr * a.Wr * a.CIN;
        const T* DY = (const T*)a.dy + (long)b * a.Ho * a.Wo * a.COUT;
        __syncthreads();
        // ---- stage dY tile [pix][CBN] and X halo [pix][CBC]: thread = one pixel per round (one bounds test + address for
        //      all of its 16-byte chunks); every load of every round is issued before the first LDS store
        {
            constexpr int CPN = CBN / VEC, CPC = CBC / VEC;
            constexpr int RY = (TH * TW + 255) / 256, RX = (NPIX + 255) / 256;
            F vy[RY][CPN], vx[RX][CPC];
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int pix = tid + r * 256;
                const int oh = oh0 + pix / TW, ow = ow0 + pix % TW;
                const bool ok = pix < TH * TW && oh < a.Ho && ow < a.Wo;
                const T* src = DY + ((long)oh * a.Wo + ow) * a.COUT + n0;
#pragma unroll
                for (int k = 0; k < CPN; ++k) {
                    vy[r][k] = frag_zero<T>();
                    if (ok && n0 + k * VEC < a.COUT) vy[r][k] = *reinterpret_cast<const F*>(src + k * VEC);
                }
            }
#pragma unroll
            for (int r = 0; r < RX; ++r) {
                const int pix = tid + r * 256;
                const int il = pix / IW, jl = pix - il * IW;
                const int ih = ih0 + il, iw = iw0 + jl;
                const bool ok = pix < NPIX && ih >= 0 && ih < a.Hr && iw >= 0 && iw < a.Wr;
                const T* src = X + ((long)ih * a.Wr + iw) * a.CIN + c0;
#pragma unroll
                for (int k = 0; k < CPC; ++k) {
                    vx[r][k] = frag_zero<T>();
                    if (ok && c0 + k * VEC < a.CIN) vx[r][k] = *reinterpret_cast<const F*>(src + k * VEC);
                }
                if (ok && a.mean) {
#pragma unroll
                    for (int k = 0; k < CPC; ++k)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const int ch = c0 + k * VEC + e;
                            if (ch < a.CIN) vx[r][k][e] = from_f32<T>((to_f32(vx[r][k][e]) - a.mean[b * a.CIN + ch]) * a.rstd[b * a.CIN + ch]);
                        }
                }
            }
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int pix = tid + r * 256;
                if (pix < TH * TW)
#pragma unroll
                    for (int k = 0; k < CPN; ++k) *reinterpret_cast<F*>(Ys + (long)pix * NP + k * VEC) = vy[r][k];
            }
#pragma unroll
            for (int r = 0; r < RX; ++r) {
                const int pix = tid + r * 256;
                if (pix < NPIX)
#pragma unroll
                    for (int k = 0; k < CPC; ++k) *reinterpret_cast<F*>(Xs + (long)pix * CP + k * VEC) = vx[r][k];
            }
        }
        __syncthreads();
        if (do_bias) {   // bias gradient: column sums of the staged dY tile (thread = channel x pixel phase)
            constexpr int NPH = 256 / CBN;
            const int ch = tid % CBN;
            for (int pix = tid / CBN; pix < TH * TW; pix += NPH) bsum += to_f32(Ys[(long)pix * NP + ch]);
        }
        if constexpr (TR) {
            // bf16: k-major operand fragments straight out of the [pixel][channel] tiles with ds_read_b64_tr_b16.
            // Lane (q, p, cb, h) supplies the address of pixel row q, channels 16cb+4p.. of its 16-lane group's 4x16 block
            // and receives 4 consecutive pixels of channel 16cb + (lane & 15) (verified on hardware, scratch/trtest.hip).
            const int q = (lane & 15) >> 2, chan = ((lane >> 4) & 1) * 16 + (lane & 3) * 4, hh = lane >> 5;
            for (int k0 = wk * 16; k0 < TH * TW; k0 += WK * 16) {
                const int pk = k0 + 8 * hh + q;                // u = 0 block; the u = 1 block is 4 pixels further (same tile row)
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(Ys + (long)pk * NP + wn * 32 + chan));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(Ys + (long)(pk + 4) * NP + wn * 32 + chan));
                const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const int r = pk >> 5, col = pk & 31;
                const T* xrow = Xs + (long)((r * SH) * IW + col * SW) * CP + wc * 32 + chan;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap % 3;
                    const T* xp = xrow + (long)(kh * IW + kw) * CP;
                    const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)xp);
                    const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(xp + (long)4 * SW * CP));
                    const bf16x8 bf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    mma32(acc[tap], af, bf);
                }
            }
        } else {
            const int nl = wn * 32 + (lane & 31), cl = wc * 32 + (lane & 31);
            for (int k0 = wk * KStep<T>::value; k0 < TH * TW; k0 += WK * KStep<T>::value) {
                const int pbase = k0 + (lane >> 5) * VEC;     // VEC consecutive pixels of one tile row
                const int r = pbase / TW, col = pbase % TW;
                F af;
#pragma unroll
                for (int e = 0; e < VEC; ++e) af[e] = Ys[(long)(pbase + e) * NP + nl];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap % 3;
                    F bf;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) bf[e] = Xs[(long)((r * SH + kh) * IW + (col + e) * SW + kw) * CP + cl];
                    mma32(acc[tap], af, bf);
                }
            }
        }
    }
    if (do_bias) {   // combine the per-thread partial sums in LDS first: lanes of one wave hitting the same address serialise badly
        float* red = reinterpret_cast<float*>(smem_raw);
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < CBN) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 256 / CBN; ++k) v += red[k * CBN + tid];
            if (n0 + tid < a.COUT) atomicAdd(&a.db[n0 + tid], v);
        }
    }
    // accumulate: dw[n][tap][c] (fp32 atomics; the grad buffer is zeroed once per step)
    const int c = c0 + wc * 32 + (lane & 31);
    if (c < a.CIN) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 32 + acc_row(r, lane);
                if (n < a.COUT) atomicAdd(&a.dw[((long)n * 9 + tap) * a.CIN + c], acc[tap][r]);
            }
    }
}

template <typename T, int TH, int CBN, int CBC, int SH, int SW> int launch_wgrad3(WgradArgs a, hipStream_t s) {
    a.tiles_w = cdiv(a.Wo, TW);
    a.tiles_h = cdiv(a.Ho, TH);
    constexpr int IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3;
    constexpr bool TR = std::is_same<T, bf16>::value;
    constexpr int NP = TR ? CBN + 8 : CBN + 4, CP = TR ? CBC + 8 : CBC + 4;   // +16 B: conflict-free 16-byte staging stores, 2-way at worst on the tr reads
    size_t shm = ((size_t)TH * TW * NP + (size_t)IH * IW * CP) * sizeof(T);
    if (shm > 160 * 1024) return OMR_ERR_UNSUPPORTED;
    auto kern = conv3x3_wgrad_kernel<T, TH, CBN, CBC, SH, SW>;
    const int gy = cdiv(a.COUT, CBN) * cdiv(a.CIN, CBC);
    const int ntiles = a.B * a.tiles_h * a.tiles_w;
    static std::atomic<int> occ_cache{0};   // resident blocks per CU of this instantiation (0 -> value once; see conv3x3_mfma.h)
    int occv = occ_cache.load(std::memory_order_acquire);
    if (occv == 0) {
        if (shm > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return OMR_ERR_LAUNCH;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, shm) != hipSuccess || occ < 1) occ = 2;
        occv = occ;
        occ_cache.store(occ, std::memory_order_release);
    }
    int gx = (256 * occv + gy - 1) / gy; if (gx < 1) gx = 1; if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), shm, s, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

template <typename T, int TH, int SH, int SW> int launch_wgrad2(const WgradArgs& a, hipStream_t s) {
    if (a.COUT > 32 && a.CIN > 32) return launch_wgrad3<T, TH, 64, 64, SH, SW>(a, s);
    if (a.COUT > 32) return launch_wgrad3<T, TH, 64, 32, SH, SW>(a, s);
    return launch_wgrad3<T, TH, 32, 32, SH, SW>(a, s);
}

// TH1: tile rows for stride 1, TH2: for the strided convs (bigger halo)
template <typename T, int TH1, int TH2> int launch_wgrad(const WgradArgs& a, hipStream_t s) {
    if (a.sh == 1 && a.sw == 1) return launch_wgrad2<T, TH1, 1, 1>(a, s);
    if (a.sh == 2 && a.sw == 2) return launch_wgrad2<T, TH2, 2, 2>(a, s);
    if (a.sh == 2 && a.sw == 1) return launch_wgrad2<T, TH2, 2, 1>(a, s);
    return OMR_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// First layer: Cin = 1 -> COUT (<= 32) with ReLU; K = 9, HBM-bound (2 B in, 2 COUT B out per pixel).
// Thread = one image column: it walks down RC rows of one image with a 3x3 register window (3 new 2-byte loads per
// pixel, the next row's already in flight while this row's 9 COUT FMAs run) and writes its pixel's COUT channels as 16-byte
// stores.  Weights sit in LDS tap-major and are read as broadcast float4s; ~50 VGPRs keep 8 waves per SIMD resident.
template <typename T, int COUT>
__global__ __launch_bounds__(256) void conv1_direct_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, int B,
                                                           int H, int Wd, int relu, int RC) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    __shared__ __attribute__((aligned(16))) float ws[10 * COUT];              // [tap][COUT] then bias
    for (int i = threadIdx.x; i < COUT * 9; i += blockDim.x) ws[(i % 9) * COUT + i / 9] = to_f32(w[i]);
    for (int i = threadIdx.x; i < COUT; i += blockDim.x) ws[9 * COUT + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Wd) return;
    const int chunks = cdiv(H, RC);
    const int b = blockIdx.y / chunks, r0 = (blockIdx.y % chunks) * RC, r1 = min(H, r0 + RC);
    const T* xb = x + (long)b * H * Wd;
    T* yb = y + (long)b * H * Wd * COUT;
    auto load_row = [&](int r, float (&row)[3]) {
        row[0] = row[1] = row[2] = 0.f;
        if (r >= 0 && r < H) {
            const T* xr = xb + (long)r * Wd + j;
            row[1] = to_f32(xr[0]);
            if (j > 0) row[0] = to_f32(xr[-1]);
            if (j + 1 < Wd) row[2] = to_f32(xr[1]);
        }
    };
    float win[3][3], nxt[3];
    load_row(r0 - 1, win[0]);
    load_row(r0, win[1]);
    load_row(r0 + 1, win[2]);
    for (int r = r0; r < r1; ++r) {
        load_row(r + 2, nxt);                                   // in flight behind this row's math
        asm volatile("" ::: "memory");                          // re-read the weights from LDS every row: hoisting all 10 COUT of them costs the occupancy
        float acc[COUT];
#pragma unroll
        for (int n = 0; n < COUT; n += 4) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&ws[9 * COUT + n]);
            acc[n] = bv[0]; acc[n + 1] = bv[1]; acc[n + 2] = bv[2]; acc[n + 3] = bv[3];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float xv = win[t / 3][t % 3];
#pragma unroll
            for (int n = 0; n < COUT; n += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&ws[t * COUT + n]);
                acc[n] += wv[0] * xv; acc[n + 1] += wv[1] * xv; acc[n + 2] += wv[2] * xv; acc[n + 3] += wv[3] * xv;
            }
        }
        F* dst = reinterpret_cast<F*>(yb + ((long)r * Wd + j) * COUT);
#pragma unroll
        for (int v = 0; v < COUT / VEC; ++v) {
            F f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = from_f32<T>(relu ? fmaxf(acc[v * VEC + e], 0.f) : acc[v * VEC + e]);
            dst[v] = f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { win[0][c] = win[1][c]; win[1][c] = win[2][c]; win[2][c] = nxt[c]; }
    }
}
// dW[n][tap] += sum_p dY[p][n] x[p + tap]; db[n] += sum_p dY[p][n].  Workgroup = 3 waves x 64 image columns, wave = tap
// row kh: a thread keeps COUT x 3 partial sums in registers while walking image rows (16-byte dY loads, the next row's in
// flight behind this row's FMAs).  All lanes of a wave then hold sums for the SAME weights, so they fold with wave shuffles;
// one lane per wave adds to LDS, one global fp32 atomic per weight per workgroup.
template <typename T, int COUT>
__global__ __launch_bounds__(192) void conv1_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db, int B, int H, int Wd) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N, NV = COUT / VEC;
    __shared__ float red[COUT * 10];
    for (int i = threadIdx.x; i < COUT * 10; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    const int kh = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 64 + lane;
    float acc[3][COUT], accb[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) { acc[0][n] = acc[1][n] = acc[2][n] = 0.f; accb[n] = 0.f; }
    if (j < Wd) {
        const int rows = B * H;
        // Loads are unconditional (row clamped by the caller, out-of-image tap rows read a valid row and are zeroed by a
        // factor): a load behind a branch makes the compiler's vmcnt bookkeeping pessimistic and the ring collapses into
        // one round trip per row.
        auto load = [&](int row, float (&xv)[3], F (&gv)[NV]) {
            const int i = row % H;
            const int yy = i + kh - 1;
            const bool ok = yy >= 0 && yy < H;
            const float m = ok ? 1.f : 0.f;
            const T* xr = x + (long)(ok ? row + kh - 1 : row) * Wd;
            const int jl = j > 0 ? j - 1 : j, jr = j + 1 < Wd ? j + 1 : j;
            const T xl = xr[jl], xc = xr[j], xrr = xr[jr];          // three unconditional loads; the image border is a factor
            xv[0] = to_f32(xl) * (j > 0 ? m : 0.f);
            xv[1] = to_f32(xc) * m;
            xv[2] = to_f32(xrr) * (j + 1 < Wd ? m : 0.f);
            const F* gp = reinterpret_cast<const F*>(dy + ((long)row * Wd + j) * COUT);
#pragma unroll
            for (int v = 0; v < NV; ++v) gv[v] = gp[v];
        };
        // ring of PD rows in flight per lane (one row = 32 bytes of dy per lane: a single row ahead leaves the kernel waiting
        // on HBM latency at 1.4 TB/s)
        constexpr int PD = 4;
        float xq[PD][3];
        F gq[PD][NV];
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            xq[d][0] = xq[d][1] = xq[d][2] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) gq[d][v] = frag_zero<T>();
            load(min((int)(blockIdx.y + d * gridDim.y), rows - 1), xq[d], gq[d]);
        }
        for (int row0 = blockIdx.y; row0 < rows; row0 += PD * gridDim.y) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int row = row0 + d * gridDim.y;
                if (row >= rows) break;
                float xv[3] = {xq[d][0], xq[d][1], xq[d][2]};
                F gv[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) gv[v] = gq[d][v];
                load(min(row + PD * (int)gridDim.y, rows - 1), xq[d], gq[d]);       // refill this slot: PD rows stay in flight behind the FMAs
#pragma unroll
                for (int v = 0; v < NV; ++v) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float g = to_f32(gv[v][e]);
                        const int n = v * VEC + e;
                        acc[0][n] += g * xv[0]; acc[1][n] += g * xv[1]; acc[2][n] += g * xv[2];
                        if (kh == 1) accb[n] += g;
                    }
                }
            }
        }
    }
    // columns beyond the image hold zeros: every lane takes part in the wave reductions
#pragma unroll
    for (int n = 0; n < COUT; ++n) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const float v = wave_sum(acc[kw][n]);
            if (lane == 0) red[n * 9 + kh * 3 + kw] = v;        // (n, kh, kw) has exactly one writer
        }
        if (kh == 1) {
            const float v = wave_sum(accb[n]);
            if (lane == 0) red[COUT * 9 + n] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COUT * 9; i += blockDim.x) atomicAdd(&dw[i], red[i]);
    if (db) for (int i = threadIdx.x; i < COUT; i += blockDim.x) atomicAdd(&db[i], red[COUT * 9 + i]);
}

// The same weight gradient on the MFMA (bf16, 16 output channels: the benchmark's first layer; the last kernel of a backward pass, so
// its whole duration sits in front of the optimizer):  D[n][tap] += dY^T[n][pixels] . P[pixels][tap]  with K = 16 consecutive pixels of
// an image row per MFMA, P the im2col of the 1-channel image -- column `tap` of P is the image row shifted by the tap, so a lane's
// fragment (8 consecutive pixels of one tap) is one aligned 16-byte read from one of THREE copies of the image halo tile kept in LDS,
// pre-shifted by 0 / 1 / 2 pixels.  Column 9 of P is all ones: D[n][9] is the bias gradient.  dY^T comes out of the pixel-major dY
// tile with ds_read_b64_tr_b16 (lanes 16-31 of a half repeat lanes 0-15: rows 16-31 of D are a copy nobody stores).
// Workgroup = 4 waves on a tile of 8 rows x 32 pixels (a wave: two rows = four MFMAs); persistent over the tiles of its share; LDS is
// 10 KB, so a CU holds many workgroups and their load -> LDS -> MFMA phases overlap without a software pipeline.
typedef __attribute__((address_space(3))) bf16x4 C1LdsV4;
__global__ __launch_bounds__(256) void conv1_wgrad_mfma_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw,
                                                                float* __restrict__ db, int B, int H, int Wd, int tiles_h, int tiles_w) {
    constexpr int TH1 = 8, TW1 = 32, IH1 = TH1 + 2, XP = 40;               // XP: row pitch of the shifted image copies (elements; 80 B)
    __shared__ __attribute__((aligned(16))) bf16 Ys[TH1 * TW1 * 16];        // dY tile, pixel-major 32-byte rows, chunk index ^ (col >> 3) & 1
    __shared__ __attribute__((aligned(16))) bf16 Xs[3][IH1][XP];            // Xs[s][r][c] = image(tile row r - 1, tile col c - 1 + s), zero outside
    __shared__ float red[4][16][10];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = (lane & 15) >> 2, p = lane & 3, hh = lane >> 5, ntap = lane & 31;
    const int kh = ntap / 3, kw = ntap - 3 * kh;                             // valid for ntap < 9
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ntiles = B * tiles_h * tiles_w;
    // global -> registers: two 16-byte dY chunks per thread (pixel tid / 2 + 128 j, chunk tid % 2) and up to four image halo values; the
    // next tile's loads are issued before the current tile's MFMAs
    bf16x8 gv[2];
    bf16 xv[4];
    auto load_tile = [&](int tile) {
        const int b = tile / (tiles_h * tiles_w), rem = tile - b * tiles_h * tiles_w;
        const int th = rem / tiles_w, tw = rem - th * tiles_w, oh0 = th * TH1, ow0 = tw * TW1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pix = (tid >> 1) + 128 * j, oh = oh0 + (pix >> 5), ow = ow0 + (pix & 31);
            gv[j] = zero8;
            if (oh < H && ow < Wd) gv[j] = *reinterpret_cast<const bf16x8*>(dy + (((long)b * H + oh) * Wd + ow) * 16 + (tid & 1) * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                  // shifted copies: 3 x 10 x 32 = 960 values
            const int e = tid + 256 * j, sft = e / (IH1 * TW1), r = (e / TW1) % IH1, c = e % TW1;
            const int ih = oh0 - 1 + r, iw = ow0 - 1 + c + sft;
            xv[j] = (bf16)0.f;
            if (e < 3 * IH1 * TW1 && ih >= 0 && ih < H && iw >= 0 && iw < Wd) xv[j] = x[((long)b * H + ih) * Wd + iw];
        }
    };
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                               // the previous tile's MFMAs are done with the LDS tiles
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pix = (tid >> 1) + 128 * j;
            *reinterpret_cast<bf16x8*>(Ys + pix * 16 + (((tid & 1) ^ ((pix >> 3) & 1)) << 3)) = gv[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j, sft = e / (IH1 * TW1), r = (e / TW1) % IH1, c = e % TW1;
            if (e < 3 * IH1 * TW1) Xs[sft][r][c] = xv[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int row = wave * 2 + (s4 >> 1), col0 = (s4 & 1) * 16;
            // A: dY^T, lane (q, p, hh) addresses pixel col0 + 8 hh + q (+4), channels 4 p .. 4 p + 3 (both 16-lane groups of a half the same)
            const int pa = row * TW1 + col0 + 8 * hh + q;
            const int offa = pa * 16 + ((((p >> 1) ^ (((pa & 31) >> 3) & 1)) << 3) | ((p & 1) << 2));
            const int pb = pa + 4;
            const int offb = pb * 16 + ((((p >> 1) ^ (((pb & 31) >> 3) & 1)) << 3) | ((p & 1) << 2));
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((C1LdsV4*)(Ys + offa));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((C1LdsV4*)(Ys + offb));
            const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            // B: column ntap of the im2col: pixels (row + kh - 1, col0 + 8 hh + 0..7 + kw - 1) = copy kw, tile row row + kh, columns col0 + 8 hh ..
            bf16x8 bfr = ntap == 9 ? ones : zero8;
            if (ntap < 9) bfr = *reinterpret_cast<const bf16x8*>(&Xs[kw][row + kh][col0 + 8 * hh]);
            mma32(acc, af, bfr);
        }
    }
    // D[n][tap]: column = lane & 31 = tap, row(reg) = channel; fold the four waves through LDS, one atomic per value per workgroup
    if (ntap < 10) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = acc_row(r, lane);
            if (n < 16) red[wave][n][ntap] = acc[r];
        }
    }
    __syncthreads();
    if (tid < 160) {
        const int n = tid / 10, t = tid - n * 10;
        const float v = red[0][n][t] + red[1][n][t] + red[2][n][t] + red[3][n][t];
        if (t < 9) atomicAdd(&dw[n * 9 + t], v);
        else if (db) atomicAdd(&db[n], v);
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3, stride 1, pad 1 on NHWC (DepthSepConv2D.depth_conv, encoder.py:56-64).  One thread =
// one pixel x VEC channels.  flip=1 applies the taps mirrored (data gradient).  Optional fused
// InstanceNorm apply on the input and optional epilogue mask (ReLU/dropout backward of the producer).
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, const T* __restrict__ mask, float mask_scale,
                                                        int B, int H, int Wd, int C, int flip) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [9][C] taps (already mirrored when flip) + [C] bias
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
        const int t = i / C, c = i % C;
        wsm[i] = to_f32(w[c * 9 + (flip ? 8 - t : t)]);
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) wsm[9 * C + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int cv = C / VEC;
    const long total = (long)B * H * Wd * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC; long p = i / cv;
        const int j = (int)(p % Wd); long q = p / Wd; const int ii = (int)(q % H); const long b = q / H;
        F xv[9];
        bool ok[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {          // issue all nine 16-byte loads before using any
            const int yy = ii + t / 3 - 1, xx = j + t % 3 - 1;
            ok[t] = yy >= 0 && yy < H && xx >= 0 && xx < Wd;
            if (ok[t]) xv[t] = *reinterpret_cast<const F*>(x + ((b * H + yy) * Wd + xx) * C + c);
        }
        float s[VEC], mu[VEC], rs[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            s[e] = wsm[9 * C + c + e];
            mu[e] = mean ? mean[b * C + c + e] : 0.f;
            rs[e] = rstd ? rstd[b * C + c + e] : 1.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!ok[t]) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += wsm[t * C + c + e] * ((to_f32(xv[t][e]) - mu[e]) * rs[e]);
        }
        F o;
        if (mask) {
            const F mk = *reinterpret_cast<const F*>(mask + p * C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(to_f32(mk[e]) > 0.f ? s[e] * mask_scale : 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(s[e]);
        }
        *reinterpret_cast<F*>(y + p * C + c) = o;
    }
}

// Row-walking form of the same op for the DSC blocks (16 x 256 maps, 128-256 channels: tensors of 30-70 MB where the
// per-pixel kernel above is latency-bound at ~1.3 TB/s).  Thread = (image column, channel group): it walks RC rows of one
// image with a 3x3 register window of the (normalised, zero-padded) input, so a pixel costs 3 new 16-byte loads
// (two rows ahead are in flight behind this row's FMAs) instead of 9, and the InstanceNorm apply is paid once per loaded element
// instead of once per tap.  Taps and bias sit in LDS as above.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_walk_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, const T* __restrict__ mask,
                                                             float mask_scale, int B, int H, int Wd, int C, int flip, int RC) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    // taps [9][C] in the compute dtype (already mirrored when flip), read as ONE 16-byte fragment per tap per thread: lanes
    // step 16 bytes, conflict-free, where scalar fp32 reads at a 32-byte lane stride were 8-way bank conflicted and
    // dominated the kernel.  The fp32 bias follows (2 x 16 bytes per thread).
    extern __shared__ __attribute__((aligned(16))) unsigned char wraw[];
    T* wt = reinterpret_cast<T*>(wraw);
    float* wbias = reinterpret_cast<float*>(wraw + (size_t)9 * C * sizeof(T));
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
        const int t = i / C, c = i % C;
        wt[i] = w[c * 9 + (flip ? 8 - t : t)];
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) wbias[i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int cv = C / VEC;
    const int c = (threadIdx.x % cv) * VEC, j = blockIdx.x * (blockDim.x / cv) + threadIdx.x / cv;
    if (j >= Wd) return;
    const int chunks = cdiv(H, RC);
    const int b = blockIdx.y / chunks, r0 = (blockIdx.y % chunks) * RC, r1 = min(H, r0 + RC);
    float rs[VEC], nb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        rs[e] = rstd ? rstd[b * C + c + e] : 1.f;
        nb[e] = mean ? -mean[b * C + c + e] * rs[e] : 0.f;
    }
    const T* xb = x + (long)b * H * Wd * C + c;
    const bool cl = j > 0, cr = j + 1 < Wd;
    // raw row r of the three columns j-1, j, j+1 (zero fragments outside the image); `ok` tells convert() which are real
    auto fetch = [&](int r, F (&raw)[3], bool& ok) {
        ok = r >= 0 && r < H;
        raw[0] = raw[1] = raw[2] = frag_zero<T>();
        if (ok) {
            const T* xr = xb + ((long)r * Wd + j) * C;
            raw[1] = *reinterpret_cast<const F*>(xr);
            if (cl) raw[0] = *reinterpret_cast<const F*>(xr - C);
            if (cr) raw[2] = *reinterpret_cast<const F*>(xr + C);
        }
    };
    const bool norm = mean != nullptr;
    auto convert = [&](const F (&raw)[3], bool ok, F (&row)[3]) {   // normalise (rounded to T, as the MFMA convs do); padding stays exactly 0
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const bool v = ok && (kw == 1 || (kw == 0 ? cl : cr));
            if (norm) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) row[kw][e] = v ? from_f32<T>(fmaf(to_f32(raw[kw][e]), rs[e], nb[e])) : from_f32<T>(0.f);
            } else {
                row[kw] = raw[kw];                                  // fetch() already zero-filled what lies outside
            }
        }
    };
    F win[3][3], rawa[3], rawb[3];
    bool oka, okb;
    fetch(r0 - 1, rawa, oka); convert(rawa, oka, win[0]);
    fetch(r0, rawa, oka); convert(rawa, oka, win[1]);
    fetch(r0 + 1, rawa, oka); convert(rawa, oka, win[2]);
    fetch(r0 + 2, rawa, oka);                                    // two rows of loads stay in flight behind the math
    for (int r = r0; r < r1; ++r) {
        fetch(r + 3, rawb, okb);
        asm volatile("" ::: "memory");                          // keep the taps in LDS (72+ registers otherwise)
        float s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(wbias + c + e);
            s[e] = bv[0]; s[e + 1] = bv[1]; s[e + 2] = bv[2]; s[e + 3] = bv[3];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const F wv = *reinterpret_cast<const F*>(wt + t * C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] = fmaf(to_f32(wv[e]), to_f32(win[t / 3][t % 3][e]), s[e]);
        }
        const long p = ((long)b * H + r) * Wd + j;
        F o;
        if (mask) {
            const F mk = *reinterpret_cast<const F*>(mask + p * C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(to_f32(mk[e]) > 0.f ? s[e] * mask_scale : 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(s[e]);
        }
        *reinterpret_cast<F*>(y + p * C + c) = o;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { win[0][kw] = win[1][kw]; win[1][kw] = win[2][kw]; }
        convert(rawa, oka, win[2]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) rawa[kw] = rawb[kw];
        oka = okb;
    }
}

// Tile form of the same op (the DSC blocks' 16 x 256 maps): the walker above keeps only two rows of loads in flight per thread
// and fetches every input element three times (columns j-1, j, j+1), so its 8-row walk is a chain of exposed latencies
// (1.4 TB/s).  Here a workgroup stages an (8+2) x (TC+2) x C halo tile through LDS -- every element requested once, ALL of a
// thread's requests in flight together, the InstanceNorm apply and the zero padding done on the way in -- and then each
// thread produces its (column, channel group)'s 8 outputs from 9 conflict-free 16-byte LDS reads per output against taps
// held in registers as fp32 (one contiguous 9 x VEC run of the [C][9] weight per thread).
constexpr int DW_TR = 8;      // output rows per tile
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_tile_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, const T* __restrict__ mask,
                                                             float mask_scale, int B, int H, int Wd, int C, int flip) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N, NLD = 16;                    // NLD: 16-byte halo chunks per thread (host guarantees the tile fits)
    extern __shared__ __attribute__((aligned(16))) unsigned char traw[];
    T* tile = reinterpret_cast<T*>(traw);                         // [DW_TR + 2][TC + 2][C]
    const int cv = C / VEC, TC = 256 / cv, IWt = TC + 2;
    const int tid = threadIdx.x, cg = tid % cv, col = tid / cv, c = cg * VEC;
    const int tiles_h = cdiv(H, DW_TR);
    const int b = blockIdx.y / tiles_h, r0 = (blockIdx.y % tiles_h) * DW_TR, j0 = blockIdx.x * TC;
    const T* xb = x + (long)b * H * Wd * C;
    // ---- all halo requests of this thread, then the taps, before anything is consumed
    const int nchunk = (DW_TR + 2) * IWt * cv;
    F ld[NLD];
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ch = tid + i * 256, pp = ch / cv, gi = ch - pp * cv;
        const int ti = pp / IWt, tj = pp - ti * IWt, r = r0 - 1 + ti, j = j0 - 1 + tj;
        const bool ok = ch < nchunk && r >= 0 && r < H && j >= 0 && j < Wd;
        ld[i] = frag_zero<T>();
        if (ok) ld[i] = *reinterpret_cast<const F*>(xb + ((long)r * Wd + j) * C + gi * VEC);
        okbits |= (unsigned)ok << i;
    }
    F wraw[9];                                                    // w[c*9 .. c*9 + 9*VEC): element e*9 + t is tap t of channel c + e
#pragma unroll
    for (int i = 0; i < 9; ++i) wraw[i] = *reinterpret_cast<const F*>(w + (long)c * 9 + i * VEC);
    float bv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bv[e] = bias ? bias[c + e] : 0.f;
    // ---- normalise on the way into LDS (rounded to T like the MFMA convs; padding stays exactly 0)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ch = tid + i * 256;
        if (ch >= nchunk) break;
        if (mean && ((okbits >> i) & 1)) {
            const int gi = ch % cv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float rs = rstd[b * C + gi * VEC + e], nb = -mean[b * C + gi * VEC + e] * rs;
                ld[i][e] = from_f32<T>(fmaf(to_f32(ld[i][e]), rs, nb));
            }
        }
        *reinterpret_cast<F*>(tile + (long)ch * VEC) = ld[i];
    }
    float wt[9][VEC];
    auto unpack = [&](auto fl) {                                  // (mirrored taps for the data gradient) -- indices are compile-time
        constexpr bool FL = decltype(fl)::value;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                constexpr int dummy = 0; (void)dummy;
                const int idx = e * 9 + (FL ? 8 - t : t);
                wt[t][e] = to_f32(wraw[idx / VEC][idx % VEC]);
            }
    };
    if (flip) unpack(std::true_type()); else unpack(std::false_type());
    __syncthreads();
    const int j = j0 + col;
    if (j >= Wd) return;
#pragma unroll 2
    for (int rr = 0; rr < DW_TR; ++rr) {
        const int r = r0 + rr;
        if (r >= H) break;
        float sacc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sacc[e] = bv[e];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const F xv = *reinterpret_cast<const F*>(tile + ((long)((rr + t / 3) * IWt + col + t % 3) * cv + cg) * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) sacc[e] = fmaf(wt[t][e], to_f32(xv[e]), sacc[e]);
        }
        const long p = ((long)b * H + r) * Wd + j;
        F o;
        if (mask) {
            const F mk = *reinterpret_cast<const F*>(mask + p * C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(to_f32(mk[e]) > 0.f ? sacc[e] * mask_scale : 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(sacc[e]);
        }
        *reinterpret_cast<F*>(y + p * C + c) = o;
    }
}

// Weight / bias gradient on the same tile: dW[c][tap] += sum_p dY[p][c] xin[p+tap][c], db[c] += sum_p dY[p][c].  The halo tile
// of the (normalised) input goes through LDS as above, the thread's 8 dY fragments ride in registers with it (one round
// trip for everything), 80 fp32 partial sums per thread.  Fold: the columns a wave holds for one channel group sit 16 / 32
// lanes apart -> v_permlane16_swap / v_permlane32_swap + add; the four waves through LDS; one atomic per weight per
// workgroup.  (The row walker below keeps a whole image column per thread: 512 workgroups, 7x its HBM time.)
template <typename T, int NLD>
__global__ __launch_bounds__(256, 2) void dwconv3x3_wgrad_tile_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd, int B, int H, int Wd, int C) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;                               // NLD: 16-byte halo chunks per thread (12 for the 128-channel bf16 tile, else 16)
    extern __shared__ __attribute__((aligned(16))) unsigned char traw[];
    T* tile = reinterpret_cast<T*>(traw);                         // [DW_TR + 2][TC + 2][C]; afterwards the cross-wave fold scratch
    const int cv = C / VEC, TC = 256 / cv, IWt = TC + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cg = tid % cv, col = tid / cv, c = cg * VEC;
    const int tiles_h = cdiv(H, DW_TR), tiles_w = cdiv(Wd, TC), ntiles = B * tiles_h * tiles_w;
    const int nchunk = (DW_TR + 2) * IWt * cv;
    float acc[10][VEC];                                           // [tap 0..8 | bias][channel], kept over all tiles of this workgroup
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    // persistent over tiles: the fold and the atomics at the end are paid once per workgroup (a memory-side float atomic
    // serialises per cache line -- one per tile was 1.3 M atomics onto 40 lines, 3x the time of everything else)
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int b = tile_id / (tiles_h * tiles_w), rem = tile_id - b * tiles_h * tiles_w;
        const int r0 = (rem / tiles_w) * DW_TR, j0 = (rem % tiles_w) * TC, j = j0 + col;
        const T* xb = x + (long)b * H * Wd * C;
        F ld[NLD];
        unsigned okbits = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ch = tid + i * 256, pp = ch / cv, gi = ch - pp * cv;
            const int ti = pp / IWt, tj = pp - ti * IWt, r = r0 - 1 + ti, jj = j0 - 1 + tj;
            const bool ok = ch < nchunk && r >= 0 && r < H && jj >= 0 && jj < Wd;
            ld[i] = frag_zero<T>();
            if (ok) ld[i] = *reinterpret_cast<const F*>(xb + ((long)r * Wd + jj) * C + gi * VEC);
            okbits |= (unsigned)ok << i;
        }
        const T* dyp = dy + (((long)b * H + r0) * Wd + j) * C + c;   // this thread's dY fragments, one row ahead of their use
        auto load_gy = [&](int rr) { return (j < Wd && r0 + rr < H) ? *reinterpret_cast<const F*>(dyp + (long)rr * Wd * C) : frag_zero<T>(); };
        F gnext = load_gy(0);
        __syncthreads();                                          // the previous tile's reads are done
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ch = tid + i * 256;
            if (ch < nchunk) {
                if (mean && ((okbits >> i) & 1)) {
                    const int gi = ch % cv;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float rs = rstd[b * C + gi * VEC + e], nb = -mean[b * C + gi * VEC + e] * rs;
                        ld[i][e] = from_f32<T>(fmaf(to_f32(ld[i][e]), rs, nb));
                    }
                }
                *reinterpret_cast<F*>(tile + (long)ch * VEC) = ld[i];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int rr = 0; rr < DW_TR; ++rr) {                      // zero dY fragments (rows / columns past the image) add nothing
            const F gcur = gnext;
            if (rr + 1 < DW_TR) gnext = load_gy(rr + 1);
            float g[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { g[e] = to_f32(gcur[e]); acc[9][e] += g[e]; }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const F xv = *reinterpret_cast<const F*>(tile + ((long)((rr + t / 3) * IWt + col + t % 3) * cv + cg) * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[t][e] = fmaf(g[e], to_f32(xv[e]), acc[t][e]);
            }
        }
    }
    // ---- fold the columns of this wave that share the channel group (lanes cv apart: cv = 16 or 32; cv >= 64: one column per wave)
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (cv <= 32) acc[t][e] += __shfl_xor(acc[t][e], 32, 64);
            if (cv == 16) acc[t][e] += __shfl_xor(acc[t][e], 16, 64);
        }
    // ---- the waves through LDS (the tile is dead), then one atomic per weight per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(traw);                  // [wave-slot][cg][10][VEC]
    const bool wide = cv >= 64;                                   // a column spans whole waves: every thread is the only holder of its (column, group)
    const int slot = wide ? col : wave, nslot = wide ? TC : 4;
    if (wide || lane < cv) {
        float* rp = red + ((long)slot * cv + cg) * 10 * VEC;
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int e = 0; e < VEC; ++e) rp[t * VEC + e] = acc[t][e];
    }
    __syncthreads();
    for (int i = tid; i < cv * 10 * VEC; i += 256) {
        float v = 0.f;
        for (int sl = 0; sl < nslot; ++sl) v += red[(long)sl * cv * 10 * VEC + i];
        const int g2 = i / (10 * VEC), rem = i - g2 * 10 * VEC, t = rem / VEC, e = rem - t * VEC, ch = g2 * VEC + e;
        if (t < 9) atomicAdd(&dw[(long)ch * 9 + t], v);
        else if (db) atomicAdd(&db[ch], v);
    }
}

// dW[c][tap] += sum_p dY[p][c] xin[p+tap][c];  db[c] += sum_p dY[p][c].
// Thread = (image column j, channel group): it walks DOWN the rows of one image with a 3x3 register window of the
// normalised input (kept in the compute dtype) -- per pixel 3 new 16-byte x loads + 1 dY load, issued a row ahead of their
// use, for 9*VEC FMAs.  The 10*VEC partial sums stay in registers for the whole column (splitting the rows over more
// workgroups was tried: the 80-value fold below then dominates); wave shuffles fold the columns, then LDS and one global
// atomic per weight per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, int B, int H, int Wd, int C) {
    typedef typename Frag<T>::type F;
    constexpr int VEC = Frag<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [C][10]
    for (int i = threadIdx.x; i < C * 10; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    const int ncg = C / VEC, cpb = blockDim.x / ncg;
    const int cg = threadIdx.x % ncg, j = blockIdx.x * cpb + threadIdx.x / ncg, b = blockIdx.y;
    float acc[10][VEC];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    if (j < Wd) {
        const bool norm = mean != nullptr;
        float rs[VEC], nb[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            rs[e] = rstd ? rstd[(long)b * C + cg * VEC + e] : 1.f;
            nb[e] = mean ? -mean[(long)b * C + cg * VEC + e] * rs[e] : 0.f;
        }
        const T* xb = x + (long)b * H * Wd * C + cg * VEC;
        const T* dyp = dy + ((long)b * H * Wd + j) * C + cg * VEC;
        const long rstride = (long)Wd * C;
        const bool cl = j > 0, cr = j + 1 < Wd;
        auto fetch = [&](int r, F (&raw)[3], bool& ok) {
            ok = r >= 0 && r < H;
            raw[0] = raw[1] = raw[2] = frag_zero<T>();
            if (ok) {
                const T* xr = xb + ((long)r * Wd + j) * C;
                raw[1] = *reinterpret_cast<const F*>(xr);
                if (cl) raw[0] = *reinterpret_cast<const F*>(xr - C);
                if (cr) raw[2] = *reinterpret_cast<const F*>(xr + C);
            }
        };
        auto convert = [&](const F (&raw)[3], bool ok, F (&row)[3]) {   // zero padding lives in the normalised space
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool v = ok && (kw == 1 || (kw == 0 ? cl : cr));
                if (norm) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) row[kw][e] = v ? from_f32<T>(fmaf(to_f32(raw[kw][e]), rs[e], nb[e])) : from_f32<T>(0.f);
                } else {
                    row[kw] = raw[kw];
                }
            }
        };
        F win[3][3], raw[3], gcur, gnext = frag_zero<T>();
        bool ok;
        fetch(-1, raw, ok); convert(raw, ok, win[0]);
        fetch(0, raw, ok); convert(raw, ok, win[1]);
        fetch(1, raw, ok); convert(raw, ok, win[2]);
        gcur = *reinterpret_cast<const F*>(dyp);
        for (int r = 0; r < H; ++r) {
            fetch(r + 2, raw, ok);                               // next row's operands fly behind this row's FMAs
            if (r + 1 < H) gnext = *reinterpret_cast<const F*>(dyp + (long)(r + 1) * rstride);
            float g[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { g[e] = to_f32(gcur[e]); acc[9][e] += g[e]; }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[t][e] = fmaf(g[e], to_f32(win[t / 3][t % 3][e]), acc[t][e]);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) { win[0][kw] = win[1][kw]; win[1][kw] = win[2][kw]; }
            convert(raw, ok, win[2]);
            gcur = gnext;
        }
    }
    // fold the columns that share this lane's channel group (lanes cg, cg+ncg, ...), then LDS, then global
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = acc[t][e];
            for (int o = ncg; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < ncg) atomicAdd(&red[(cg * VEC + e) * 10 + t], v);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 10; i += blockDim.x) {
        const int c = i / 10, tt = i % 10;
        if (tt < 9) atomicAdd(&dw[c * 9 + tt], red[i]);
        else if (db) atomicAdd(&db[c], red[i]);
    }
}


inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace

#define DISPATCH_T(dtype, CALL)                         \
    if ((dtype) == OMR_F32) { typedef float T; CALL; }  \
    else if ((dtype) == OMR_BF16) { typedef bf16 T; CALL; } \
    else return OMR_ERR_UNSUPPORTED;

extern "C" int omr_conv3x3_fwd(int dtype, const void* x, const void* w, const float* bias, void* y, const float* in_mean, const float* in_rstd,
                               const void* out_mask, float mask_scale, int B, int H, int W, int CIN, int COUT, int stride_h, int stride_w,
                               int dil_h, int dil_w, int Ho, int Wo, int relu, float drop_p, unsigned long long drop_seed,
                               int drop_channel_mode, int stat_mode, double* stat_ws, int stat_slots, const void* stat_x,
                               const float* stat_mean, const float* stat_rstd, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || CIN <= 0 || COUT <= 0 || !x || !w || (!y && stat_mode != 4)) return OMR_ERR_ARG;
    if (stride_h < 1 || stride_w < 1 || dil_h < 1 || dil_w < 1 || dil_h > 2 || dil_w > 2) return OMR_ERR_ARG;
    if ((stride_h > 1 || stride_w > 1) && (dil_h > 1 || dil_w > 1)) return OMR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (drop_p < 0.f || drop_p >= 1.f || stat_mode < 0 || stat_mode == 3 || stat_mode > 5) return OMR_ERR_ARG;
    if (stat_mode && (!stat_ws || stat_slots < 1)) return OMR_ERR_ARG;
    if (stat_mode >= 2 && (!stat_x || !stat_mean || !stat_rstd)) return OMR_ERR_ARG;
    if (stat_mode >= 4 && (out_mask || bias || drop_p > 0.f)) return OMR_ERR_ARG;      // data-gradient modes: `relu` / mask_scale describe stat_x's ReLU mask
    if (CIN == 1) {
        if (drop_p > 0.f || stat_mode) return OMR_ERR_UNSUPPORTED;
        if (dil_h != 1 || dil_w != 1 || stride_h != 1 || stride_w != 1 || in_mean || out_mask || Ho != H || Wo != W) return OMR_ERR_UNSUPPORTED;
        const int RC = 32;                                   // rows per workgroup: 2 halo rows per 32 re-read
        dim3 g1(cdiv(W, 256), B * cdiv(H, RC));
        DISPATCH_T(dtype, {
            if (COUT == 16) hipLaunchKernelGGL((conv1_direct_kernel<T, 16>), g1, 256, 0, s, (const T*)x, (const T*)w, bias, (T*)y, B, H, W, relu, RC);
            else if (COUT == 32) hipLaunchKernelGGL((conv1_direct_kernel<T, 32>), g1, 256, 0, s, (const T*)x, (const T*)w, bias, (T*)y, B, H, W, relu, RC);
            else return OMR_ERR_UNSUPPORTED;
        });
        OMR_CHECK_LAUNCH();
        return OMR_OK;
    }
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.mean = in_mean; a.rstd = in_rstd; a.mask = out_mask; a.mask_scale = mask_scale;
    a.B = B; a.Hr = H; a.Wr = W; a.CIN = CIN; a.Ho = Ho; a.Wo = Wo; a.COUT = COUT;
    a.sh = stride_h; a.sw = stride_w; a.dh = dil_h; a.dw = dil_w; a.relu = relu; a.tiles_w = a.tiles_h = 0;
    a.drop_thresh = OMR_DROP_THRESH16(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.drop_seed = drop_seed;
    a.drop_channel = drop_channel_mode;
    a.stat_mode = stat_mode; a.stat_ws = stat_ws; a.stat_x = stat_x; a.stat_mean = stat_mean; a.stat_rstd = stat_rstd; a.stat_slots = stat_slots;
    a.stat_relu = 0; a.stat_relu_scale = 1.f;
    if (stat_mode >= 4) { a.stat_relu = relu; a.stat_relu_scale = mask_scale; a.relu = 0; a.mask_scale = 1.f; }
    if (dtype == OMR_BF16) return omr_conv3x3_dispatch_bf16(a, s);
    if (dtype == OMR_F32) return omr_conv3x3_dispatch_f32(a, s);
    return OMR_ERR_UNSUPPORTED;
}

/* upper bound on the blocks per image of the persistent conv grid (<= 8 resident 256-thread blocks per CU on 256 CUs, and
 * never more than the 4-row x 32-column tiles of an image): the slot count of the fused-statistics workspace */
extern "C" int omr_conv3x3_stat_slots(int B, int Ho, int Wo) {
    if (B <= 0 || Ho <= 0 || Wo <= 0) return OMR_ERR_ARG;
    const long tiles = (long)cdiv(Ho, 4) * cdiv(Wo, TW), slots = (256L * 8 + B - 1) / B;
    return (int)(tiles < slots ? tiles : slots);
}

extern "C" int omr_conv3x3_weight_flip(int dtype, const void* w, void* wd, int COUT, int CIN, void* stream) {
    if (COUT <= 0 || CIN <= 0) return OMR_ERR_ARG;
    long total = (long)COUT * 9 * CIN;
    DISPATCH_T(dtype, hipLaunchKernelGGL((weight_flip_kernel<T>), ew_grid(total), 256, 0, (hipStream_t)stream, (const T*)w, (T*)wd, COUT, CIN));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_conv3x3_weight_flip_grouped(int dtype, int n, const omr_flip_desc* descs, void* stream) {
    if (n < 0 || (n && !descs)) return OMR_ERR_ARG;
    for (int base = 0; base < n; base += FLIP_MAX) {
        FlipTable t = {};
        const int cnt = n - base < FLIP_MAX ? n - base : FLIP_MAX;
        long most = 0;
        for (int i = 0; i < cnt; ++i) {
            const omr_flip_desc& d = descs[base + i];
            if (!d.w || !d.wd || d.cout <= 0 || d.cin <= 0) return OMR_ERR_ARG;
            t.w[i] = d.w; t.wd[i] = d.wd; t.cout[i] = d.cout; t.cin[i] = d.cin;
            const long total = (long)d.cout * 9 * d.cin;
            if (total > most) most = total;
        }
        const dim3 grid((unsigned)((most + 255) / 256 < 64 ? (most + 255) / 256 : 64), (unsigned)cnt);
        DISPATCH_T(dtype, hipLaunchKernelGGL((weight_flip_grouped_kernel<T>), grid, 256, 0, (hipStream_t)stream, t));
    }
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_conv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, const float* in_mean, const float* in_rstd, int B, int H,
                                 int W, int CIN, int COUT, int stride_h, int stride_w, int Ho, int Wo, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || CIN <= 0 || COUT <= 0 || !x || !dy || !dw) return OMR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (CIN == 1) {
        if (stride_h != 1 || stride_w != 1 || in_mean) return OMR_ERR_UNSUPPORTED;
        int gy = B * H; if (gy > 64) gy = 64;      // few, long-lived blocks: each ends with 10 COUT atomics onto the same five cache lines
        dim3 grid(cdiv(W, 64), gy);
        if (dtype == OMR_BF16 && COUT == 16 && (((uintptr_t)dy) & 15) == 0) {
            const int th = cdiv(H, 8), tw = cdiv(W, 32);
            long nt = (long)B * th * tw;
            const int nblk = (int)(nt < 256 * 8 ? nt : 256 * 8);           // persistent (8 workgroups per CU): each ends with 160 atomics onto the same cache lines
            hipLaunchKernelGGL(conv1_wgrad_mfma_kernel, dim3(nblk), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, dw, db, B, H, W, th, tw);
            OMR_CHECK_LAUNCH();
            return OMR_OK;
        }
        DISPATCH_T(dtype, {
            if (COUT == 16) hipLaunchKernelGGL((conv1_wgrad_kernel<T, 16>), grid, 192, 0, s, (const T*)x, (const T*)dy, dw, db, B, H, W);
            else if (COUT == 32) hipLaunchKernelGGL((conv1_wgrad_kernel<T, 32>), grid, 192, 0, s, (const T*)x, (const T*)dy, dw, db, B, H, W);
            else return OMR_ERR_UNSUPPORTED;
        });
        OMR_CHECK_LAUNCH();
        return OMR_OK;
    }
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (CIN % vec || COUT % vec) return OMR_ERR_UNSUPPORTED;
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.db = db; a.mean = in_mean; a.rstd = in_rstd; a.B = B; a.Hr = H; a.Wr = W; a.CIN = CIN; a.Ho = Ho; a.Wo = Wo;
    a.COUT = COUT; a.sh = stride_h; a.sw = stride_w; a.tiles_w = a.tiles_h = 0;
    if (dtype == OMR_BF16) {
        const int rc = omr_wgrad_dma_bf16(a, s);            // asynchronous-staging kernel (conv_wgrad_dma.hip) where it covers the shape
        if (rc != OMR_ERR_UNSUPPORTED) return rc;
        return launch_wgrad<bf16, 8, 4>(a, s);
    }
    if (dtype == OMR_F32) return launch_wgrad<float, 4, 2>(a, s);
    return OMR_ERR_UNSUPPORTED;
}

extern "C" int omr_dwconv3x3(int dtype, const void* x, const void* w, const float* bias, void* y, const float* in_mean, const float* in_rstd,
                             const void* out_mask, float mask_scale, int B, int H, int W, int C, int flip, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return OMR_ERR_ARG;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (C % vec) return OMR_ERR_UNSUPPORTED;
    const int cv = C / vec;
    const size_t esz = dtype == OMR_BF16 ? 2 : 4;
    const int tc = cv <= 256 && 256 % cv == 0 ? 256 / cv : 0;
    const size_t tile_bytes = (size_t)(DW_TR + 2) * (tc + 2) * C * esz;
    if (tc >= 2 && H >= 4 && tile_bytes <= 64 * 1024 && (DW_TR + 2) * (tc + 2) * cv <= 16 * 256 && ((uintptr_t)w & 15) == 0) {      // LDS tile (DSC blocks)
        static std::atomic<int> opt_in{0};                           // large-LDS opt-in issued once per dtype (0 -> set only)
        const int bit = dtype == OMR_BF16 ? 1 : 2;
        if (tile_bytes > 32 * 1024 && !(opt_in.load(std::memory_order_acquire) & bit)) {
            const void* kern = dtype == OMR_BF16 ? (const void*)dwconv3x3_tile_kernel<bf16> : (const void*)dwconv3x3_tile_kernel<float>;
            if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return OMR_ERR_LAUNCH;
            opt_in.fetch_or(bit, std::memory_order_release);
        }
        dim3 gridt(cdiv(W, tc), B * cdiv(H, DW_TR));
        DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_tile_kernel<T>), gridt, 256, tile_bytes, (hipStream_t)stream, (const T*)x, (const T*)w, bias, (T*)y, in_mean, in_rstd,
                                             (const T*)out_mask, mask_scale, B, H, W, C, flip));
        OMR_CHECK_LAUNCH();
        return OMR_OK;
    }
    if (cv <= 256 && 256 % cv == 0 && H >= 4 && (size_t)10 * C * sizeof(float) <= 48 * 1024) {      // row walker (tiles that do not fit the LDS budget)
        int RC = 8;                                              // rows per thread: 2 halo rows re-read per RC
        while (RC < H && (long)cdiv(W, 256 / cv) * B * cdiv(H, RC) > 4096) RC *= 2;
        dim3 gridw(cdiv(W, 256 / cv), B * cdiv(H, RC));
        DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_walk_kernel<T>), gridw, 256, (size_t)9 * C * sizeof(T) + (size_t)C * sizeof(float), (hipStream_t)stream, (const T*)x, (const T*)w, bias,
                                             (T*)y, in_mean, in_rstd, (const T*)out_mask, mask_scale, B, H, W, C, flip, RC));
        OMR_CHECK_LAUNCH();
        return OMR_OK;
    }
    long total = (long)B * H * W * (C / vec);
    int grid = (int)((total + 1023) / 1024); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;   // >= 4 pixels x groups per thread amortise the weight staging
    DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_kernel<T>), grid, 256, (size_t)10 * C * sizeof(float), (hipStream_t)stream, (const T*)x, (const T*)w, bias, (T*)y,
                                         in_mean, in_rstd, (const T*)out_mask, mask_scale, B, H, W, C, flip));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

extern "C" int omr_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, const float* in_mean, const float* in_rstd,
                                   int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return OMR_ERR_ARG;
    const int vec = dtype == OMR_BF16 ? 8 : 4;
    if (C % vec || 256 % (C / vec)) return OMR_ERR_UNSUPPORTED;
    {
        const size_t esz = dtype == OMR_BF16 ? 2 : 4;
        const int cv = C / vec, tc = cv <= 256 && 256 % cv == 0 ? 256 / cv : 0;
        size_t tile_bytes = (size_t)(DW_TR + 2) * (tc + 2) * C * esz;
        const size_t red_bytes = (size_t)(cv >= 64 ? tc : 4) * cv * 10 * vec * sizeof(float);
        if (red_bytes > tile_bytes) tile_bytes = red_bytes;
        if ((cv == 16 || cv == 32 || (cv >= 64 && cv <= 256)) && tc >= 1 && H >= 4 && tile_bytes <= 64 * 1024 && (DW_TR + 2) * (tc + 2) * cv <= 16 * 256) {
            static std::atomic<int> opt_in{0};
            const int bit = dtype == OMR_BF16 ? 1 : 2;
            if (tile_bytes > 32 * 1024 && !(opt_in.load(std::memory_order_acquire) & bit)) {
                const void* k12 = dtype == OMR_BF16 ? (const void*)dwconv3x3_wgrad_tile_kernel<bf16, 12> : (const void*)dwconv3x3_wgrad_tile_kernel<float, 12>;
                const void* k16 = dtype == OMR_BF16 ? (const void*)dwconv3x3_wgrad_tile_kernel<bf16, 16> : (const void*)dwconv3x3_wgrad_tile_kernel<float, 16>;
                if (hipFuncSetAttribute(k12, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(k16, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return OMR_ERR_LAUNCH;
                opt_in.fetch_or(bit, std::memory_order_release);
            }
            const long ntiles = (long)cdiv(W, tc) * B * cdiv(H, DW_TR);
            dim3 gridt((unsigned)(ntiles < 256 ? ntiles : 256));       // persistent: one workgroup per CU (the closing atomics are per workgroup)
            if ((DW_TR + 2) * (tc + 2) * cv <= 12 * 256) {
                DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_wgrad_tile_kernel<T, 12>), gridt, 256, tile_bytes, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, db, in_mean,
                                                     in_rstd, B, H, W, C));
            } else {
                DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_wgrad_tile_kernel<T, 16>), gridt, 256, tile_bytes, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, db, in_mean,
                                                     in_rstd, B, H, W, C));
            }
            OMR_CHECK_LAUNCH();
            return OMR_OK;
        }
    }
    const int ncg = C / vec;
    if (ncg > 64 || 64 % ncg) return OMR_ERR_UNSUPPORTED;          // a wave holds whole channel-group sets
    dim3 grid(cdiv(W, 256 / ncg), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<T>), grid, 256, (size_t)C * 10 * sizeof(float), (hipStream_t)stream,
                                         (const T*)x, (const T*)dy, dw, db, in_mean, in_rstd, B, H, W, C));
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
