// Backward of a stride-1 3x3 convolution with <= 32 channels on either side, bf16, in ONE pass over its operands
// (aten::convolution_backward of nn.Conv2d, reference encoder.py:132-150, for the full-resolution layers of the first two
// ConvBlocks -- where the bytes of the encoder are):
//     dX[q][ci]       = sum_tap sum_co G[q + tap - 1][co] * Wflip[ci][tap][co]      * (X[q][ci] > 0) * mask_scale
//     dW[co][8-tap][ci] += sum_q   G[q + tap - 1][co] * X[q][ci]
//     db[co]          += sum_q    G[q][co]
// Why one kernel: these layers are HBM-bound (16 / 32 channels: 37-75 FLOP per byte), the data gradient already reads the G
// halo tile AND the X tile of the same pixels (its ReLU mask), which are exactly the operands of the weight gradient of the
// same tile -- summed over the X pixels q the tile owns, with G taken from the halo, every (q, tap) pair of the image is met
// exactly once.  The separate weight-gradient kernel read both tensors a second time (5.9 GB per C2 step).
//
// One 16-wave workgroup per CU, persistent over the 8 x 32 tiles of ONE image (grid z = image):
//   * operand tiles travel global -> LDS asynchronously (global_load_lds_dwordx4, dma_common.h) into a ring of 2-6 slots: the
//     next one to five tiles (G halo 10 x 34 pixels + X tile each; 53-97 KB) are in flight behind the current tile's math, no staging
//     registers; out-of-image pixels are fetched from a line of zeros.  Both halves of the workgroup issue DMA (waves 8-15 the G halo,
//     waves 0-7 the X tile / Y halo) and every wave's count of vector-memory operations per tile is a compile-time constant -- the
//     data-gradient lanes of pixels outside the image store to a sink -- so "my pieces of this tile have landed" is an exact s_waitcnt.
//   * LDS tiles are dense pixel-major rows with the 16-byte chunk index XORed by a function of the pixel's COLUMN (the DMA
//     permutes the source chunk instead of the destination): conflict-free for the data gradient's 16-byte row reads (16
//     consecutive pixels, one chunk) and for the transposing ds_read_b64_tr_b16 of the weight gradient; because the swizzle
//     depends on the column only, every address in the MFMA loops is a per-lane base + a compile-time offset;
//   * data gradient (waves 0-7): wave w owns tile row w: D[ci][pixel], 9 taps x (CO / 16) k-steps; a lane owns a pixel and stores
//     its masked channels straight from the accumulators;
//   * weight gradient (waves 8-15): K = the tile's 256 pixels in 16 slabs of 16; wave w owns tap w for all slabs plus a quarter
//     of the slabs of tap 8 (waves 0-3) or of the bias gradient = centre tap against a fragment of ones (waves 4-7): 20 MFMAs per
//     wave and tile; its accumulators hold sums nobody else has and leave by one atomic per element per workgroup at the end;
//   * APPLY: G is not stored anywhere -- it is the InstanceNorm backward of the layer above,
//     G = (Y > 0) * relu_scale * rstd * (Ghat - mean(Ghat) - yhat * mean(Ghat * yhat)),   yhat = (Y - mean) * rstd,
//     computed in LDS from the Ghat and Y halo tiles the ring delivers (an in-place pass before the MFMAs): the stand-alone
//     apply pass (norm.hip), its output and both re-reads of it go.
//   * XN (16 channels): the conv normalises its input on load (ConvBlock conv3): the X tile is normalised in LDS -- xhat is the weight
//     gradient's operand -- and the data gradient dL/dxhat leaves with {sum dx, sum dx * xhat} reduced deterministically into the
//     InstanceNorm-backward slots (what conv3x3_mfma.h's stat_mode 2 epilogue does for the separate data-gradient kernel).
#include <atomic>
#include <type_traits>
#include "omr_common.h"
#include "omr_hip.h"
#include "dma_common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 LdsV4;
constexpr int TW = 32, TH = 8, IH = TH + 2, IW = TW + 2, NHALO = IH * IW, NCORE = TH * TW;
constexpr int NDG = 512, NWG = 512;           // threads of the data-gradient / weight-gradient halves of the workgroup
constexpr int NLD = NWG;                      // the weight-gradient waves also issue the DMA

__device__ uint4 g_zero_line;                 // 16 zero bytes: the DMA source of every out-of-image chunk
__device__ uint4 g_store_sink[4];             // where the data-gradient lanes of pixels outside the image store (every lane always stores: exact vmcnt)

template <int CB> struct Tile {               // dense pixel-major LDS tile with CB channels; all offsets in bytes
    static constexpr int CPP = CB / 8, PITCH = CB * 2;
    __device__ static __forceinline__ int swz(int col) { return CB == 32 ? (col >> 2) & 3 : (col >> 3) & 1; }
    __device__ static __forceinline__ int chunk(int pix, int col, int c) { return pix * PITCH + ((c ^ swz(col)) << 4); }
    // address a lane hands to ds_read_b64_tr_b16 for pixel (pix, col), channels 16 cb + 4 p .. + 3.  16-channel tiles have no
    // channels 16..31: their cb = 1 lanes repeat the cb = 0 address (a broadcast), feeding rows / columns >= 16 of the product
    // with a copy nobody stores
    __device__ static __forceinline__ int tr(int pix, int col, int cb, int p) {
        return pix * PITCH + ((((CB == 32 ? 2 * cb : 0) + (p >> 1)) ^ swz(col)) << 4) + ((p & 1) << 3);
    }
};

// DMA descriptor of one operand tile (NPIXT pixels in rows of IWT): LDS chunk slot L holds channel chunk (L % CPP) ^ swz(column) of
// pixel L / CPP.  Per round a lane keeps the element offset of its source chunk from the tile-origin pixel and its (row, column);
// every issuing wave executes exactly ROUNDS instructions with all lanes active (lanes past the tile fetch zeros, waves whose 64
// chunks lie wholly past it write a scratch KB), so the waits can count.
template <int CB, int IWT, int NPIXT> struct Issuer {
    static constexpr int CPP = CB / 8, NCH = NPIXT * CPP, WINS = (NCH + 63) / 64, ROUNDS = (WINS * 64 + NLD - 1) / NLD, BYTES = WINS * 1024;
    int rel[ROUNDS], ij[ROUNDS];
    __device__ __forceinline__ void init(int Wimg, int wtid) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int L = r * NLD + wtid, pix = L / CPP, il = pix / IWT, jl = pix - il * IWT;
            const int c = (L % CPP) ^ Tile<CB>::swz(jl);
            ij[r] = il | (jl << 16);
            rel[r] = L < NCH ? (il * Wimg + jl) * CB + c * 8 : -1;
        }
    }
    // Interior tiles (tile-uniform, the common case) take a straight-line path: one 64-bit add and one select per round -- the issue
    // code runs on waves that have MFMAs waiting, every instruction of it is on the tile's critical path.
    __device__ __forceinline__ void issue(const bf16* origin, int y0, int x0, int Himg, int Wimg, bool interior, unsigned lds_base, unsigned scratch,
                                          int wtid) const {
        const bf16* zsrc = reinterpret_cast<const bf16*>(&g_zero_line);
        if (interior) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const bf16* src = rel[r] >= 0 ? origin + rel[r] : zsrc;
                const int L0 = r * NLD + (wtid & ~63);
                dma16(src, L0 < WINS * 64 ? lds_base + (unsigned)L0 * 16u : scratch);
            }
        } else {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int ih = y0 + (ij[r] & 0xffff), iw = x0 + (ij[r] >> 16);
                const bool ok = rel[r] >= 0 && (unsigned)ih < (unsigned)Himg && (unsigned)iw < (unsigned)Wimg;
                const bf16* src = ok ? origin + rel[r] : zsrc;
                const int L0 = r * NLD + (wtid & ~63);
                dma16(src, L0 < WINS * 64 ? lds_base + (unsigned)L0 * 16u : scratch);
            }
        }
    }
};

struct FusedArgs {
    const bf16* g; const bf16* x; const bf16* w; bf16* dx; float* dw; float* db;
    const bf16* ny; const float* mean; const float* rstd; const double* sums; float inv_hw; float relu_scale;
    const float* xmean; const float* xrstd; double* stat_ws; int stat_slots;      // XN: normalise X in LDS, reduce {sum dx, sum dx * xhat}
    int B, H, W, tiles_w, tiles_h, mask; float mask_scale;
#ifdef OMR_FUSED_DEBUG
    int dbg = 0;      // bring-up ablations (OMR_FUSED_DBG=bits): 1 no data-gradient MFMAs, 2 no weight-gradient MFMAs, 4 no stores, 8 no DMA
#endif
};
#ifdef OMR_FUSED_DEBUG
#include <cstdlib>
#define DBG(bit) (a.dbg & (bit))
// phase timing of block (0, 0, 0): wave 0 of each half adds the cycles since its previous mark to g_prof[half][mark]
__device__ unsigned long long g_prof[2][8];
#define PROF(k)                                                                                                  \
    do {                                                                                                         \
        if (blockIdx.x == 0 && blockIdx.z == 0 && wave == 0) {                                                  \
            const unsigned long long t__ = __builtin_readcyclecounter();                                         \
            if (lane == 0) atomicAdd(&g_prof[DG ? 0 : 1][k], t__ - prof_t);                                      \
            prof_t = t__;                                                                                        \
        }                                                                                                        \
    } while (0)
#else
#define DBG(bit) false
#define PROF(k) do {} while (0)
#endif

// NSLOT = ring depth: NSLOT - 1 tiles are in flight behind the one being consumed (what hides the loaded HBM latency is bytes in
// flight per CU: 60-100 KB here)
template <int CO, int CI, bool APPLY, int NSLOT> struct Lds {
    typedef Issuer<CO, IW, NHALO> IG;
    typedef Issuer<CI, TW, NCORE> IX;
    // DMA instructions per wave and tile: the weight-gradient waves fetch the G halo, the data-gradient waves the X tile (and the Y halo)
    static constexpr int DPT_WG = IG::ROUNDS, DPT_DG = IX::ROUNDS + (APPLY ? IG::ROUNDS : 0);
    static constexpr int WP = (CO + 8) * 2;                                           // weight row pitch
    static constexpr int GB = IG::BYTES, XB = IX::BYTES, SLOT = GB * (APPLY ? 2 : 1) + XB;
    static constexpr int WS = 32 * 9 * WP;
    static constexpr int OFF_WS = NSLOT * SLOT, OFF_SCRATCH = OFF_WS + WS, TOTAL = OFF_SCRATCH + 1024;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int off0, int off1, int imm) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(base + off0 + imm));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(base + off1 + imm));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// One tile loop, compiled twice: DG = the data-gradient waves (0-7), !DG = the weight-gradient / DMA waves (8-15).  Both meet at the
// same barriers; each keeps only its own accumulators in registers.
// Measured alternatives (B = 32, 256 x 2048, 32 -> 32 channels; this form: 1 076 us plain, 1 172-1 212 us with the InstanceNorm apply):
//   * four separate loader waves (a wave that issues DMA into a busy memory pipe stalls ~1 900 cycles per tile in the issue itself, on
//     these waves in front of their MFMAs) beside 8 + 4 or 4 + 8 compute waves: 1 118-1 191 us / 1 454-1 616 us -- the MFMA phase is
//     bound by LDS bandwidth (583 KB of operand reads per tile), concurrent DMA writes stretch it by what the issue stall had cost;
//   * issuing the DMA behind the MFMAs, beside the store loop: 1 097 / 1 288 us;
//   * register-staged tiles, two 8-wave workgroups per CU (the conv3x3_mfma.h pipeline): 1 042 us plain, but the apply needs the
//     second tensor in staging registers too and spills at 128 registers.
template <int CO, int CI, bool APPLY, bool XN, int NSLOT, bool DG>
__device__ __forceinline__ void tile_loop(const FusedArgs& a, unsigned char* smem, const float* cst) {
    typedef Tile<CO> GT;
    typedef Tile<CI> XT;
    typedef Lds<CO, CI, APPLY, NSLOT> L;
    constexpr int XCPP = CI / 8, KC = CO / 16, WP = L::WP;
    constexpr bool WG = !DG;
    // vector-memory operations a wave may still have in flight when its pieces of the CURRENT tile must have landed: the younger tiles'
    // pieces and, on the data-gradient waves, the XCPP stores of each of the NSLOT - 1 tiles finished since this tile was requested
    // (every lane always stores -- pixels outside the image go to a sink --, so the count is exact)
    constexpr int VM_ALLOWED = DG ? (NSLOT - 2) * L::DPT_DG + (NSLOT - 1) * XCPP : (NSLOT - 2) * L::DPT_WG;
    static_assert(NSLOT >= 2 && VM_ALLOWED < 64, "ring depth / vmcnt range");
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;         // index inside the half; wave-uniform and the compiler knows it
    const int wtid = tid & 511;
    const int frow = lane & 31, h = lane >> 5;
    const unsigned char* Ws = smem + L::OFF_WS;
    const unsigned lds0 = lds_address(smem);

    const bf16* G = a.g + (long)b * a.H * a.W * CO;
    const bf16* NY = APPLY ? a.ny + (long)b * a.H * a.W * CO : nullptr;
    const bf16* X = a.x + (long)b * a.H * a.W * CI;
    bf16* DX = a.dx + (long)b * a.H * a.W * CI;

    // ---- DMA descriptors (the Y halo shares the G halo's)
    typename L::IG ig;
    typename L::IX ix;
    if constexpr (WG || APPLY) ig.init(a.W, wtid);
    if constexpr (DG) ix.init(a.W, wtid);
    const int tiles_per_img = a.tiles_h * a.tiles_w;
    const int step_h = (int)gridDim.x / a.tiles_w, step_w = (int)gridDim.x - step_h * a.tiles_w;      // tile (th, tw) -> the block's next tile
    auto advance = [&](int& th, int& tw) {
        th += step_h; tw += step_w;
        if (tw >= a.tiles_w) { tw -= a.tiles_w; ++th; }
    };
    auto issue = [&](int th, int tw, int slot) {       // th >= tiles_h: a dummy (zeros) that keeps the in-flight count exact
        if (DBG(8)) return;
        const bool live = th < a.tiles_h;
        const int oh0 = live ? th * TH : -(1 << 20), ow0 = tw * TW;
        const bool in_g = live && oh0 >= 1 && oh0 + TH + 1 <= a.H && ow0 >= 1 && ow0 + TW + 1 <= a.W;
        const bool in_x = live && oh0 + TH <= a.H && ow0 + TW <= a.W;
        const unsigned base = lds0 + (unsigned)(slot * L::SLOT), scratch = lds0 + (unsigned)L::OFF_SCRATCH;
        const long og = ((long)(oh0 - 1) * a.W + (ow0 - 1)) * CO, ox = ((long)oh0 * a.W + ow0) * CI;
        if constexpr (WG) ig.issue(G + og, oh0 - 1, ow0 - 1, a.H, a.W, in_g, base, scratch, wtid);
        if constexpr (DG && APPLY) ig.issue(NY + og, oh0 - 1, ow0 - 1, a.H, a.W, in_g, base + (unsigned)L::GB, scratch, wtid);
        if constexpr (DG) ix.issue(X + ox, oh0, ow0, a.H, a.W, in_x, base + (unsigned)(L::GB * (APPLY ? 2 : 1)), scratch, wtid);
    };
    bf16* const sink = reinterpret_cast<bf16*>(g_store_sink) + 4 * (lane >> 5);
    auto dummy_stores = [&]() {                  // XCPP stores to the sink: keeps the data-gradient waves' in-flight count in its steady state
        if constexpr (DG) {
            typedef __attribute__((ext_vector_type(4))) bf16 B4;
            const B4 z = {0, 0, 0, 0};
#pragma unroll
            for (int g4 = 0; g4 < XCPP; ++g4) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(sink), "v"(z) : "memory");
        }
    };

    // ---- APPLY: this thread's channel chunk of the in-LDS pass is fixed (1024 % CPP == 0): its constants live in registers
    constexpr int GCPP = CO / 8, NGCH = NHALO * GCPP, RT = (NGCH + 1023) / 1024;
    float cA[APPLY ? 8 : 1], cB[APPLY ? 8 : 1], cC[APPLY ? 8 : 1];
    if constexpr (APPLY) {
        const int k8 = (tid % GCPP) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { cA[e] = cst[k8 + e]; cB[e] = cst[32 + k8 + e]; cC[e] = cst[64 + k8 + e]; }
    }

    // ---- per-lane LDS offsets of the MFMA operands (everything the loops add is a compile-time constant)
    int doff[3];                                             // DG: G fragment of tap column kw: pixel (wave + .., frow + kw), chunk h
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) doff[kw] = GT::chunk(wave * IW + frow + kw, frow + kw, h);
    const int woff = frow * 9 * WP + h * 16;                 // DG: weight fragment, row ci = frow, k half h
    const int q = (lane & 15) >> 2, p = lane & 3, cb = (lane >> 4) & 1;
    // WG: wave w owns tap w for all pixel slabs (16 of 16 pixels; PAIR: 8 of 32); its second accumulator takes a quarter of the slabs of tap 8 (waves 0-3) or of the
    // bias sums = centre tap against ones (waves 4-7): 20 MFMAs per wave and tile (the data-gradient waves: 18).  (Two taps x half the
    // slabs per wave -- one X fragment for both, 448 transposing reads per tile instead of 576 -- needs a third accumulator: spills.)
    const bool third_bias = wave >= 4;
    const bool third = !third_bias || a.db != nullptr;
    const int third_q = wave & 3;
    // 16 x 16 channels (PAIR): the cb = 1 half of the wave would only feed rows / columns >= 16 of the 32 x 32 product, which nobody wants.
    // It takes the NEXT 16 pixels of the tile row instead, same 16 channels: the product becomes block diagonal -- D[0:16][0:16] sums
    // pixels k0 .. k0+15, D[16:32][16:32] sums k0+16 .. k0+31 -- so one MFMA and one pair of LDS reads cover a whole tile row (half the
    // reads, half the MFMAs: conv_wgrad_dma.hip's trick); the flush adds the two diagonal blocks.
    constexpr bool PAIR = CO == 16 && CI == 16;
    constexpr int NSLAB = PAIR ? 8 : 16;                     // pixel slabs per tile
    int xo[2], go[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = 8 * h + q + 4 * u + (PAIR ? 16 * cb : 0);
        xo[u] = XT::tr(col, col, cb, p);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tap = j == 0 ? wave : (third_bias ? 4 : 8), dr = tap / 3, dc = tap - 3 * dr;
            go[j][u] = GT::tr(dr * IW + dc + col, dc + col, cb, p);
        }
    }
    f32x16 wacc[WG ? 2 : 1];
    if constexpr (WG) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[j][r] = 0.f;
    }
    const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};

    const float oscale = a.mask ? a.mask_scale : 1.f;
    constexpr int NSV = XN ? XCPP * 4 : 1;                  // XN, DG: partial sums of this lane's channels 8 g4 + 4 h + e
    float ssum[NSV], ssq[NSV];
#pragma unroll
    for (int e = 0; e < NSV; ++e) ssum[e] = ssq[e] = 0.f;
#ifdef OMR_FUSED_DEBUG
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif
    int th = (int)blockIdx.x / a.tiles_w, tw = (int)blockIdx.x - th * a.tiles_w;       // this block's first tile
    int ith = th, itw = tw;                                                          // the next tile to request
#pragma unroll
    for (int st = 0; st < NSLOT - 1; ++st) { issue(ith, itw, st); advance(ith, itw); dummy_stores(); }
    int cur = 0;

    for (; th < a.tiles_h; advance(th, tw), cur = cur + 1 == NSLOT ? 0 : cur + 1) {
        const int oh0 = th * TH, ow0 = tw * TW;
        unsigned char* Gs = smem + cur * L::SLOT;
        unsigned char* Ys = Gs + L::GB;
        unsigned char* Xt = Gs + L::GB * (APPLY ? 2 : 1);
        // this tile has landed (the DMA waves' only outstanding vector-memory operations are tile pieces, NSLOT - 2 younger tiles may
        // still be in flight); everybody is done with the slot consumed last (the previous tile's store loop)
        PROF(0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM_ALLOWED) : "memory");
        PROF(1);
        __syncthreads();
        PROF(2);
        // (issuing after the MFMAs instead, beside the other half's store loop, was measured: 1076 -> 1097 us, 1212 -> 1288 us with two slots)
        issue(ith, itw, cur == 0 ? NSLOT - 1 : cur - 1);
        advance(ith, itw);
        PROF(3);
        if constexpr (APPLY) {
            // G = cA * Ghat + cB * Y + cC where Y > 0, else 0 (out-of-image pixels arrive as Y = 0), in place over Ghat
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int c = tid + 1024 * r, pix = c / GCPP;
                if (c >= NGCH) continue;
                const int off = GT::chunk(pix, pix % IW, c % GCPP);
                bf16x8 v = *reinterpret_cast<const bf16x8*>(Gs + off);
                const bf16x8 y = *reinterpret_cast<const bf16x8*>(Ys + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float yf = (float)y[e];
                    const float d = fmaf(cA[e], (float)v[e], fmaf(cB[e], yf, cC[e]));
                    v[e] = (bf16)(yf > 0.f ? d : 0.f);
                }
                *reinterpret_cast<bf16x8*>(Gs + off) = v;
            }
        }
        if constexpr (XN) {
            // xhat = x * rstd - mean * rstd in place over the X tile (pixels of an overhanging tile stay 0)
            constexpr int NXCH = NCORE * XCPP;
            static_assert(NXCH <= 1024, "one chunk per thread");
            if (tid < NXCH) {
                const int pix = tid / XCPP, k = tid % XCPP;
                if (oh0 + (pix >> 5) < a.H && ow0 + (pix & 31) < a.W) {
                    unsigned char* px = Xt + XT::chunk(pix, pix & 31, k);
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(px);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaf((float)v[e], cst[k * 8 + e], cst[32 + k * 8 + e]);
                    *reinterpret_cast<bf16x8*>(px) = v;
                }
            }
        }
        if constexpr (APPLY || XN) __syncthreads();

        if constexpr (DG) {
            // ---- data gradient: D[ci][pixel] for tile row `wave`
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < (DBG(1) ? 0 : 9); ++tap) {
                const int kh = tap / 3, kw = tap % 3;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Ws + woff + tap * WP + kc * 32);
                    const bf16x8 gf = *reinterpret_cast<const bf16x8*>(Gs + (doff[kw] ^ (kc << 5)) + kh * IW * GT::PITCH);
                    mma32(acc, wf, gf);
                }
            }
            // accumulators -> global memory directly: a lane owns pixel (wave, frow) and holds four runs of 4 consecutive channels
            // (8 bytes); the four stores of a lane complete its pixel's 64-byte line within a few instructions, the ReLU mask comes from the
            // X tile in LDS.  (Through an LDS staging tile + 16-byte row stores by all data-gradient threads -- conv3x3_mfma.h's epilogue --
            // the tile cost one more workgroup barrier and ~1 100 cycles of serial store loop.)
            const int oh = oh0 + wave, ow = ow0 + frow;
            const bool inside = oh < a.H && ow < a.W;
            bf16* drow = inside ? DX + ((long)oh * a.W + ow) * CI + 4 * h : sink;
            const int dstep = inside ? 8 : 0;
            const unsigned char* xrow = Xt + 8 * h;
#pragma unroll
            for (int g4 = 0; g4 < XCPP; ++g4) {
                typedef __attribute__((ext_vector_type(4))) bf16 B4;
                typedef __attribute__((ext_vector_type(4))) short S4;
                B4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[4 * g4 + e] * oscale);
                if (a.mask) {
                    const S4 m = *reinterpret_cast<const S4*>(xrow + XT::chunk(wave * TW + frow, frow, g4));
                    const S4 keep = m > (short)0;
                    S4 bits;
                    __builtin_memcpy(&bits, &o, sizeof(bits));
                    bits &= keep;
                    __builtin_memcpy(&o, &bits, sizeof(bits));
                }
                // (asm: the store must be ONE instruction per run whatever the compiler thinks of the addresses -- the waits count them)
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(drow + dstep * g4), "v"(o) : "memory");
                if constexpr (XN) {      // InstanceNorm-backward sums over the stored values (conv3x3_mfma.h stat_mode 2)
                    const B4 xh = *reinterpret_cast<const B4*>(xrow + XT::chunk(wave * TW + frow, frow, g4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float f = inside ? (float)o[e] : 0.f; ssum[4 * g4 + e] += f; ssq[4 * g4 + e] += f * (float)xh[e]; }
                }
            }
        }
        if constexpr (WG) {
            // ---- weight gradient: D[co][ci] += G(tap)^T X over the 16 pixel slabs of the tile, four straight-line quarters (the second
            //      accumulator works in one of them: the wave-uniform test sits outside the unrolled MFMA code)
            auto slabs = [&](auto quarter_c, auto with_third) {
                constexpr int QUARTER = decltype(quarter_c)::value;
                constexpr bool T3 = decltype(with_third)::value;
#pragma unroll
                for (int s4 = 0; s4 < (DBG(2) ? 0 : NSLAB / 4); ++s4) {
                    const int s = QUARTER * (NSLAB / 4) + s4;
                    const int srow = PAIR ? s : s >> 1, scol = PAIR ? 0 : (s & 1) * 16;
                    const int ximm = (srow * TW + scol) * XT::PITCH, gimm = (srow * IW + scol) * GT::PITCH;
                    const bf16x8 xf = tr_frag(Xt, xo[0], xo[1], ximm);
                    mma32(wacc[0], tr_frag(Gs, go[0][0], go[0][1], gimm), xf);
                    if constexpr (T3) mma32(wacc[1], tr_frag(Gs, go[1][0], go[1][1], gimm), third_bias ? ones : xf);
                }
            };
            auto quarter = [&](auto qc) {
                if (third && third_q == decltype(qc)::value) slabs(qc, std::true_type{});
                else slabs(qc, std::false_type{});
            };
            quarter(std::integral_constant<int, 0>{});
            quarter(std::integral_constant<int, 1>{});
            quarter(std::integral_constant<int, 2>{});
            quarter(std::integral_constant<int, 3>{});
        }
        PROF(4);
        PROF(6);
    }

    dma_drain();        // the trailing dummy DMA must not outlive the workgroup's LDS
    if constexpr (XN) {
        // DETERMINISTIC reduction of the sums (conv3x3_mfma.h's protocol: fixed-order fp64 sum of the block's partials, plain store into
        // the block's own slot stat_ws[b][blockIdx.x][CI][2]; slots no block owns are zeroed by block 0)
        float* red = reinterpret_cast<float*>(smem);          // [2][CI][256]: the ring is dead
        __syncthreads();
        if constexpr (DG) {
            const int j = wave * 32 + frow;
#pragma unroll
            for (int g4 = 0; g4 < XCPP; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 8 * g4 + 4 * h + e;
                    red[c * 256 + j] = ssum[4 * g4 + e];
                    red[(CI + c) * 256 + j] = ssq[4 * g4 + e];
                }
        }
        __syncthreads();
        if (DG && tid < 2 * CI) {
            const int k = tid / CI, c = tid - k * CI;
            double acc2 = 0.0;
            for (int j = 0; j < 256; ++j) acc2 += (double)red[(k * CI + c) * 256 + j];
            a.stat_ws[(((long)b * a.stat_slots + blockIdx.x) * CI + c) * 2 + k] = acc2;
            if (blockIdx.x == 0)
                for (int sl = gridDim.x; sl < a.stat_slots; ++sl) a.stat_ws[(((long)b * a.stat_slots + sl) * CI + c) * 2 + k] = 0.0;
        }
    }
    if constexpr (WG) {
        // ---- the workgroup's weight-gradient sums: one atomic per element (tap 8 and the bias arrive in four quarters)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && third_bias) break;
            const int tap = j == 0 ? wave : 8, tw8 = 8 - tap;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = acc_row(r, lane), ci = lane & 31;
                if constexpr (PAIR) {
                    if ((co < 16) == (ci < 16)) atomicAdd(&a.dw[((long)(co & 15) * 9 + tw8) * CI + (ci & 15)], wacc[j][r]);
                } else if (co < CO && ci < CI) atomicAdd(&a.dw[((long)co * 9 + tw8) * CI + ci], wacc[j][r]);
            }
        }
        if (third_bias && a.db != nullptr && (lane & 31) == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = acc_row(r, lane);
                if (PAIR || co < CO) atomicAdd(&a.db[co & (PAIR ? 15 : 31)], wacc[1][r]);       // PAIR: rows 16-31 hold the second pixel half
            }
        }
    }
}

template <int CO, int CI, bool APPLY, bool XN, int NSLOT>
__global__ __launch_bounds__(1024) void conv_bwd_fused_kernel(FusedArgs a) {
    typedef Lds<CO, CI, APPLY, NSLOT> L;
    constexpr int GCPP = CO / 8, WP = L::WP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ __attribute__((aligned(16))) float cst[3 * 32];    // APPLY: G = cA * Ghat + cB * Y + cC per channel
    const int tid = threadIdx.x, b = blockIdx.z;
    unsigned char* Ws = smem + L::OFF_WS;                          // flipped weights [32 rows ci][9][CO], rows >= CI zero
    for (int c = tid; c < 32 * 9 * GCPP; c += 1024) {
        const int row = c / GCPP, kc = c - row * GCPP;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < CI * 9) v = *reinterpret_cast<const bf16x8*>(a.w + (long)row * CO + kc * 8);
        *reinterpret_cast<bf16x8*>(Ws + row * WP + kc * 16) = v;
    }
    if constexpr (APPLY) {
        if (tid < 32) {
            float ca = 0.f, cbv = 0.f, cc = 0.f;
            if (tid < CO) {
                const long bc = (long)b * CO + tid;
                const float mu = a.mean[bc], rs = a.rstd[bc];
                const float s1 = (float)(a.sums[2 * bc] * (double)a.inv_hw), s2 = (float)(a.sums[2 * bc + 1] * (double)a.inv_hw);
                ca = a.relu_scale * rs;
                cbv = -a.relu_scale * rs * rs * s2;
                cc = a.relu_scale * (-rs * s1 + mu * rs * rs * s2);
            }
            cst[tid] = ca; cst[32 + tid] = cbv; cst[64 + tid] = cc;
        }
    }
    if constexpr (XN) {
        static_assert(!APPLY, "one set of constants");
        if (tid < 32) {
            const float rs = tid < CI ? a.xrstd[(long)b * CI + tid] : 0.f, mu = tid < CI ? a.xmean[(long)b * CI + tid] : 0.f;
            cst[tid] = rs; cst[32 + tid] = -mu * rs;
        }
    }
    __syncthreads();
    if (tid < NDG) tile_loop<CO, CI, APPLY, XN, NSLOT, true>(a, smem, cst);
    else tile_loop<CO, CI, APPLY, XN, NSLOT, false>(a, smem, cst);
}

template <int CO, int CI, bool APPLY, bool XN, int NSLOT> int launch(FusedArgs a, hipStream_t s) {
    typedef Lds<CO, CI, APPLY, NSLOT> L;
    static_assert(L::TOTAL <= 160 * 1024, "LDS ring does not fit");
    a.tiles_w = cdiv(a.W, TW);
    a.tiles_h = cdiv(a.H, TH);
#ifdef OMR_FUSED_DEBUG
    { const char* e = getenv("OMR_FUSED_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    if ((long)(IH + 1) * a.W * 32 >= (1L << 30)) return OMR_ERR_UNSUPPORTED;          // 32-bit tile-relative offsets
    auto kern = conv_bwd_fused_kernel<CO, CI, APPLY, XN, NSLOT>;
    static std::atomic<int> ready{0};
    if (ready.load(std::memory_order_acquire) == 0) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL) != hipSuccess) return OMR_ERR_LAUNCH;
        ready.store(1, std::memory_order_release);
    }
    const long tiles_per_img = (long)a.tiles_w * a.tiles_h;
    long gx = (256 + a.B - 1) / a.B;                  // one workgroup per CU, split evenly over the images
    if (gx > tiles_per_img) gx = tiles_per_img;
    if (XN && gx > a.stat_slots) gx = a.stat_slots;          // one workspace slot per block of an image
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1, a.B), dim3(1024), L::TOTAL, s, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

template <bool APPLY> int pick(const FusedArgs& a, int CO, int CI, hipStream_t s) {
    // ring as deep as 160 KB of LDS allows
    if (CO == 32 && CI == 32) return launch<32, 32, APPLY, false, APPLY ? 2 : 3>(a, s);
    if (CO == 32 && CI == 16) return launch<32, 16, APPLY, false, APPLY ? 2 : 4>(a, s);
    if (CO == 16 && CI == 16) return launch<16, 16, APPLY, false, APPLY ? 4 : 6>(a, s);
    return OMR_ERR_UNSUPPORTED;
}


// ------------------------------------------------------------------------------------------------
// Stride (2, 2): ConvBlock 1's conv3 (32 -> 32 channels, InstanceNorm applied on load).  Same workgroup, ring and roles; what changes
// is the geometry.  The tile is 8 x 32 pixels of the FULL-resolution input X (= of the outgoing gradient dL/dxhat); the gradient G
// of the half-resolution output contributes through  q = 2 p + k - 1:  pixel q of X meets tap k = (kh, kw) iff q + 1 - k is even,
// at p = (q + 1 - k) / 2 -- rows 4t .. 4t + 4, columns 16 tw .. 16 tw + 16 of G: a 5 x 17 halo tile.
//   * data gradient: a wave owns one tile row, so the row parity fixes the tap rows (even rows: kh = 1; odd rows: kh = 0, 2) for the
//     whole wave; the column parity differs lane by lane, so a tap column's MFMA carries the G fragment only in the lanes of its parity
//     (zeros in the others: kw = 1 on even columns, kw = 0 / 2 on odd ones) -- 6 or 12 MFMAs per row instead of 18, no dilated zeros staged;
//   * weight gradient: K = the X pixels of ONE column parity of a row (16 pixels, read from LDS at pixel stride 2) against 16 consecutive
//     G pixels; tap (kh, kw) meets exactly four such slabs per tile (rows of its parity x its column parity): wave w owns tap w, tap 8
//     and the bias sums (the tile's own 4 x 16 G pixels against ones) are one slab per wave;
//   * X is normalised in LDS and the InstanceNorm-backward sums of the outgoing gradient are reduced as in the stride-1 XN mode.
struct S2 {
    static constexpr int GH = TH / 2 + 1, GW = TW / 2 + 1, NG = GH * GW;
    typedef Issuer<32, GW, NG> IG;
    typedef Issuer<32, TW, NCORE> IX;
    static constexpr int NSLOT = 4, GB = IG::BYTES, XB = IX::BYTES, SLOT = GB + XB, WP = 40 * 2, WS = 32 * 9 * WP;
    static constexpr int OFF_WS = NSLOT * SLOT, OFF_SCRATCH = OFF_WS + WS, TOTAL = OFF_SCRATCH + 1024;
    static constexpr int DPT_WG = IG::ROUNDS, DPT_DG = IX::ROUNDS;
};

template <bool DG>
__device__ __forceinline__ void tile_loop_s2(const FusedArgs& a, unsigned char* smem, const float* cst) {
    typedef Tile<32> T32;
    constexpr int NSLOT = S2::NSLOT, GW = S2::GW, WP = S2::WP, XCPP = 4;
    constexpr bool WG = !DG;
    constexpr int VM_ALLOWED = DG ? (NSLOT - 2) * S2::DPT_DG + (NSLOT - 1) * XCPP : (NSLOT - 2) * S2::DPT_WG;
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int wtid = tid & 511;
    const int frow = lane & 31, h = lane >> 5;
    const unsigned char* Ws = smem + S2::OFF_WS;
    const unsigned lds0 = lds_address(smem);
    const int Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;
    const bf16* G = a.g + (long)b * Ho * Wo * 32;
    const bf16* X = a.x + (long)b * a.H * a.W * 32;
    bf16* DX = a.dx + (long)b * a.H * a.W * 32;

    S2::IG ig;
    S2::IX ix;
    if constexpr (WG) ig.init(Wo, wtid);
    if constexpr (DG) ix.init(a.W, wtid);
    const int step_h = (int)gridDim.x / a.tiles_w, step_w = (int)gridDim.x - step_h * a.tiles_w;
    auto advance = [&](int& th, int& tw) {
        th += step_h; tw += step_w;
        if (tw >= a.tiles_w) { tw -= a.tiles_w; ++th; }
    };
    auto issue = [&](int th, int tw, int slot) {
        const bool live = th < a.tiles_h;
        const int oh0 = live ? th * TH : -(1 << 20), ow0 = tw * TW;
        const int ph0 = live ? th * (TH / 2) : -(1 << 20), pw0 = tw * (TW / 2);
        const unsigned base = lds0 + (unsigned)(slot * S2::SLOT), scratch = lds0 + (unsigned)S2::OFF_SCRATCH;
        if constexpr (WG) {
            const bool in_g = live && ph0 + S2::GH <= Ho && pw0 + GW <= Wo;
            ig.issue(G + ((long)ph0 * Wo + pw0) * 32, ph0, pw0, Ho, Wo, in_g, base, scratch, wtid);
        }
        if constexpr (DG) {
            const bool in_x = live && oh0 + TH <= a.H && ow0 + TW <= a.W;
            ix.issue(X + ((long)oh0 * a.W + ow0) * 32, oh0, ow0, a.H, a.W, in_x, base + (unsigned)S2::GB, scratch, wtid);
        }
    };
    bf16* const sink = reinterpret_cast<bf16*>(g_store_sink) + 4 * (lane >> 5);
    auto dummy_stores = [&]() {
        if constexpr (DG) {
            typedef __attribute__((ext_vector_type(4))) bf16 B4;
            const B4 z = {0, 0, 0, 0};
#pragma unroll
            for (int g4 = 0; g4 < XCPP; ++g4) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(sink), "v"(z) : "memory");
        }
    };

    // ---- data gradient: row `wave` of the tile; tap rows by the row's parity, tap columns by the lane's
    const bool row_odd = wave & 1;
    const int woff = frow * 9 * WP + h * 16;
    int dgo[3];                              // G fragment of tap column kw for THIS lane (valid where its column parity takes that tap), row 0 of the G tile
    bool dga[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int num = frow + 1 - kw;       // = 2 * (G column inside the tile) where the parity fits
        dga[kw] = (num & 1) == 0;
        const int pc = dga[kw] ? num >> 1 : 0;
        dgo[kw] = T32::chunk(pc, pc, h);
    }
    // ---- weight gradient: tap `wave` (kh, kw) over its four slabs; second accumulator: one slab of tap 8 (waves 0-3) / of the bias sums (waves 4-7)
    const int q = (lane & 15) >> 2, p = lane & 3, cb = (lane >> 4) & 1;
    const int wkh = wave / 3, wkw = wave - 3 * wkh;
    const bool third_bias = wave >= 4;
    const bool third = !third_bias || a.db != nullptr;
    // lane's K index j = 8 h + q (+4): X pixel column 2 j + cp, G pixel column j + dcol
    auto slab_offsets = [&](int kw, int (&xo)[2], int (&go)[2]) {
        const int cp = kw == 1 ? 0 : 1, dcol = (cp + 1 - kw) >> 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 8 * h + q + 4 * u;
            xo[u] = T32::tr(2 * j + cp, 2 * j + cp, cb, p);
            go[u] = T32::tr(dcol + j, dcol + j, cb, p);
        }
    };
    int xo1[2], go1[2], xo2[2], go2[2];
    slab_offsets(wkw, xo1, go1);
    slab_offsets(2, xo2, go2);               // tap 8 = (2, 2)
    int gb2[2];                              // bias: the tile's own G pixels, columns j = 0 .. 15 of a row
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int j = 8 * h + q + 4 * u; gb2[u] = T32::tr(j, j, cb, p); }
    f32x16 wacc[WG ? 2 : 1];
    if constexpr (WG) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[j][r] = 0.f;
    }
    const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float ssum[DG ? 16 : 1], ssq[DG ? 16 : 1];
#pragma unroll
    for (int e = 0; e < (DG ? 16 : 1); ++e) ssum[e] = ssq[e] = 0.f;

    int th = (int)blockIdx.x / a.tiles_w, tw = (int)blockIdx.x - th * a.tiles_w;
    int ith = th, itw = tw;
#pragma unroll
    for (int st = 0; st < NSLOT - 1; ++st) { issue(ith, itw, st); advance(ith, itw); dummy_stores(); }
    int cur = 0;
    for (; th < a.tiles_h; advance(th, tw), cur = (cur + 1) & (NSLOT - 1)) {
        const int oh0 = th * TH, ow0 = tw * TW;
        unsigned char* Gs = smem + cur * S2::SLOT;
        unsigned char* Xt = Gs + S2::GB;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM_ALLOWED) : "memory");
        __syncthreads();
        issue(ith, itw, (cur + NSLOT - 1) & (NSLOT - 1));
        advance(ith, itw);
        {   // xhat = x * rstd - mean * rstd in place (one 16-byte chunk per thread; pixels of an overhanging tile stay 0)
            const int pix = tid >> 2, k = tid & 3;
            if (oh0 + (pix >> 5) < a.H && ow0 + (pix & 31) < a.W) {
                unsigned char* px = Xt + T32::chunk(pix, pix & 31, k);
                bf16x8 v = *reinterpret_cast<const bf16x8*>(px);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaf((float)v[e], cst[k * 8 + e], cst[32 + k * 8 + e]);
                *reinterpret_cast<bf16x8*>(px) = v;
            }
        }
        __syncthreads();

        if constexpr (DG) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // G row inside the tile: (wave + 1 - kh) / 2
            auto taps = [&](auto kh_c) {
                constexpr int KH = decltype(kh_c)::value;
                const int prow = (wave + 1 - KH) >> 1;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tapf = 8 - (KH * 3 + kw);                  // index into the flipped weights
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Ws + woff + tapf * WP + kc * 32);
                        bf16x8 gf = *reinterpret_cast<const bf16x8*>(Gs + prow * GW * 64 + (dgo[kw] ^ (kc << 5)));
                        if (!dga[kw]) gf = zero8;
                        mma32(acc, wf, gf);
                    }
                }
            };
            if (row_odd) { taps(std::integral_constant<int, 0>{}); taps(std::integral_constant<int, 2>{}); }
            else taps(std::integral_constant<int, 1>{});
            const int oh = oh0 + wave, ow = ow0 + frow;
            const bool inside = oh < a.H && ow < a.W;
            bf16* drow = inside ? DX + ((long)oh * a.W + ow) * 32 + 4 * h : sink;
            const int dstep = inside ? 8 : 0;
            const unsigned char* xrow = Xt + 8 * h;
#pragma unroll
            for (int g4 = 0; g4 < XCPP; ++g4) {
                typedef __attribute__((ext_vector_type(4))) bf16 B4;
                B4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[4 * g4 + e];
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(drow + dstep * g4), "v"(o) : "memory");
                const B4 xh = *reinterpret_cast<const B4*>(xrow + T32::chunk(wave * TW + frow, frow, g4));
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float f = inside ? (float)o[e] : 0.f; ssum[4 * g4 + e] += f; ssq[4 * g4 + e] += f * (float)xh[e]; }
            }
        }
        if constexpr (WG) {
            // tap (wkh, wkw): rows of the parity that fits kh, G row (r + 1 - kh) / 2
            auto tap_slabs = [&](auto par_c) {
                constexpr int PAR = decltype(par_c)::value;                  // 0: even rows (kh = 1), 1: odd rows (kh = 0, 2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 2 * i + PAR;
                    const int prow = (r + 1 - wkh) >> 1;                     // wave-uniform
                    const bf16x8 xf = tr_frag(Xt, xo1[0], xo1[1], r * TW * 64);
                    mma32(wacc[0], tr_frag(Gs + prow * GW * 64, go1[0], go1[1], 0), xf);
                }
            };
            if (wkh == 1) tap_slabs(std::integral_constant<int, 0>{});
            else tap_slabs(std::integral_constant<int, 1>{});
            if (third) {
                if (third_bias) {        // bias: G row (wave - 4) of the tile's own four rows
                    mma32(wacc[1], tr_frag(Gs + (wave - 4) * GW * 64, gb2[0], gb2[1], 0), ones);
                } else {                 // tap 8 = (2, 2): odd rows r = 2 wave + 1, G row (r - 1) / 2 = wave
                    const int r = 2 * wave + 1;
                    const bf16x8 xf = tr_frag(Xt + r * TW * 64, xo2[0], xo2[1], 0);
                    mma32(wacc[1], tr_frag(Gs + wave * GW * 64, go2[0], go2[1], 0), xf);
                }
            }
        }
    }
    dma_drain();
    {   // deterministic reduction of the InstanceNorm-backward sums (see the stride-1 XN mode)
        float* red = reinterpret_cast<float*>(smem);          // [2][32][256]
        __syncthreads();
        if constexpr (DG) {
            const int j = wave * 32 + frow;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 8 * g4 + 4 * h + e;
                    red[c * 256 + j] = ssum[4 * g4 + e];
                    red[(32 + c) * 256 + j] = ssq[4 * g4 + e];
                }
        }
        __syncthreads();
        if (DG && tid < 64) {
            const int k = tid >> 5, c = tid & 31;
            double acc2 = 0.0;
            for (int j = 0; j < 256; ++j) acc2 += (double)red[(k * 32 + c) * 256 + j];
            a.stat_ws[(((long)b * a.stat_slots + blockIdx.x) * 32 + c) * 2 + k] = acc2;
            if (blockIdx.x == 0)
                for (int sl = gridDim.x; sl < a.stat_slots; ++sl) a.stat_ws[(((long)b * a.stat_slots + sl) * 32 + c) * 2 + k] = 0.0;
        }
    }
    if constexpr (WG) {
        const int tapw = third_bias ? -1 : 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && (third_bias || !third)) break;
            const int tap = j == 0 ? wave : tapw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = acc_row(r, lane), ci = lane & 31;
                atomicAdd(&a.dw[((long)co * 9 + tap) * 32 + ci], wacc[j][r]);
            }
        }
        if (third_bias && a.db != nullptr && (lane & 31) == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(&a.db[acc_row(r, lane)], wacc[1][r]);
        }
    }
}

__global__ __launch_bounds__(1024) void conv_bwd_fused_s2_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ __attribute__((aligned(16))) float cst[2 * 32];
    const int tid = threadIdx.x, b = blockIdx.z;
    unsigned char* Ws = smem + S2::OFF_WS;
    for (int c = tid; c < 32 * 9 * 4; c += 1024) {
        const int row = c >> 2, kc = c & 3;
        *reinterpret_cast<bf16x8*>(Ws + row * S2::WP + kc * 16) = *reinterpret_cast<const bf16x8*>(a.w + (long)row * 32 + kc * 8);
    }
    if (tid < 32) {
        const float rs = a.xrstd[(long)b * 32 + tid], mu = a.xmean[(long)b * 32 + tid];
        cst[tid] = rs; cst[32 + tid] = -mu * rs;
    }
    __syncthreads();
    if (tid < NDG) tile_loop_s2<true>(a, smem, cst);
    else tile_loop_s2<false>(a, smem, cst);
}

}  // namespace

extern "C" int omr_conv3x3_bwd_fused(const void* g, const void* x, const void* w_flipped, void* dx, float* dw, float* db, int B, int H, int W, int CIN,
                                     int COUT, int mask_input, float mask_scale, const void* norm_y, const float* norm_mean, const float* norm_rstd,
                                     const void* norm_workspace, int norm_slots, int relu_mask, float relu_scale, const float* x_mean,
                                     const float* x_rstd, void* stat_workspace, int stat_slots, void* stream) {
    if (!g || !x || !w_flipped || !dx || !dw || B <= 0 || H <= 0 || W <= 0) return OMR_ERR_ARG;
    if ((x_mean != nullptr) != (x_rstd != nullptr) || (x_mean != nullptr) != (stat_workspace != nullptr)) return OMR_ERR_ARG;
    if ((((uintptr_t)g) | ((uintptr_t)x) | ((uintptr_t)norm_y)) & 15) return OMR_ERR_UNSUPPORTED;       // tiles are fetched in 16-byte pieces
    FusedArgs a{};
    a.g = (const bf16*)g; a.x = (const bf16*)x; a.w = (const bf16*)w_flipped; a.dx = (bf16*)dx; a.dw = dw; a.db = db;
    a.B = B; a.H = H; a.W = W; a.mask = mask_input; a.mask_scale = mask_scale;
    if (x_mean) {
        // the conv normalises its input on load (ConvBlock conv3, stride 1): xhat feeds the weight gradient, and the data gradient
        // dL/dxhat leaves with the InstanceNorm-backward sums reduced into the slots (omr_conv3x3_fwd stat_mode 2's protocol)
        if (norm_y || mask_input || stat_slots < 1) return OMR_ERR_ARG;
        if (COUT != 16 || CIN != 16) return OMR_ERR_UNSUPPORTED;
        a.xmean = x_mean; a.xrstd = x_rstd; a.stat_ws = (double*)stat_workspace; a.stat_slots = stat_slots;
        return launch<16, 16, false, true, 6>(a, (hipStream_t)stream);
    }
    if (norm_y) {
        if (!norm_mean || !norm_rstd || !norm_workspace || norm_slots < 1) return OMR_ERR_ARG;
        if (!relu_mask) return OMR_ERR_UNSUPPORTED;       // the in-LDS pass relies on (Y > 0) to keep out-of-image pixels zero
        a.ny = (const bf16*)norm_y; a.mean = norm_mean; a.rstd = norm_rstd;
        a.sums = (const double*)norm_workspace + (long)B * norm_slots * COUT * 2;      // the compact [B][COUT][2] sums behind the slots
        a.inv_hw = (float)(1.0 / ((double)H * W)); a.relu_scale = relu_scale;
        return pick<true>(a, COUT, CIN, (hipStream_t)stream);
    }
    return pick<false>(a, COUT, CIN, (hipStream_t)stream);
}

#ifdef OMR_FUSED_DEBUG
extern "C" int omr_fused_prof_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16) != hipSuccess) return OMR_ERR_LAUNCH;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return OMR_ERR_LAUNCH; }
    return OMR_OK;
}
#endif

extern "C" int omr_conv3x3_bwd_fused_s2(const void* g, const void* x, const void* w_flipped, void* dx, float* dw, float* db, int B, int H, int W,
                                        const float* x_mean, const float* x_rstd, void* stat_workspace, int stat_slots, void* stream) {
    if (!g || !x || !w_flipped || !dx || !dw || !x_mean || !x_rstd || !stat_workspace || B <= 0 || H <= 0 || W <= 0 || stat_slots < 1) return OMR_ERR_ARG;
    if ((((uintptr_t)g) | ((uintptr_t)x)) & 15) return OMR_ERR_UNSUPPORTED;
    if ((long)(IH + 1) * W * 32 >= (1L << 30)) return OMR_ERR_UNSUPPORTED;
    FusedArgs a{};
    a.g = (const bf16*)g; a.x = (const bf16*)x; a.w = (const bf16*)w_flipped; a.dx = (bf16*)dx; a.dw = dw; a.db = db;
    a.B = B; a.H = H; a.W = W; a.xmean = x_mean; a.xrstd = x_rstd; a.stat_ws = (double*)stat_workspace; a.stat_slots = stat_slots;
    a.tiles_w = cdiv(W, TW); a.tiles_h = cdiv(H, TH);
    static std::atomic<int> ready{0};
    if (ready.load(std::memory_order_acquire) == 0) {
        if (hipFuncSetAttribute((const void*)conv_bwd_fused_s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S2::TOTAL) != hipSuccess) return OMR_ERR_LAUNCH;
        ready.store(1, std::memory_order_release);
    }
    const long tiles_per_img = (long)a.tiles_w * a.tiles_h;
    long gx = (256 + B - 1) / B;
    if (gx > tiles_per_img) gx = tiles_per_img;
    if (gx > stat_slots) gx = stat_slots;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(conv_bwd_fused_s2_kernel, dim3((unsigned)gx, 1, B), dim3(1024), S2::TOTAL, (hipStream_t)stream, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}
