// bf16 weight gradient of the 3x3 convolutions (nn.Conv2d backward-weight, reference encoder.py:132-150) for gfx950:
//   dW[n][tap][c] += sum_pix dY[pix][n] * X[pix*stride + tap][c]          (K = output pixels, fp32 atomics into the flat grad)
//
// What bounds it: HBM for <= 32 channels, LDS read bandwidth / MFMA issue above.  The generic kernel in conv.hip stages
// tiles through registers (load -> wait -> ds_write -> barrier -> MFMA), so every tile pays the full HBM latency with
// nothing else in flight.  Here the operand tiles travel global -> LDS asynchronously (global_load_lds_dwordx4: the
// wave's 64 lanes deposit 64 consecutive 16-byte chunks at M0; semantics checked on hardware, scratch/dmatest.hip)
// into a double-buffered LDS ring: the DMA of tile t+1 is in flight while the MFMAs consume tile t, no staging
// registers, one barrier per tile.
//   * LDS tiles are dense [pixel][CB] rows (CB = 16 / 32 / 64 channels).  Because each lane chooses its own global
//     source, bank conflicts are removed by permuting the SOURCE chunk (XOR swizzle on 128-byte rows) instead of padding;
//     16-channel layers use 32-byte rows and leave the upper half of the MFMA operand zero in registers.
//   * k-major MFMA operands come straight out of the pixel-major tiles with ds_read_b64_tr_b16.
//   * Out-of-image halo pixels / channel tails are zero-filled by ordinary ds_writes from the lane that would have
//     fetched them.
//   * Fused: bias gradient (column sums of the staged dY tile) and, for the strided convs that normalise on load
//     (ConvBlock conv3, encoder.py:151-156), the InstanceNorm-apply as an in-place LDS pass over the landed X tile.
#include <atomic>
#include "omr_common.h"
#include "omr_hip.h"

#include "conv_wgrad.h"
#include "dma_common.h"

// Ablation switches for bring-up (make CXXFLAGS+=-DOMR_WGRAD_DEBUG; OMR_WGRAD_DBG=bits at run time): 1 dummy sources,
// 2 no MFMA loop, 4 no bias sums, 8 no DMA, 16 no barrier, 32 no final accumulation.  Never compiled into the product.
#ifdef OMR_WGRAD_DEBUG
#include <cstdlib>
#define DBG(bit) (a.dbg & (bit))
#else
#define DBG(bit) false
#endif

namespace {

constexpr int TW = 32;   // output pixels per tile row = two MFMA k-steps
constexpr int NUM_CU = 256;
typedef __attribute__((address_space(3))) bf16x4 LdsV4;

__device__ uint4 g_zero16;   // 16 zero bytes in global memory: the DMA source of every out-of-image / out-of-range chunk

// Placement of 16-byte chunk c of tile pixel `pix` inside a dense [pixel][CB] LDS tile.  Each lane of the DMA chooses its
// own global source, so bank conflicts are removed by permuting where a chunk lands instead of padding rows:
//   * the 4 pixel rows x 64 bytes that half a wave touches in one transposing read must cover all 64 banks (256 B);
//   * unit pixel step (dY tiles, X tiles of stride-1 convs): rows narrower than 128 B are conflict-free as they lie;
//     128-byte rows swap their 64-byte halves on every other pixel pair;
//   * pixel step 2 (X tiles of the strided convs): the four rows share their parity, so additionally neighbouring pixels
//     swap places on every other group of four.
// PS = pixel step of the reads (1 or 2).  Both maps are involutions on (pix, c), so the DMA uses the same function to
// find the source of an LDS slot.
template <int CB, int PS> struct Place {
    static constexpr int CPP = CB / 8;
    // 32-byte rows (CB = 16): the two halves of a wave read pixels 16 apart (paired k-steps, see the kernel) -- 512 bytes, the
    // same banks -- so every other group of 16 pixels has its aligned pixel quads swapped in pairs (+-128 bytes)
    __device__ static __forceinline__ int pixel(int pix) {
        return CB == 16 ? pix ^ (((pix >> 4) & 1) << 2) : (PS == 2 && CB >= 32) ? pix ^ ((pix >> 2) & 1) : pix;
    }
    __device__ static __forceinline__ int chunk(int pix, int c) { return CB == 64 ? c ^ (((pix >> 1) & 1) << 2) : c; }
    // element offset of channel ch of pixel pix
    __device__ static __forceinline__ int off(int pix, int ch) { return pixel(pix) * CB + ((chunk(pix, ch >> 3) << 3) | (ch & 7)); }
    // LDS chunk slot L -> (pix, c) stored there
    __device__ static __forceinline__ void source(int L, int& pix, int& c) {
        pix = pixel(L / CPP);
        c = chunk(pix, L % CPP);
    }
};

template <int CB, int NPIXT, int NTHR> struct TileDma {
    // the placement may swap a pixel with its pair neighbour, so slots exist for an even number of pixels
    static constexpr int CPP = CB / 8, NCH = (CB == 16 ? (NPIXT + 7) / 8 * 8 : (NPIXT + 1) / 2 * 2) * CPP, ROUNDS = (NCH + NTHR - 1) / NTHR;
    static constexpr int ELEMS = (NCH + 63) / 64 * 64 * 8;   // LDS footprint: whole wave instructions (tail lanes fetch zeros)
};

// DMA descriptor of one operand tile (NPIXT pixels x CB channels; tile-local pixel p = (p / IWT, p % IWT)).  Everything
// that does not depend on the tile position is computed once per kernel: per DMA round a lane keeps the element offset
// of its source chunk from the tile-origin pixel and its (row, column) inside the tile.  Per tile that leaves one 64-bit
// add per round for interior tiles (the bounds tests run only for tiles that touch the image border).
// Every wave issues exactly ROUNDS instructions with all lanes active (lanes without a source read g_zero16, waves whose
// 64 chunks lie wholly past the tile write a shared 1 KB scratch), so the number of DMA instructions in flight is a
// compile-time quantity the waits can count on.
template <int CB, int PS, int IWT, int NPIXT, int NTHR> struct TileIssuer {
    typedef TileDma<CB, NPIXT, NTHR> D;
    int rel[D::ROUNDS];     // element offset of the source chunk from the tile origin; -1: none (pad lane / channel tail)
    int ij[D::ROUNDS];      // tile row | tile column << 16
    __device__ __forceinline__ void init(int Wimg, int C, int cvalid, int tid) {
#pragma unroll
        for (int r = 0; r < D::ROUNDS; ++r) {
            const int L = r * NTHR + tid;                   // LDS chunk this lane fills
            int pix, c;
            Place<CB, PS>::source(L, pix, c);
            const int il = pix / IWT, jl = pix - il * IWT;
            ij[r] = il | (jl << 16);
            rel[r] = (L < D::NCH && pix < NPIXT && c * 8 < cvalid) ? (il * Wimg + jl) * C + c * 8 : -1;
        }
    }
    // origin = address of image pixel (y0, x0) (may lie outside the image: only in-bounds lanes dereference their offset)
    __device__ __forceinline__ void issue(const bf16* origin, int y0, int x0, int Himg, int Wimg, bool interior, unsigned slot_addr, unsigned scratch,
                                          int tid) const {
        const bf16* zsrc = reinterpret_cast<const bf16*>(&g_zero16);
#pragma unroll
        for (int r = 0; r < D::ROUNDS; ++r) {
            bool ok = rel[r] >= 0;
            if (!interior) {                                // tile-uniform branch
                const int ih = y0 + (ij[r] & 0xffff), iw = x0 + (ij[r] >> 16);
                ok = ok && (unsigned)ih < (unsigned)Himg && (unsigned)iw < (unsigned)Wimg;
            }
            const bf16* src = ok ? origin + rel[r] : zsrc;
            const int L0 = r * NTHR + (tid & ~63);
            dma16(src, L0 < D::NCH ? slot_addr + (unsigned)L0 * 16u : scratch);
        }
    }
};

// n / d for 0 <= n < 2^23 with a precomputed float reciprocal (tile-index decoding without integer division sequences)
__device__ __forceinline__ int fdiv(int n, int d, float inv_d) {
    int qv = (int)((float)n * inv_d);
    const int r = n - qv * d;
    qv += (r >= d) - (r < 0);
    return qv;
}

template <int CBN, int CBC, int TH, int SH, int SW, int NW, int NSTAGE, bool NORM>
__global__ __launch_bounds__(NW * 64) void wgrad_dma_kernel(WgradArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr int WN = CBN >= 32 ? CBN / 32 : 1, WC = CBC >= 32 ? CBC / 32 : 1, WK = NW / (WN * WC);
    constexpr int IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3, NPX = IH * IW, NPY = TH * TW;
    typedef TileDma<CBN, NPY, NTHR> DY_;
    typedef TileDma<CBC, NPX, NTHR> DX_;
    constexpr int YS = DY_::ELEMS, BUF = YS + DX_::ELEMS;              // elements per ring slot
    constexpr int DMA_PER_TILE = DY_::ROUNDS + DX_::ROUNDS;
    constexpr int KUNROLL = NPY / 16 / WK <= 2 ? 2 : 1;     // k-steps per wave per tile: unroll the short loops only (register budget)
    static_assert(NPY % 32 == 0, "a tile row is 32 pixels: paired k-steps stay inside a row");
    static_assert(WK >= 1 && WN * WC * WK == NW && NSTAGE >= 2, "wave split / ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* ring = reinterpret_cast<bf16*>(smem_raw);                      // [NSTAGE][BUF]
    // after the 1 KB DMA scratch; NORM: 8 x 512-byte slots [mean | rstd][CBC], one per image the ring currently spans
    float* sstat_base = reinterpret_cast<float*>(ring + NSTAGE * BUF + 512);
    const unsigned ring_addr = lds_address(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (WN * WC), wn = (wave % (WN * WC)) / WC, wc = wave % WC;
    const int ncb = cdiv(a.CIN, CBC);
    const int n0 = (blockIdx.y / ncb) * CBN, c0 = (blockIdx.y % ncb) * CBC;
    const int tiles_per_img = a.tiles_h * a.tiles_w, ntiles = a.B * tiles_per_img;
    const bf16* X = (const bf16*)a.x;
    const bf16* DY = (const bf16*)a.dy;

    TileIssuer<CBN, 1, TW, NPY, NTHR> iy;
    TileIssuer<CBC, SW, IW, NPX, NTHR> ix;
    typedef Place<CBN, 1> PY;
    typedef Place<CBC, SW> PX;
    iy.init(a.Wo, a.COUT, a.COUT - n0, tid);
    ix.init(a.Wr, a.CIN, a.CIN - c0, tid);
    const float inv_tpi = 1.0f / (float)tiles_per_img, inv_tw = 1.0f / (float)a.tiles_w;
    int iss_b = -1, iss_slot = 0;      // NORM: image and statistics slot of the most recently issued tile
    auto issue = [&](int tile, int slot) {   // tile >= ntiles: a dummy issue (zeros) that keeps the in-flight count exact
        const bool live = tile < ntiles && !DBG(1);
        if (DBG(8)) return;
        const int tl = live ? tile : 0;
        const int b = fdiv(tl, tiles_per_img, inv_tpi), rem = tl - b * tiles_per_img;
        const int th = fdiv(rem, a.tiles_w, inv_tw), tw = rem - th * a.tiles_w;
        const int oh0 = live ? th * TH : -(1 << 20), ow0 = tw * TW;            // a dead tile fails every bounds test
        const int ih0 = live ? oh0 * SH - 1 : -(1 << 20), iw0 = ow0 * SW - 1;
        const bool in_y = live && oh0 + TH <= a.Ho && ow0 + TW <= a.Wo;
        const bool in_x = live && ih0 >= 0 && ih0 + IH <= a.Hr && iw0 >= 0 && iw0 + IW <= a.Wr;
        const unsigned ya = ring_addr + (unsigned)(slot * BUF) * 2u, scratch = ring_addr + (unsigned)(NSTAGE * BUF) * 2u;
        if constexpr (NORM) {
            // The tile's image statistics ride the same DMA queue, AHEAD of the tile's own pieces (in-order completion: landed when the tile has): on an image change (block-uniform, so every wave's
            // instruction count stays equal) wave 0 fetches mean | rstd of this block's channels into the next 1 KB
            // statistics slot; the other waves send zeros to the scratch.  An ordinary load here would make the compiler
            // wait for vmcnt(0) and drain the ring once per image.
            if (live && b != iss_b) {
                iss_b = b;
                iss_slot = (iss_slot + 1) & 7;
                const int l4 = lane * 4;                       // this lane's first channel (mean) or CBC + channel (rstd)
                const float* src = l4 < CBC ? a.mean : a.rstd;
                const int ch = c0 + (l4 < CBC ? l4 : l4 - CBC);
                const bool ok = l4 < 2 * CBC && ch < a.CIN;
                if (tid >= 64) dma16(&g_zero16, scratch);        // keeps the other waves' DMA count in step
                else if (lane < 32)                              // 512-byte slot: the upper half-wave writes nothing
                    dma16(ok ? (const void*)(src + (long)b * a.CIN + ch) : (const void*)&g_zero16,
                          ring_addr + (unsigned)(NSTAGE * BUF) * 2u + 1024u + (unsigned)iss_slot * 512u);
            }
        }
        iy.issue(DY + ((long)b * a.Ho * a.Wo + (long)th * TH * a.Wo + ow0) * a.COUT + n0, oh0, ow0, a.Ho, a.Wo, in_y, ya, scratch, tid);
        ix.issue(X + ((long)b * a.Hr * a.Wr + (long)(th * TH * SH - 1) * a.Wr + iw0) * a.CIN + c0, ih0, iw0, a.Hr, a.Wr, in_x, ya + (unsigned)YS * 2u,
                 scratch, tid);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = a.db != nullptr && (blockIdx.y % ncb) == 0;     // one cin-block column of the grid owns the bias sums
    float bsum = 0.f;

    // lane roles of the transposing reads (scratch/trtest.hip): lane (q, p, cb, hh) addresses pixel row q, channels
    // 16 cb + 4 p of its 16-lane group and receives 4 consecutive pixels of channel 16 cb + (lane & 15)
    // 16 x 16 channels (PAIR): the cb = 1 half of the wave would only feed rows / columns >= 16 of the 32x32 product, which
    // nobody wants.  It takes the NEXT 16 pixels of the tile row instead, same 16 channels: the product becomes block diagonal
    // -- D[0:16][0:16] sums pixels k0..k0+15, D[16:32][16:32] sums k0+16..k0+31 -- so one MFMA and one pair of LDS reads
    // cover 32 pixels (the kernel is bound by its ds_read_tr traffic: half the reads, half the MFMAs); the epilogue adds
    // the two diagonal blocks.  Mixed 16/32 tiles keep the plain form: cb = 1 lanes of the 16-channel operand read the
    // neighbouring pixel's bytes (valid LDS) into rows / columns the epilogue never stores.
    constexpr bool PAIR = CBN == 16 && CBC == 16;
    constexpr int KS = PAIR ? 32 : 16;                           // pixels per k-step
    const int q = (lane & 15) >> 2, cb = (lane >> 4) & 1, chan = (PAIR ? 0 : cb * 16) + (lane & 3) * 4, hh = lane >> 5;
    const int cha = (CBN >= 32 ? wn * 32 : 0) + chan, chb = (CBC >= 32 ? wc * 32 : 0) + chan;

    int tile = blockIdx.x;
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) issue(tile + st * (int)gridDim.x, st);
    int cur = 0, stat_b = -1, stat_slot = 0;
    const float* sstat = sstat_base;
    float rs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, nb8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // NORM: this thread's channel statistics
    // 128-byte rows (CBC = 64): the placement swaps the 64-byte halves of every other pixel pair, so a thread meets TWO channel
    // chunks (c and c ^ 4): the second set.  (Reading the statistics from LDS per element instead cost 16 scalar LDS reads per
    // 16-byte chunk: 80 per thread and tile on the strided 64-channel layers.)
    float rs8b[CBC == 64 ? 8 : 1], nb8b[CBC == 64 ? 8 : 1];
#pragma unroll
    for (int e = 0; e < (CBC == 64 ? 8 : 1); ++e) { rs8b[e] = 0.f; nb8b[e] = 0.f; }
    for (; tile < ntiles; tile += gridDim.x) {
        // tile `cur` has landed in every wave's share; the slot consumed in the previous iteration is free again
        if (!DBG(16)) dma_wait_barrier<(NSTAGE - 2) * DMA_PER_TILE>();
        const int nslot = cur == 0 ? NSTAGE - 1 : cur - 1;
        bf16* Ys = ring + cur * BUF;
        bf16* Xs = Ys + YS;
        if constexpr (NORM) {
            // xhat = x * rstd - mean * rstd in place; out-of-image halo pixels stay 0 (the conv pads the NORMALISED input).
            // The pass walks the same (round, lane) -> chunk map as the DMA, so validity and tile coordinates come from the
            // issuer's descriptors; with rows of <= 64 bytes a thread always meets the same 8 channels and keeps their
            // statistics in registers (refreshed once per image).
            const int b = fdiv(tile, tiles_per_img, inv_tpi), rem = tile - b * tiles_per_img;
            if (b != stat_b) {      // block-uniform: the slot was filled by the DMA that preceded this tile's own (already landed)
                stat_b = b;
                stat_slot = (stat_slot + 1) & 7;
                sstat = sstat_base + stat_slot * 128;
                const int cf = (tid % (CBC / 8)) * 8;          // NTHR is a multiple of the chunks per row: fixed for the thread
#pragma unroll
                for (int e = 0; e < 8; ++e) { rs8[e] = sstat[CBC + cf + e]; nb8[e] = -sstat[cf + e] * rs8[e]; }
                if constexpr (CBC == 64) {
                    const int cg = cf ^ 32;                    // chunk c ^ 4
#pragma unroll
                    for (int e = 0; e < 8; ++e) { rs8b[e] = sstat[CBC + cg + e]; nb8b[e] = -sstat[cg + e] * rs8b[e]; }
                }
            }
            issue(tile + (NSTAGE - 1) * (int)gridDim.x, nslot);
            const int th = fdiv(rem, a.tiles_w, inv_tw);
            const int ih0 = th * TH * SH - 1, iw0 = (rem - th * a.tiles_w) * TW * SW - 1;
            const bool in_x = ih0 >= 0 && ih0 + IH <= a.Hr && iw0 >= 0 && iw0 + IW <= a.Wr;
#pragma unroll
            for (int r = 0; r < DX_::ROUNDS; ++r) {
                if (ix.rel[r] < 0) continue;                  // pad lane / channel tail: nothing was fetched
                if (!in_x) {
                    const int ih = ih0 + (ix.ij[r] & 0xffff), iw = iw0 + (ix.ij[r] >> 16);
                    if ((unsigned)ih >= (unsigned)a.Hr || (unsigned)iw >= (unsigned)a.Wr) continue;
                }
                bf16* px = Xs + (long)(r * NTHR + tid) * 8;
                bf16x8 v = *reinterpret_cast<bf16x8*>(px);
                if constexpr (CBC <= 32) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaf((float)v[e], rs8[e], nb8[e]);
                } else {
                    int pix, c;
                    PX::source(r * NTHR + tid, pix, c);
                    const bool other = c != (tid % (CBC / 8));      // the slot holds chunk (L % 8) ^ 4 of its pixel
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaf((float)v[e], other ? rs8b[e] : rs8[e], other ? nb8b[e] : nb8[e]);
                }
                *reinterpret_cast<bf16x8*>(px) = v;
            }
            lds_barrier();
        } else {
            issue(tile + (NSTAGE - 1) * (int)gridDim.x, nslot);
        }
        if (do_bias && !DBG(4)) {   // bias gradient: column sums of the staged dY tile (thread = channel x pixel phase)
            constexpr int NPH = NTHR / CBN;
            const int ch = tid % CBN;
            for (int pix = tid / CBN; pix < NPY; pix += NPH) bsum += (float)Ys[PY::off(pix, ch)];
        }
#pragma unroll KUNROLL
        for (int k0 = wk * KS; k0 < (DBG(2) ? 0 : NPY); k0 += WK * KS) {
            const int pk = k0 + (PAIR ? 16 * cb : 0) + 8 * hh + q;   // pixels pk..pk+3 (u = 0) and pk+4..pk+7 (u = 1), same tile row
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(Ys + PY::off(pk, cha)));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(Ys + PY::off(pk + 4, cha)));
            const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const int xp = ((pk >> 5) * SH) * IW + (pk & 31) * SW;
            // (Cutting the three taps of a tap row out of ONE 12-pixel register window -- 11 transposing reads per k-step instead
            // of 20 -- was measured: no gain (32x32 channels 593 -> 607 us, 32x16 547 -> 599 us); the phase is not LDS-read bound.)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int px = xp + (tap / 3) * IW + (tap % 3);
                const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(Xs + PX::off(px, chb)));
                const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsV4*)(Xs + PX::off(px + 4 * SW, chb)));
                const bf16x8 bf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                mma32(acc[tap], af, bf);
            }
        }
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }
    dma_drain();        // the trailing dummy DMAs must not outlive the workgroup's LDS
    // Device-scope float atomics execute at the memory side and serialise per address (~3 ns each on one cache line), so
    // every partial sum is first combined inside the workgroup through LDS (the ring is dead by now).
    float* red = reinterpret_cast<float*>(smem_raw);        // [NW][16][64]
    if (do_bias) {                                          // block-uniform
        lds_barrier();
        red[tid] = bsum;
        lds_barrier();
        if (tid < CBN) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < NTHR / CBN; ++k) v += red[k * CBN + tid];
            if (n0 + tid < a.COUT) atomicAdd(&a.db[n0 + tid], v);
        }
    }
    // Accumulate into dw[n][tap][c]: the WK waves that hold partial sums of the same 32x32 block are combined first, one
    // atomic per element per workgroup.
    constexpr int GROUPS = WN * WC, PER_THR = GROUPS * 1024 / NTHR;
    static_assert(GROUPS * 1024 % NTHR == 0 && NW * 1024 * sizeof(float) <= 2 * BUF * sizeof(bf16), "reduction scratch");
#pragma unroll
    for (int tap = 0; tap < (DBG(32) ? 0 : 9); ++tap) {
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tap][r];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < PER_THR; ++i) {
            const int e = i * NTHR + tid;                   // (group, r, lane)
            const int g = e >> 10, r = (e >> 6) & 15, ln = e & 63;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < WK; ++k) v += red[((k * GROUPS + g) * 16 + r) * 64 + ln];
            const int gwn = g / WC, gwc = g % WC;
            const int cl = ln & 31, nl = acc_row(r, ln);
            const int c = c0 + (CBC >= 32 ? gwc * 32 : 0) + cl, n = n0 + (CBN >= 32 ? gwn * 32 : 0) + nl;
            if constexpr (CBN == 16 && CBC == 16) {        // the two diagonal 16x16 blocks hold the two pixel halves of every k-step
                if ((cl < 16) == (nl < 16)) {
                    const int c2 = c0 + (cl & 15), n2 = n0 + (nl & 15);
                    if (c2 < a.CIN && n2 < a.COUT) atomicAdd(&a.dw[((long)n2 * 9 + tap) * a.CIN + c2], v);
                }
            } else if (c < a.CIN && n < a.COUT && (CBC >= 32 || cl < 16) && (CBN >= 32 || nl < 16)) atomicAdd(&a.dw[((long)n * 9 + tap) * a.CIN + c], v);
        }
    }
}

template <int CBN, int CBC, int TH, int SH, int SW, int NW, int NSTAGE, bool NORM> int launch(WgradArgs a, hipStream_t s) {
    a.tiles_w = cdiv(a.Wo, TW);
#ifdef OMR_WGRAD_DEBUG
    { const char* e = getenv("OMR_WGRAD_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    a.tiles_h = cdiv(a.Ho, TH);
    constexpr int IH = (TH - 1) * SH + 3, IW = (TW - 1) * SW + 3;
    constexpr size_t shm = (size_t)NSTAGE * (TileDma<CBN, TH * TW, NW * 64>::ELEMS + TileDma<CBC, IH * IW, NW * 64>::ELEMS) * sizeof(bf16) +
                           1024 + (NORM ? 8 * 512 : 0);
    static_assert(shm <= 160 * 1024, "LDS ring does not fit");
    auto kern = wgrad_dma_kernel<CBN, CBC, TH, SH, SW, NW, NSTAGE, NORM>;
    static std::atomic<int> occ_cache{0};   // resident blocks per CU of this instantiation (0 -> value once; see conv3x3_mfma.h)
    int occv = occ_cache.load(std::memory_order_acquire);
    if (occv == 0) {
        if (shm > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return OMR_ERR_LAUNCH;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NW * 64, shm) != hipSuccess || occ < 1) occ = 1;
        occv = occ;
        occ_cache.store(occ, std::memory_order_release);
    }
    const int gy = cdiv(a.COUT, CBN) * cdiv(a.CIN, CBC);
    const int ntiles = a.B * a.tiles_h * a.tiles_w;
    if ((long)a.B * a.tiles_h * a.tiles_w >= (1 << 23) || (long)IH * a.Wr * a.CIN >= (1L << 30)) return OMR_ERR_UNSUPPORTED;   // fdiv / 32-bit offsets
    int gx = (NUM_CU * occv + gy - 1) / gy; if (gx < 1) gx = 1; if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(NW * 64), shm, s, a);
    OMR_CHECK_LAUNCH();
    return OMR_OK;
}

// Tile / ring configuration per channel-block pair: one 8-wave workgroup per CU, ring as deep as 160 KB of LDS allows
// (what hides the ~4 us loaded HBM latency is bytes in flight per CU, not occupancy).
template <int SH, int SW, bool NORM> int pick(const WgradArgs& a, hipStream_t s) {
    const int cn = a.COUT > 32 ? 64 : a.COUT > 16 ? 32 : 16, cc = a.CIN > 32 ? 64 : a.CIN > 16 ? 32 : 16;
    if constexpr (SH == 1 && SW == 1) {
        // (4-row tiles with a 3-deep ring -- two tiles in flight -- were measured in round 3: 439 -> 457 us at 64 -> 64, 402 -> 435 us at 128 -> 128)
        if (cn == 64 && cc == 64) return launch<64, 64, 8, 1, 1, 8, 2, NORM>(a, s);
        if (cn == 64 && cc == 32) return launch<64, 32, 8, 1, 1, 8, 2, NORM>(a, s);
        if (cn == 32 && cc == 64) return launch<32, 64, 8, 1, 1, 8, 2, NORM>(a, s);
        // 32-channel tiles: TWO 4-wave workgroups per CU with a 2-deep ring each (one's DMA wait behind the other's MFMA loop:
        // 726 -> 648 us and 730 -> 695 us against one 8-wave workgroup with a 4-5 deep ring; the paired 16x16 form loses: 383 -> 594)
        if (cn == 32 && cc == 32) return launch<32, 32, 8, 1, 1, 4, 2, NORM>(a, s);
        if (cn == 32 && cc == 16) return launch<32, 16, 8, 1, 1, 4, 2, NORM>(a, s);
        if (cn == 16 && cc == 16) return launch<16, 16, 8, 1, 1, 8, 6, NORM>(a, s);
    }
    if constexpr (SH == 2 && SW == 2) {   // input tile is 4x the output tile: fewer rows per tile
        if (cn == 64 && cc == 64) return launch<64, 64, 2, 2, 2, 8, 3, NORM>(a, s);
        if (cn == 32 && cc == 32) return launch<32, 32, 4, 2, 2, 8, 3, NORM>(a, s);
        if (cn == 16 && cc == 16) return launch<16, 16, 8, 2, 2, 8, 3, NORM>(a, s);
    }
    if constexpr (SH == 2 && SW == 1) {
        if (cn == 64 && cc == 64) return launch<64, 64, 4, 2, 1, 8, 2, NORM>(a, s);
        if (cn == 32 && cc == 32) return launch<32, 32, 8, 2, 1, 8, 2, NORM>(a, s);
    }
    return OMR_ERR_UNSUPPORTED;
}

}  // namespace

int omr_wgrad_dma_bf16(const WgradArgs& a, hipStream_t s) {
    if (a.CIN % 8 || a.COUT % 8) return OMR_ERR_UNSUPPORTED;
    const bool norm = a.mean != nullptr;
    if (norm && ((((uintptr_t)a.mean) | ((uintptr_t)a.rstd)) & 15)) return OMR_ERR_UNSUPPORTED;      // statistics are fetched in 16-byte pieces
    if (a.sh == 1 && a.sw == 1) return norm ? pick<1, 1, true>(a, s) : pick<1, 1, false>(a, s);
    if (a.sh == 2 && a.sw == 2) return norm ? pick<2, 2, true>(a, s) : pick<2, 2, false>(a, s);
    if (a.sh == 2 && a.sw == 1) return norm ? pick<2, 1, true>(a, s) : pick<2, 1, false>(a, s);
    return OMR_ERR_UNSUPPORTED;
}
