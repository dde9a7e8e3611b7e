// KV-cached greedy decoding as ONE host call per run of tokens (SURVEY.md section 8b `decode_step`, section 8f rank 1).
// Reference loop: Transformer.validation_step / get_pred_seq_and_pred_prob_seq (src/transformer/model.py:182-193,247-260)
// re-runs the whole decoder over the prefix for every token and reads the argmax back to the host each time.  Here the
// host-side executor below issues the ~12 kernels per decoder layer of ONE new position back to back from C++ (no Python, no
// per-kernel argument marshalling), appends that position's self-attention K|V to the cache by letting the K|V projection
// GEMM write straight into its cache row, reads the cross-attention K|V that were projected once per input, and chains the
// chosen token to the next step THROUGH DEVICE MEMORY (the embedding kernel of step t+1 reads the token the argmax kernel of
// step t wrote), so n_steps tokens are produced without a single host synchronisation.  Same kernels and the same per-row
// arithmetic as the training forward pass: the tokens equal the full re-run's (tests/test_model_gpu.py).
#include "omr_common.h"
#include "omr_hip.h"

namespace {

struct Ws {          // activation scratch of one step, carved out of the caller's workspace
    char* x; char* q; char* o; char* proj; char* h; char* logits; float* logits32; float* lse; float* mean; float* rstd;
    float* split; long split_floats;
    unsigned char* a8; float* sa8;          // fp8 mode: the quantised input rows of the current GEMM and their scales
};

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

size_t carve(const omr_decode_desc& d, char* base, Ws* w) {
    const size_t es = d.dtype == OMR_BF16 ? 2 : 4, B = (size_t)d.B;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    char* x = take(B * d.d * es); char* q = take(B * d.d * es); char* o = take(B * d.d * es); char* proj = take(B * d.d * es);
    char* h = take(B * (size_t)d.ff * es); char* logits = take(B * (size_t)d.ldv * es);
    float* l32 = (float*)take(B * (size_t)d.ldv * 4); float* lse = (float*)take(B * (size_t)d.nhead * 4);
    float* mean = (float*)take(B * 4); float* rstd = (float*)take(B * 4);
    const int smax = d.S > d.max_len ? d.S : d.max_len;                // key-split partials of the longer of the two attentions
    const long sf = omr_attn_split_workspace_floats(d.B, d.nhead, 1, smax, d.d / d.nhead);
    float* split = (float*)take((size_t)sf * 4);
    const size_t kmax = (size_t)(d.d > d.ff ? d.d : d.ff);
    unsigned char* a8 = (unsigned char*)take(d.fp8 ? B * ((kmax + 15) / 16 * 16) : 0);
    float* sa8 = (float*)take(d.fp8 ? B * 4 : 0);
    if (w) *w = Ws{x, q, o, proj, h, logits, l32, lse, mean, rstd, split, sf, a8, sa8};
    return off;
}

#define TRY(call) do { int rc__ = (call); if (rc__ != OMR_OK) return rc__; } while (0)

}  // namespace

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
