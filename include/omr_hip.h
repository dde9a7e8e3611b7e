/* libomr_hip.so -- C ABI of the MI355X (gfx950) encoder/decoder hot path.
 *
 * The reference (mariaalfaroc/omr_a2s_multimodal_transformer) has no FFI: its hot path is the set of ATen ops that
 * src/transformer/{encoder,decoder,model}.py reach through torch.nn.  Each entry point below
 * replaces one of those op families; the reference call site it stands in for is cited per function.
 *
 * Conventions: every function returns 0 (OMR_OK) or a negative error code and never throws; the caller owns
 * all buffers (device pointers); `stream` is a hipStream_t passed as void*; dtype is OMR_F32 (0) or OMR_BF16 (1)
 * and names the activation/weight element type (accumulation is always fp32); "small vectors" (bias, LayerNorm
 * gamma/beta, statistics) are always fp32.  Encoder activations are NHWC; token matrices are row-major [rows][ld].
 * No hidden state: every call is re-entrant across streams and host threads (the only process-wide data are per-kernel
 * launch constants -- resident blocks per CU, dynamic-LDS opt-in -- cached in atomics on first use; one process drives one GPU).
 */
#ifndef OMR_HIP_H
#define OMR_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define OMR_DTYPE_F32 0
#define OMR_DTYPE_BF16 1

int omr_abi_version(void);

/* ---- elementwise / gather ------------------------------------------------------------------------------ */
int omr_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
/* residual adds: encoder.py:289 (x + xt) */
int omr_add(int dtype, const void* a, const void* b, void* out, long n, void* stream);
/* ReLU(+dropout) backward: dx = dy * (y > 0) * scale   (aten::threshold_backward; encoder.py:163-176) */
int omr_relu_bwd(int dtype, const void* dy, const void* y, void* dx, long n, float scale, void* stream);
/* nn.Dropout / nn.Dropout2d (encoder.py:98-104, decoder.py:19, model.py:31): counter-based mask, regenerated not stored.
 * channel_mode=1: one decision per (sample, channel) of an NHWC tensor with `per_sample` = H*W*C elements per sample. */
int omr_dropout(int dtype, const void* x, void* out, long n, float p, unsigned long long seed, int channel_mode, long per_sample, int C,
                void* stream);
/* nn.Embedding + PositionalEncoding1D: out[m] = table[tok[m]] + pe[m % T]   (decoder.py:124) */
int omr_embed_pe_fwd(int dtype, const long* tokens, const void* table, const float* pe, void* out, long M, int T_len, int d, int vocab,
                     void* stream);
/* embedding backward: dtable[tok[m]] += dout[m], PAD row skipped (nn.Embedding padding_idx, decoder.py:73-77) */
int omr_embed_bwd(int dtype, const long* tokens, const void* dout, float* dtable, long M, int d, int pad_idx, int vocab, void* stream);
/* PositionalEncoding2D on NHWC maps: out[b,i,j,c] = x[b,i,j,c] + pe[i,j,c], pe laid out [maxh][maxw][C]   (model.py:45-47) */
int omr_add_pe2d(int dtype, const void* x, const float* pe_hwc, void* out, int B, int h, int w, int C, int maxh, int maxw, void* stream);
/* bias gradients: db[n] += sum_m dy[m][n] */
int omr_colsum(int dtype, const void* dy, float* db, long M, int N, long ld, void* stream);
/* torch.optim.Adam step over one flat fp32 buffer (model.py:134-139; torch optim/adam.py:347); step is 1-based.
 * p_bf16 (nullable) receives the bf16 compute copy of the updated parameters in the same pass. */
int omr_adam(float* p, const float* g, float* m, float* v, void* p_bf16, long n, int step, float lr, float b1, float b2, float eps,
             float grad_scale, void* stream);
/* greedy token pick: argmax(dim=-1) / topk(1) of the last-step logits (model.py:187,253), one row per decoded sample
 * (x [rows][ld], first n columns); first-index tie rule */
int omr_argmax(const float* x, int rows, int n, long ld, long* idx_out, float* val_out, void* stream);
/* weighted late fusion (src/multimodal/weighted_multimodal/test.py:50-61): the token of one decoding step of two unimodal models
 * run in lock-step, argmax(alpha * softmax(logits_a) + (1 - alpha) * softmax(logits_b)) over fp32 rows of n logits, first-index
 * tie rule; prob_out (nullable) receives the winning mixed probability.  alpha is rounded like the reference's Python float. */
int omr_weighted_argmax(const float* logits_a, const float* logits_b, int n, float alpha, long* idx_out, float* prob_out, void* stream);
/* audio front end (preprocessing.py:17-30): after the windowed DFT -- ONE omr_gemm of the centred,
 * hop-strided frames (lda = hop) against the Hann-weighted [cos | -sin] basis of the kept bins -- spec [frames][2*bins] holds
 * (re | im); this turns it into the reference's normalised log-spectrogram out [bins][frames] =
 * amplitude_to_db(|S|, ref=max, top_db=80) / 80 + 1.  spec is overwritten (magnitudes); max_ws: 4 bytes of device scratch. */
int omr_log_stft_post(float* spec, long frames, int bins, unsigned* max_ws, float* out, void* stream);
/* score-image front end (preprocessing.py:44-52: PIL convert("L") -> Image.resize (Pillow's default BICUBIC, antialiased,
 * 8-bit fixed point) -> ToTensor).  Bit-exact with Pillow.
 *   omr_resample_ksize / omr_resample_coeffs: HOST functions (no GPU work): taps per output pixel, and the tables
 *     bounds [out_size][2] = (first input index, count), coefs [out_size][ksize] (22 fractional bits); coeffs returns ksize.
 *   omr_image_gray_hpass: interleaved 8-bit pixels (channels 1 = L, 3 = RGB, 4 = RGBA; row_stride in bytes) -> luma ->
 *     horizontal pass -> dst [h][out_w] uint8.  bounds = coefs = NULL (out_w == w): conversion only.
 *   omr_image_vpass_to_float: vertical pass over src [h][w] uint8 -> value / 255 as out_dtype (OMR_F32 / OMR_BF16) into
 *     dst rows of dst_row_stride elements (a sample's slot inside the padded batch tensor, preprocessing.py:55-75).
 *     bounds = coefs = NULL (out_h == h): conversion only. */
int omr_resample_ksize(int in_size, int out_size);
int omr_resample_coeffs(int in_size, int out_size, int* bounds, int* coefs);
int omr_image_gray_hpass(const unsigned char* src, int h, int w, int channels, long row_stride, const int* bounds, const int* coefs,
                         int ksize, int out_w, unsigned char* dst, void* stream);
int omr_image_vpass_to_float(const unsigned char* src, int h, int w, const int* bounds, const int* coefs, int ksize, int out_h,
                             int out_dtype, void* dst, long dst_row_stride, void* stream);
/* late (prediction-level) fusion, src/multimodal/smith_waterman/test.py:136-150: Smith-Waterman local alignment of the
 * image model's and the audio model's token sequences.  HOST function (no GPU work).  Restates swalign 0.3.x's
 * LocalAlignment (package absent here: PARITY UNPINNED).  ref / query: token ids; ops receives one of 'm' 'i' 'd' per
 * alignment column (capacity nr + nq), r_pos / q_pos the 0-based start of the aligned region in ref / query; returns the
 * number of columns. */
/* Token noise of Transformer.apply_teacher_forcing (model.py:152-160) in one HOST call, continuing Python's own Mersenne
 * Twister: for each of `count` tokens (row-major B x T) draw random.random(); if it is < prob and the token is not pad_idx,
 * replace it by random.randint(0, vocab - 1).  mt_state[624] / *mt_index = random.getstate()[1] (words, position); both are
 * advanced exactly as the interpreter's calls would advance them, for random.setstate().  Host memory only. */
int omr_teacher_forcing_noise(long* tokens, long count, double prob, long pad_idx, long vocab, unsigned int* mt_state, int* mt_index);
int omr_sw_align(const int* ref, int nr, const int* query, int nq, int match, int mismatch, int gap_penalty,
                 int gap_extension_penalty, char* ops, int* r_pos, int* q_pos, int* score);
/* beam-search expansion (BASELINE config C5; an extension: the reference decodes greedily): per row the k largest
 * log_softmax values and their token ids, same tie rule as omr_argmax, idx_out / val_out [rows][k] */
int omr_topk_logprob(const float* x, int rows, int n, long ld, int k, long* idx_out, float* val_out, void* stream);

/* ---- normalisation ------------------------------------------------------------------------------------- */
/* nn.InstanceNorm2d(eps=1e-3, affine=False) on x[B][HW][C] (encoder.py:151-156,174,232); the apply is fused into the consumer
 * conv (in_mean / in_rstd arguments below).  Statistics are reduced DETERMINISTICALLY (bit-identical from run to run, like
 * the reference's CPU path): every producer block stores fp64 partials into its own slot of workspace[B][slots][C][2] and
 * the consumer adds an image's slots in index order -- no atomics.  workspace_bytes covers the slots plus the compact
 * [B][C][2] sums the backward apply reads; slots = omr_instnorm_slots(B, HW) for the stand-alone passes
 * (omr_instnorm_stats / omr_instnorm_bwd) or omr_conv3x3_stat_slots(B, Ho, Wo) when a conv epilogue is the producer
 * (a conv launch that uses fewer blocks per image than there are slots writes zeros into the rest: no clearing needed). */
int omr_instnorm_slots(int B, long HW);
long omr_instnorm_workspace_bytes(int B, int C, int slots);
int omr_instnorm_stats(int dtype, const void* x, float* mean, float* rstd, int B, long HW, int C, float eps, void* workspace, void* stream);
int omr_instnorm_finalize(const void* workspace, int slots, float* mean, float* rstd, int B, long HW, int C, float eps, void* stream);
int omr_instnorm_bwd_apply(int dtype, const void* dxhat, const void* x, const float* mean, const float* rstd, void* dx, int B, long HW, int C,
                           int relu_mask, float relu_scale, void* workspace, int slots, void* stream);
int omr_instnorm_bwd(int dtype, const void* dxhat, const void* x, const float* mean, const float* rstd, void* dx, int B, long HW, int C,
                     int relu_mask, float relu_scale, void* workspace, void* stream);
/* post-norm residual with the sublayer dropout fused: out = LayerNorm(dropout(x) + res) (eps 1e-5), torch
 * nn/modules/transformer.py:1146-1154 (x = dropout1/2/3(sublayer), res = the residual stream).  drop_p = 0: plain add.
 * The mask is omr_dropout's counter-based mask (same seed / element-index convention) and is regenerated in backward.
 * Backward: ds = gradient of res (and of x when drop_p = 0); with drop_p > 0, dx receives the gradient of x. */
int omr_add_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* out, float* mean,
                          float* rstd, long M, int d, float eps, float drop_p, unsigned long long drop_seed, void* stream);
int omr_add_layernorm_bwd(int dtype, const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                          const float* rstd, void* ds, float* dgamma, float* dbeta, long M, int d, float drop_p,
                          unsigned long long drop_seed, void* dx, void* stream);

/* ---- GEMM: C[M,N] (+)= act(opA(A) . opB(B)^T + bias) ------------------------------------------------------ */
/* aten::linear/addmm/mm of the decoder layers (torch nn/modules/transformer.py:1158-1199), 1x1 point_conv
 * (encoder.py:65-70), Conv1d(k=1) head (decoder.py:98-102,146).  transA/transB: operand stored reduction-major.
 * split_k > 1 accumulates into fp32 C with atomics (C must hold the running sum / zeros).
 * colsum_a (nullable, transA only): colsum_a[m] += sum_k A[k][m] -- the bias gradient of a linear layer comes out of the
 * same pass that computes its weight gradient dW = dY^T . X.
 * drop_p > 0: nn.Dropout on the (activated) output, omr_dropout's mask over the flat [M][ldc] element index (FFN dropout,
 * torch nn/modules/transformer.py:1197-1199).
 * row_group_operand != 0: the weight-side operand is a row-group view -- logical row i is physical row
 * (i / row_group) * row_group_stride + row_group_base + i % row_group (row_group a multiple of 128); operand 1 = rows of B and
 * bias entries, 2 = reduction rows of a transposed B, 3 = rows of C and colsum_a entries.  One call then covers the K|V rows
 * of all decoder layers' packed in_proj_weight ([Wq;Wk;Wv], torch nn/modules/activation.py) laid back to back. */
int omr_gemm(int dtype, int c_dtype, int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C,
             long ldc, const float* bias, int relu, int accumulate, int split_k, float* colsum_a, float drop_p,
             unsigned long long drop_seed, int row_group, int row_group_stride, int row_group_base, int row_group_operand, void* stream);

/* fp8 path (BASELINE config 5 "fp8 MFMA weights"; an extension -- the reference has no fp8; inference only).  Operands are
 * OCP e4m3 with one fp32 scale per ROW: weights [N][K] quantised once (scale per output feature), activations [M][K] per
 * token on the fly.  omr_quantize_rows_fp8: scale[m] = absmax(row) / 448, q = round_to_fp8(x / scale) (x in `dtype`).
 * omr_gemm_fp8: C[M][N] = (A8 . W8^T) * scale_a[m] * scale_w[n] + bias[n] [-> ReLU], products on the fp8 MFMA
 * (v_mfma_f32_32x32x16_fp8_fp8), fp32 accumulation, C in c_dtype; K and both row strides multiples of 16. */
int omr_quantize_rows_fp8(int dtype, const void* x, long ldx, unsigned char* q, long ldq, float* scale, int M, int K, void* stream);
int omr_gemm_fp8(int c_dtype, int M, int N, int K, const unsigned char* A8, long lda, const float* scale_a, const unsigned char* W8, long ldb,
                 const float* scale_w, void* C, long ldc, const float* bias, int relu, void* stream);

/* Weight (and bias) gradients of SEVERAL linear layers in one launch: for each problem dw[n_out][n_in] += dy^T . x and
 * db[n_out] += column sums of dy (db nullable), reducing over `rows`; dy [rows][ld_dy], x [rows][ld_x] in `dtype`, dw / db fp32
 * accumulators (fp32 atomics: the buffers hold the running sums).  A linear's dW at d_model = 256 is too small to fill the
 * chip on its own (aten::mm of the decoder layers / 1x1 point_convs, torch nn/modules/transformer.py:1158-1199,
 * encoder.py:65-70): the backward pass collects them and launches them together.  row_group* (0 = off): dw rows / db
 * entries are a row-group view as in omr_gemm (operand 3).  `problems` is a HOST array. */
typedef struct omr_dw_problem {
    const void* dy; const void* x; float* dw; float* db;
    int rows, n_out, n_in, row_group; long ld_dy, ld_x, ld_dw; int row_group_stride, row_group_base;
} omr_dw_problem;
int omr_linear_wgrad_grouped(int dtype, int nprob, const omr_dw_problem* problems, void* stream);

/* ---- convolutions (NHWC) --------------------------------------------------------------------------------- */
/* nn.Conv2d 3x3 pad 1 (encoder.py:132-150) with fused bias + ReLU, optional fused InstanceNorm apply on the input
 * (in_mean/in_rstd [B][CIN]) and optional epilogue mask (y = mask>0 ? y*mask_scale : 0).  Weights [COUT][3][3][CIN].
 * dil_* > 1 reads the input as if zero-dilated: with flipped weights (omr_conv3x3_weight_flip) this is the data
 * gradient of a strided conv.  CIN == 1 takes the direct (non-MFMA) first-layer path. */
/* Fused on the output: MixDropout after the ReLU (drop_p > 0; encoder.py:165-179; elementwise mask keyed by the flat NHWC
 * index, or per (image, channel) when drop_channel_mode) and a per-(image, channel) reduction over the stored tile into
 * the fp64 slots stat_ws[B][stat_slots][COUT][2] (every slot is written; stat_slots = omr_conv3x3_stat_slots(B, Ho, Wo), an
 * upper bound on the persistent grid's blocks per image; deterministic, see "normalisation"): stat_mode 1 = {sum y, sum y^2}
 * (InstanceNorm statistics of this output, finalised by omr_instnorm_finalize), stat_mode 2 = {sum g, sum g*xhat} with
 * xhat = (stat_x - mean)*rstd (InstanceNorm backward sums when this call is the data gradient that produces g = dL/dxhat;
 * consumed by omr_instnorm_bwd_apply).
 * stat_mode 4 / 5 -- the data gradient through a normalise-on-load conv without ever storing g: mode 4 takes the sums of
 * mode 2 and stores nothing (y may be NULL); omr_instnorm_reduce_sums adds the slots of each image; mode 5 recomputes the
 * data gradient and applies the InstanceNorm backward in its epilogue, y = rstd * (g - mean(g) - xhat * mean(g * xhat)),
 * times (stat_x > 0) * mask_scale when `relu` is set (the ReLU / dropout backward of the layer that produced stat_x).  In
 * these two modes bias and out_mask must be NULL.  (The stand-alone form is omr_instnorm_bwd_apply on a stored g.) */
int omr_conv3x3_stat_slots(int B, int Ho, int Wo);
int omr_instnorm_reduce_sums(void* workspace, int slots, int B, int C, void* stream);
int omr_conv3x3_fwd(int dtype, const void* x, const void* w, const float* bias, void* y, const float* in_mean, const float* in_rstd,
                    const void* out_mask, float mask_scale, int B, int H, int W, int CIN, int COUT, int stride_h, int stride_w, int dil_h,
                    int dil_w, int Ho, int Wo, int relu, float drop_p, unsigned long long drop_seed, int drop_channel_mode,
                    int stat_mode, double* stat_ws, int stat_slots, const void* stat_x, const float* stat_mean, const float* stat_rstd,
                    void* stream);
int omr_conv3x3_weight_flip(int dtype, const void* w, void* wd, int COUT, int CIN, void* stream);
/* The same re-layout for n convs in ONE launch (descs is a HOST array): the data-gradient weights of every 3x3 conv are
 * refreshed once per optimizer step, behind omr_adam, so the backward pass (aten::convolution_backward of encoder.py:132-150)
 * carries no re-layout launches. */
typedef struct omr_flip_desc { const void* w; void* wd; int cout, cin; } omr_flip_desc;
int omr_conv3x3_weight_flip_grouped(int dtype, int n, const omr_flip_desc* descs, void* stream);
/* dw[COUT][3][3][CIN] (fp32) += dy^T * im2col(x);  db[COUT] (nullable, fp32) += column sums of dy (bias gradient) */
int omr_conv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, const float* in_mean, const float* in_rstd, int B, int H, int W,
                      int CIN, int COUT, int stride_h, int stride_w, int Ho, int Wo, void* stream);
/* The whole backward of a stride-1 3x3 conv with CIN, COUT in {16, 32} (bf16; (COUT, CIN) = (32,32), (32,16), (16,16)) in one pass
 * over its operands (aten::convolution_backward of nn.Conv2d, encoder.py:132-150): dx = conv^T(g) [* (x > 0) * mask_scale],
 * dw[COUT][3][3][CIN] += g^T im2col(x), db[COUT] (nullable) += column sums of g.  w_flipped = omr_conv3x3_weight_flip(w).
 * norm_y != NULL: g is dL/d(InstanceNorm output) of the layer above and norm_y that layer's input (= this conv's stored output);
 * the InstanceNorm backward (omr_instnorm_bwd_apply's arithmetic, sums from norm_workspace after omr_instnorm_reduce_sums)
 * and the ReLU / dropout mask of norm_y are applied while the tile is loaded, so the gradient w.r.t. this conv's output never
 * exists in memory.  x_mean / x_rstd != NULL (CIN = COUT = 16): the conv normalised its input on load; xhat = (x - mean) * rstd is
 * formed in LDS for the weight gradient and dx = dL/dxhat leaves with {sum dx, sum dx * xhat} reduced into the slots of
 * stat_workspace (omr_conv3x3_fwd stat_mode 2's layout: omr_instnorm_bwd_apply or the norm_y form above consumes them).
 * OMR_ERR_UNSUPPORTED for other shapes (callers then use omr_conv3x3_fwd + omr_conv3x3_wgrad). */
int omr_conv3x3_bwd_fused(const void* g, const void* x, const void* w_flipped, void* dx, float* dw, float* db, int B, int H, int W, int CIN,
                          int COUT, int mask_input, float mask_scale, const void* norm_y, const float* norm_mean, const float* norm_rstd,
                          const void* norm_workspace, int norm_slots, int relu_mask, float relu_scale, const float* x_mean, const float* x_rstd,
                          void* stat_workspace, int stat_slots, void* stream);
/* The same for the stride-(2,2) normalise-on-load conv with 32 -> 32 channels (ConvBlock 1's conv3, encoder.py:132-156): g is the
 * gradient of the [ceil(H/2)][ceil(W/2)] output, x the [H][W] input, dx = dL/dxhat with the InstanceNorm-backward sums in the slots. */
int omr_conv3x3_bwd_fused_s2(const void* g, const void* x, const void* w_flipped, void* dx, float* dw, float* db, int B, int H, int W,
                             const float* x_mean, const float* x_rstd, void* stat_workspace, int stat_slots, void* stream);
/* depthwise 3x3, stride 1, pad 1 (DepthSepConv2D.depth_conv, encoder.py:56-64); flip=1 mirrors the taps (data gradient) */
int omr_dwconv3x3(int dtype, const void* x, const void* w, const float* bias, void* y, const float* in_mean, const float* in_rstd,
                  const void* out_mask, float mask_scale, int B, int H, int W, int C, int flip, void* stream);
int omr_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, const float* in_mean, const float* in_rstd, int B,
                        int H, int W, int C, void* stream);

/* ---- attention ------------------------------------------------------------------------------------------ */
/* softmax(Q K^T / sqrt(hd) + masks) V per head (torch nn/functional.py multi_head_attention_forward; decoder.py:86-95,
 * model.py:292-297,323).  q/k/v/o are [B][rows][ld*] with head h at columns [h*hd, (h+1)*hd).  key_bias [B][S] is
 * ADDED (float +1.0 padding masks, -inf for bool masks); causal/window per decoder.py:191-217; blk_lq/blk_lkv [B]
 * reproduce CrossAttention.create_attention_mask incl. its (b*H+h)%B tiling (model.py:343-354). */
int omr_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                 long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, int causal, int window,
                 const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p, unsigned long long seed,
                 const unsigned long long* drop_words, void* stream);
/* Attention-probability dropout (nn.MultiheadAttention dropout, decoder.py:91; model.py:292-297).  The keep bits are a pure
 * function of (seed, b, h, query, key).  With dropout_p > 0 the attention kernels do not hash: they READ the bits, 1 per score,
 * from `drop_words` -- omr_attn_dropout_words_count(B, H, T, S) 64-bit words that omr_attn_dropout_words fills for
 * (dropout_p, seed), once per (layer, step), in the accumulator layout of the kernels (word = one register of a 32-query x
 * 64-key tile, bit = lane): forward and dQ take a word as the SGPR mask of one v_cndmask per score, dK/dV reads a 32-bit
 * column of the same words per key.  Forward and backward of a layer get the same buffer.  drop_words may be NULL when
 * dropout_p == 0. */
long omr_attn_dropout_words_count(int B, int H, int T, int S);
int omr_attn_dropout_words(unsigned long long* words, int B, int H, int T, int S, float dropout_p, unsigned long long seed, void* stream);
/* omr_attn_fwd / omr_attn_bwd with caller-provided scratch that lets the library split the KEYS of a (batch, head, query block)
 * over several workgroups where that fills the chip better (the query-per-lane kernels have B*H*T/32 waves whatever S is:
 * 2 per SIMD at B 32, T 512): partial softmaxes / partial dQ sums are merged by fixed-order kernels, results are
 * deterministic.  ws: omr_attn_workspace_floats(B, H, T, S, head_dim, causal, backward) floats (0 = this shape is not split;
 * NULL / 0 is then fine).  Same arguments and semantics as omr_attn_fwd / omr_attn_bwd otherwise. */
long omr_attn_workspace_floats(int B, int H, int T, int S, int head_dim, int causal, int backward);
int omr_attn_fwd_ws(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                    long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, int causal, int window,
                    const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p, unsigned long long seed,
                    const unsigned long long* drop_words, float* ws, long ws_floats, void* stream);
int omr_attn_bwd_ws(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta_ws,
                    void* dq, void* dk, void* dv, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, long bsq,
                    long bsk, long bsv, long bso, long bsdo, long bsdq, long bsdk, long bsdv, int B, int H, int T, int S, int head_dim,
                    int causal, int window, const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p,
                    unsigned long long seed, const unsigned long long* drop_words, float* ws, long ws_floats, void* stream);
/* omr_attn_fwd for ONE block of at most 32 query rows (KV-cached decode: T = 1) with the keys split over workgroups of 256 keys
 * (partial softmaxes merged by a second small kernel): a single query row would otherwise keep each (batch, head) on one
 * workgroup walking all S keys.  No causal / block masks (a decode step sees every cached key), no dropout.
 * split_ws: omr_attn_split_workspace_floats(...) floats of device scratch (0 = the shape needs no split). */
long omr_attn_split_workspace_floats(int B, int H, int T, int S, int head_dim);
int omr_attn_fwd_split(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv, long ldo,
                       long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, const float* key_bias,
                       float* split_ws, long split_ws_floats, void* stream);
/* omr_attn_fwd_split without its merge pass (decode: the out-projection merges while it loads its input rows).  *nsplit > 1:
 * `o` is not written and split_ws holds, for (b, h, split j), head_dim un-normalised outputs + the running max (log2 domain)
 * + the sum at ((b*H + h)*nsplit + j)*T*(head_dim + 2); *nsplit = 1: `o` holds the finished rows. */
int omr_attn_fwd_split_partials(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, long ldq, long ldk, long ldv,
                                long ldo, long bsq, long bsk, long bsv, long bso, int B, int H, int T, int S, int head_dim, float* split_ws,
                                long split_ws_floats, int* nsplit, void* stream);
/* Test / debug entry: the attention-probability dropout keep-mask (1 = kept) for (seed, dropout_p) -- the function
 * omr_attn_dropout_words packs -- one byte per score, mask[B][H][T][S].  Lets a checker inject the very same mask into a CPU
 * restatement of nn.MultiheadAttention's dropout (tests/test_dropout_parity_gpu.py). */
int omr_attn_dropout_mask(unsigned char* mask, int B, int H, int T, int S, float dropout_p, unsigned long long seed, void* stream);
int omr_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta_ws,
                 void* dq, void* dk, void* dv, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, long bsq,
                 long bsk, long bsv, long bso, long bsdo, long bsdq, long bsdk, long bsdv, int B, int H, int T, int S, int head_dim,
                 int causal, int window, const float* key_bias, const int* blk_lq, const int* blk_lkv, float dropout_p,
                 unsigned long long seed, const unsigned long long* drop_words, void* stream);

/* ---- KV-cached greedy decoding: one host call per run of tokens ------------------------------------------------------ */
/* Reference: the autoregressive loop of Transformer.validation_step / get_pred_seq_and_pred_prob_seq (model.py:182-193,
 * 247-260) -- embed the last token, run every decoder layer, project to the vocabulary, argmax, repeat.  omr_decode_steps
 * runs n_steps positions t0 .. t0+n_steps-1 for B independent rows without returning to the caller in between: per position
 * it appends the self-attention K|V to the cache, attends over the cached keys (banded by `window`, decoder.py:213-214) and
 * over the cross-attention K|V projected once per input, and (out_tokens != NULL) writes the argmax token [step][B] (and its
 * fp32 logit, out_top1, nullable) and feeds it to the next position through `tokens` (device int64 [B], updated in place).
 * out_tokens == NULL: one position only, no token pick (beam search / late fusion choose the token themselves).
 * last_logits (nullable): fp32 [B][ldv] logits of the last position.  All pointers are device pointers except `layer_w`.
 * layer_w: HOST array [L][OMR_DECODE_LAYER_PTRS] of device pointers, per layer in this order (torch parameter names):
 *   self_attn.in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, norm1.weight, norm1.bias,
 *   multihead_attn.in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, norm2.weight, norm2.bias,
 *   linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm3.weight, norm3.bias
 * (matrices in `dtype`, biases / LayerNorm vectors fp32).  self_kv [L][B][max_len][2d]; cross_kv: layer l's K|V rows of
 * sample b, key s at cross_kv + b*cross_bs + s*cross_ld + l*2d (elements; cross_bs = 0 shares one memory between rows). */
#define OMR_DECODE_LAYER_PTRS 18
/* fp8 != 0 (BASELINE config 5 "fp8 MFMA weights", an extension): the linears run as omr_quantize_rows_fp8 + omr_gemm_fp8 on
 * weights quantised once per output row.  layer_w8 / layer_s8: HOST arrays [L][OMR_DECODE_LAYER_FP8] of device pointers to the
 * e4m3 codes [rows][in] and fp32 row scales of, in order: self_attn.in_proj_weight, self_attn.out_proj.weight,
 * multihead_attn.in_proj_weight, multihead_attn.out_proj.weight, linear1.weight, linear2.weight; head_w8 / head_s8: the head.
 * Biases, LayerNorm, embedding, attention and the cross-attention K|V (projected once at init) stay in `dtype`. */
#define OMR_DECODE_LAYER_FP8 6
typedef struct omr_decode_desc {
    int dtype, B, L, d, nhead, ff, V, ldv, max_len, S, window, fp8;
    const void* emb; const float* pe; const void* const* layer_w; const void* head_w; const float* head_b;
    void* self_kv; const void* cross_kv; long cross_ld, cross_bs; void* ws; long ws_bytes;
    const unsigned char* const* layer_w8; const float* const* layer_s8; const unsigned char* head_w8; const float* head_s8;
} omr_decode_desc;
long omr_decode_workspace_bytes(const omr_decode_desc* desc);
/* One linear layer of a decode position, y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n]) for a handful of rows (M = batch rows
 * of ONE position), with the element-wise kernel that would precede it folded into the loading of x (`pro`):
 *   0  x = rows of `x` (ld ldx)
 *   1  x = LayerNorm(y_prev + res) * gamma + beta   (nn.TransformerDecoderLayer norm1/2/3, decoder.py:86-95; y_prev = `x`)
 *   2  x = embedding[tokens[m]] + pe_row            (decoder.py:124 at one position)
 *   3  x = the merged key-split attention partials of omr_attn_fwd_split_partials (part, nsplit, H, hd; K = H * hd)
 * Prologues 1 and 2 also store the rows they build to xn_out [M][K] (the residual input of the next sub-layer).  Columns
 * [0, n0) go to out0 (ld0), the rest to out1 (ld1) -- a q | k|v projection writes its k|v part straight into the cache row;
 * out32 (nullable, ld32): fp32 copy of the value rounded to `dtype` (the vocabulary head's logits).  Each (row, column) is
 * a fixed-order fp32 sum that does not depend on M: a batched step reproduces the single-row step to the bit.
 * amax_idx (nullable): also return, per row, the first index of the largest rounded output and its value (the greedy pick of
 * model.py:187,253): every workgroup leaves its 16-column candidate in amax_part (>= 2 * M * ceil(N/16) floats of scratch) and
 * a second, one-wave-per-row launch reduces them -- cheaper than a pass over the N logits.
 * w8 (nullable; BASELINE config 5 "fp8 MFMA weights", an extension): the weight rows as OCP e4m3 codes [N][K] with one fp32
 * scale per output row (omr_quantize_rows_fp8 of the matrix); the codes are dequantised as they are loaded -- half the weight
 * bytes of a bf16 row, the dominant traffic of a decode position -- the products are summed in fp32 against the same input
 * rows and the row scale multiplies the finished sum.  `w` is then unused.
 * Requires K a multiple of 128 (bf16 / fp8 weights) / 64 (fp32) columns, K <= 2048, 16-byte aligned weight rows. */
typedef struct omr_decode_linear_args {
    int dtype, pro, M, N, K, relu, n0, nsplit, H, hd, vocab, pad_;
    float eps, pad2_;
    const void* x; long ldx; const void* res; long ldres; const float* gamma; const float* beta; void* xn_out;
    const long* tokens; const void* emb; const float* pe_row; const float* part;
    const void* w; const float* bias; void* out0; long ld0; void* out1; long ld1; float* out32; long ld32;
    long* amax_idx; float* amax_val; float* amax_part;
    const unsigned char* w8; const float* w8_scale;
} omr_decode_linear_args;
int omr_decode_linear(const omr_decode_linear_args* args, void* stream);
int omr_decode_steps(const omr_decode_desc* desc, long* tokens, int t0, int n_steps, long* out_tokens, float* out_top1, float* last_logits,
                     void* stream);
/* Weighted late fusion (src/multimodal/weighted_multimodal/test.py:21-70) without a host round trip per token: positions
 * t0 .. t0+n_steps-1 of TWO models (descriptors with B = 1 and the same vocabulary) in lock-step; per position
 * argmax(alpha * softmax(logits_a) + (1 - alpha) * softmax(logits_b)) (omr_weighted_argmax) is written to out_tokens[s]
 * (and its mixed probability to out_prob[s], nullable) and fed to both models' next position through `tokens` (device,
 * 1 entry: in = the token of position t0).  logits_a / logits_b: fp32 scratch of ldv entries each. */
int omr_weighted_decode_steps(const omr_decode_desc* desc_a, const omr_decode_desc* desc_b, float alpha, long* tokens, int t0, int n_steps,
                              long* out_tokens, float* out_prob, float* logits_a, float* logits_b, void* stream);

/* ---- loss ------------------------------------------------------------------------------------------------ */
/* CrossEntropyLoss(ignore_index=pad) (model.py:109,166) on row-major logits [M][ldv]; acc2 = {sum, count} (fp64). */
int omr_ce_fwd(int dtype, const void* logits, const long* target, float* lse, double* acc2, float* loss_out, long M, int V, long ldv,
               int pad_idx, void* stream);
/* grad_out (nullable): device pointer to the upstream scalar gradient dL/dloss (multiplies grad_scale). */
int omr_ce_bwd(int dtype, const void* logits, const long* target, const float* lse, const double* acc2, void* dlogits, long M, int V,
               long ldv, int pad_idx, float grad_scale, const float* grad_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
