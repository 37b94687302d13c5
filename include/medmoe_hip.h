/* medmoe_hip.h - C ABI of the MI355X (gfx950) MedMoE hot-path library libmedmoe_hip.so.
 *
 * The reference (shivangchopra11/MedMoE) has no native plugin interface: its boundary is Python
 * object construction via Hydra `_target_` strings.  These entry points
 * are the operator layer a maintainer binds (ctypes) underneath the
 * reference's own modules; each comment names the reference code (file:line under
 * /root/reference/src or .../components) the call replaces.
 *
 * Conventions: plain device pointers + sizes, no allocation, no host sync; every call enqueues
 * on `stream` and returns 0, or -1 bad argument, -2 unsupported shape, -3 launch error.  The one
 * piece of process-global state is medmoe_set_option (kernel-selection switches for tests and
 * measurements; the defaults are the fastest measured and the product path never changes them).  bf16 data is raw uint16 bits; "f32" is IEEE float.  All buffers are
 * caller-owned device memory; inputs are never modified unless documented (in-place residual).
 */
#ifndef MEDMOE_HIP_H
#define MEDMOE_HIP_H
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

/* F.scaled_dot_product_attention over the fused QKV buffer (multi_head_attention.py:62-78) */
int medmoe_attn_fwd(const void* qkv, void* out, float* lse, const unsigned char* key_mask, int B, int N, int H, int head_dim, hipStream_t stream);

/* backward of the same (dQ, dK, dV written into a [B*N,3D] buffer) */
int medmoe_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const unsigned char* key_mask, void* dqkv, float* delta, int B, int N, int H, int head_dim, hipStream_t stream);

/* ViT patch extraction in Conv2d weight order (build-defined front-end) */
int medmoe_patchify(const void* img, void* out, int B, int C, int H, int W, int patch, int in_f32, hipStream_t stream);
/* the same with a row pitch ld >= C*patch*patch for `out` (rows padded to the GEMM k-step, e.g. patch 14: 588 -> 640) */
int medmoe_patchify_ld(const void* img, void* out, int B, int C, int H, int W, int patch, int in_f32, int ld, hipStream_t stream);

/* CLS + position embedding fill (build-defined front-end) */
int medmoe_init_tokens(void* x, const float* cls, const float* pos, int B, int Nt, int D, hipStream_t stream);

/* gradient of the position / CLS embeddings */
int medmoe_pos_cls_grad(const void* dx, float* dpos, float* dcls, int B, int Nt, int D, hipStream_t stream);

/* BERT-style embedding gather + LayerNorm (text tower front-end, text_encoder.py:94) */
int medmoe_text_embed_ln(const int* ids, const int* type_ids, const float* word, const float* pos, const float* type, const float* gamma, const float* beta, void* out, int B, int T, int D, int vocab, float eps, hipStream_t stream);

/* last-4-layer sum + word-piece segment-sum + sentence mean (text_encoder.py:32-90,97-117) */
int medmoe_text_aggregate(const void* h0, const void* h1, const void* h2, const void* h3, int n_layers, const int* seg, void* word_bf16, float* word_f32, float* sent, int B, int T, int D, hipStream_t stream);

/* bf16 MFMA GEMM C = epi(A B^T): every nn.Linear / Conv1d(k=1) forward and dgrad on the path (multi_head_attention.py:35-36,61,80; mlp.py:55-64; swin.py:18-30,40-41,62).
   epi: 0 none, 1 GELU (aux <- pre-activation), 2 ReLU, 3 x GELU'(aux), 4 x ReLU'(aux), 5 GELU (aux <- GELU'(pre-activation)), 6 x aux;
   order: alpha*acc + bias -> activation -> + residual -> x derivative factor */
int medmoe_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, const void* residual, int ldr, void* aux, int ldaux, const int* a_rowmap, const int* c_rowmap, const int* tiles, const int* tile_count, int max_tiles, long long strideB, long long strideBias, float alpha, int epi, int out_f32, int col_perm, hipStream_t stream);

/* medmoe_gemm_nt for grouped / row-mapped operands on the 256x256 kernel: `tiles` holds 256-row tiles (the second table
   medmoe_dispatch writes); bf16 C, alpha = 1, no col_perm.  Expert Linear layers swin.py:32-60 and the local-loss Gram gradient. */
int medmoe_gemm_nt_tiles256(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, const void* residual, int ldr, void* aux, int ldaux, const int* a_rowmap, const int* c_rowmap, const int* tiles, const int* tile_count, int max_tiles, long long strideB, long long strideBias, float alpha, int epi, int out_f32, int col_perm, hipStream_t stream);

/* wgrad dW += G^T X (+ bias grad), replaces autograd of the same Linear layers */
int medmoe_gemm_tn(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw, float* db, int M, int Nn, int Kk, const int* x_rowmap, const int* g_rowmap, const int* row_off, int n_groups, long long strideW, long long strideDb, int nsplit, hipStream_t stream);
/* the plain wgrad (no row maps, no groups) in two stages when the shape allows (Nn, Kk multiples of 256, M % 32 == 0, M >= 4096): the
   workgroups STORE their partial 256 x 256 tiles into `scratch` (scratch_floats >= tiles x row ranges x 65536; owned by the caller, one per
   stream) and a second kernel sums them into dW in a fixed order - no fp32 atomics on dW (deterministic, and ~20 us cheaper per launch than
   64 MB of memory-side atomics).  Any other shape, or scratch too small / null: medmoe_gemm_tn with nsplit 16 */
int medmoe_gemm_tn_staged(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw, float* db, int M, int Nn, int Kk, float* scratch, long long scratch_floats, hipStream_t stream);

/* cross entropy over rows/columns of a similarity matrix + gradient (losses.py:789-794,1017-1021,582-584) */
int medmoe_ce_strided(const float* X, float* dX, int rows, int cols, long long rs, long long cs, int label_off, float xscale, float w, int accumulate, float* loss_acc, hipStream_t stream);

/* Soft-GLoRIA head over the rows (or, with swapped strides, the columns) of a similarity matrix: positives soft > t1, negatives soft <= t2,
   per positive -log_softmax([x_pos, x_negatives])[0] / (1 + #negatives), averaged over positives, weighted by w; soft is [rows, cols] fp32
   row-major (losses.py:861-883 SoftGLORIAGlobalContrastiveLoss, :1180-1208 SoftGLORIALocalContrastiveLoss, softXEnt :796-803) */
int medmoe_soft_xent_strided(const float* X, float* dX, const float* soft, int rows, int cols, long long rs, long long cs, float xscale, float t1, float t2, float w, int accumulate, float* loss_acc, hipStream_t stream);

/* HardNegativeContrastiveLoss head (losses.py:885-927, nmax = 1) over the rows / columns of a cosine matrix: relu(hardest negative + margin -
   diagonal) summed with weight w, and its sub-gradient; the diagonal competes as its own negative (scores - 2 diag(diag), :903) */
int medmoe_hardneg_strided(const float* X, float* dX, int rows, int cols, long long rs, long long cs, float margin, float w, int accumulate, float* loss_acc, hipStream_t stream);

/* row L2 norms (losses.py:778-779) */
int medmoe_rownorm(const float* x, float* n, int rows, int D, hipStream_t stream);

/* cosine normalisation with eps clamp (losses.py:781-785) */
int medmoe_cos_scale(float* S, const float* na, const float* nb, int M, int N, float eps, hipStream_t stream);

/* backward of cos_scale */
int medmoe_cos_scale_bwd(float* dC, const float* C, const float* na, const float* nb, float* ca, float* cb, int M, int N, float eps, hipStream_t stream);

/* dst += coef[row] * src (norm-gradient term of the cosine) */
int medmoe_add_rowscaled(float* dst, const float* src, const float* coef, int rows, int D, hipStream_t stream);

/* word norms + transposed word matrix for the local loss (losses.py:690-695,985) */
int medmoe_words_prep(const void* words, float* wn, void* wT, int Bc, int T, int Tp, int D, hipStream_t stream);

/* fp32 padded -> bf16 dense copy of the local-loss context gradient */
int medmoe_unpad_cast(const float* src, void* dst, int B, int HW, int HWp, int D, hipStream_t stream);

/* GLoRIA local loss for one (image, caption) pair per workgroup, fwd / recomputing bwd (losses.py:979-1012, attention_fn :698-736) */
int medmoe_local_pair(const void* ctx, const void* words, const void* gmp, const float* wnorm, const int* cap_lens, const float* gsim, float* sim, void* dS, void* A, void* U, float* att, const void* a1_pre, const float* lse_pre, int B, int Bc, int HW, int T, int D, float temp1, float temp2, float eps, int backward, hipStream_t stream);

/* all word-region scores as one tiled GEMM with the word-softmax fused: A1 (bf16) + row log-sum-exp (losses.py:713-716) */
int medmoe_local_scores(const void* ctx, const void* words, const int* cap_lens, void* a1, float* lse, int B, int Bc, int HW, int T, int D, hipStream_t stream);

/* lean per-pair kernel on the local_scores tiles: sim + dS/A/U in one pass (losses.py:724-1012 after the word softmax) */
int medmoe_local_pair2(void* a1_io, const float* lse_pre, const void* gmp, const float* wnorm, const int* cap_lens, const float* gsim, float* sim, void* dS, void* U, float* att, int B, int Bc, int HW, int T, float temp1, float temp2, float eps, hipStream_t stream);

/* per-(image,caption)-block scaling of the local-loss gradient matrices by dL/dsim (single-pass mode) */
int medmoe_scale_blocks(void* X0, void* X1, const float* g, int B, int Bc, int HWp, int Tp, hipStream_t stream);

/* RAGGED local-loss layout: captions are grouped into length classes (<= 16, 32, ... words); class c holds its members
   side by side, 16*c columns each, so the B x B pair matrices are sum_i pad16(len_i) columns wide instead of B * Tp.
   Same math as the uniform entry points above (losses.py:690-695,713-716,724-1012), one launch per class.
   medmoe_words_prep_ragged with wT = null writes the word norms only (the transposed pair matrices take the words row-major). */
int medmoe_words_prep_ragged(const void* words, float* wn, void* wT, int Bc, int T, int Tp, int D, const int* col_of_cap, const int* tp_of_cap, long long ldw, hipStream_t stream);
int medmoe_local_scores_ragged(const void* ctx, const void* words, const int* cap_lens, void* a1, float* lse, int B, int Bc, int HW, int T, int D, const int* cap_list, int n_cap, int ntt, long long col_base, long long ldp, hipStream_t stream);
int medmoe_local_pair2_ragged(void* a1_io, const float* lse_pre, const void* gmp, const float* wnorm, const int* cap_lens, const float* gsim, float* sim, void* dS, void* U, int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap, int ntt, long long col_base, long long ldp, hipStream_t stream);
int medmoe_scale_blocks_ragged(void* X0, void* X1, const float* g, int B, int Bc, int HWp, const int* cap_of_chunk, long long ld, hipStream_t stream);

/* TRANSPOSED pair matrices (element (row, image b, region hw < pw) at row * ld + b * bstride + hw, row = row_base + j * 16 ntt + t; the
   engine uses the image-major form ld = pw = 32 ceil(HW / 32), bstride = rows * pw: one (image, caption, word tile) unit is 16 x 448 contiguous bytes): the pair
   stage of the local loss with one wave per (image, caption, 16-word tile) (losses.py:979-1012 after the word softmax).
   lp: fp16 LOG2-probabilities of the word softmax (medmoe_local_scores_t); lse: [B][Bc][HWp]; gm: [B][GR][GR] bf16 Gram matrices ctx_b ctx_b^T, GR = 32 ceil(HW / 32),
   zero outside [HW][HW]; stats: [B][stat_rows][2] fp32.  dS == NULL: the forward launch (writes sim, A, stats, att of the matching pairs).
   Otherwise the backward launch over the same class: reads lp, A, stats, sim and writes dS (may be lp itself) and the Gram-matrix
   gradient's operand in either form: U = d2 * A as a matrix (U != NULL; d2 = 2 dL/dn2 per word row) and / or the row weights d2 alone,
   fp32 [B][stat_rows] (d2 != NULL: medmoe_gemm_tn_gram multiplies the A rows by them); everything scaled by gsim = dL/dsim (NULL: 1). */
int medmoe_local_pair3(const void* lp, void* dS, void* A, void* U, const float* lse, const void* gm, const float* wnorm, const int* cap_lens, const float* gsim, float* sim, float* att, float* stats, long long stat_rows, int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap, int ntt, long long row_base, long long ld, long long bstride, int pw, float* d2, hipStream_t stream);
/* the backward launch of medmoe_local_pair3 that ALSO writes dwn [B][stat_rows] fp32: per (image, caption word) the coefficient of w_t in
   d loss / d w_t that comes from the word's own norm in the cosine, -dL/dcos * cos / ||w_t||^2 (losses.py:690-695, 1002); the caller sums it over
   the images.  The part of the word gradient that goes through the scores is dS^T ctx (one NT GEMM over the row-major pair matrix). */
int medmoe_local_pair3_wgrad(const void* lp, void* dS, void* A, void* U, const float* lse, const void* gm, const float* wnorm, const int* cap_lens, const float* gsim, float* sim, float* att, float* stats, long long stat_rows, int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap, int ntt, long long row_base, long long ld, long long bstride, int pw, float* d2, float* dwn, hipStream_t stream);
/* the score GEMM of one caption length class with the word softmax fused (losses.py:713-716), TRANSPOSED output for medmoe_local_pair3:
   lpT[(row_base + j * 16 ntt + t) * ld + b * bstride + hw] = fp16 ((S - lse) / ln 2), lse[b][caption][hw] fp32.  HW % 4 == 0, D % 32 == 0, D >= 128. */
int medmoe_local_scores_t(const void* ctx, const void* words, const int* cap_lens, void* lpT, float* lse, int B, int Bc, int HW, int T, int D, const int* cap_list, int n_cap, int ntt, long long row_base, long long ld, long long bstride, hipStream_t stream);
/* dW[g][Nn][Kk] += G[:, g*gcol_stride + (0..Nn)]^T X[:, g*xcol_stride + (0..Kk)], g < n_groups, over ALL M rows (fp32 atomics: zero dW first).
   bf16 operands, M % 32 == 0, rows addressed with 64-bit bases (M * ld may exceed 4 GB).  The two wgrad-shaped GEMMs of the transposed
   local loss: d ctx = dS^T words (one group) and dGm_b = U_b^T A_b (one group per image).  g_chunk_w > 0: G's columns are stored in chunks of
   g_chunk_w columns, chunk j at G + j * g_chunk_stride (image-major pair matrices as one [M][B * HWp] operand). */
int medmoe_gemm_tn_cols(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw, int M, int Nn, int Kk, int n_groups, long long gcol_stride, long long xcol_stride, long long strideW, int g_chunk_w, long long g_chunk_stride, hipStream_t stream);

/* weighted Gram products dW[g] += A_g^T diag(w_g) A_g over the column blocks A_g = A + g*col_stride of one [M][lda] bf16 matrix; w_g[m] =
   w[g*w_gstride + m*w_ld] fp32.  The local loss' dGm_b = sum over caption words of d2 * a a^T (backward of the weighted-context norm in
   attention_fn / cosine_similarity, losses.py:690-736) without a stored d2 * A matrix.  fp32 atomics into dW (zero it first). */
int medmoe_gemm_tn_gram(const void* A, int lda, const float* w, long long w_gstride, int w_ld, float* dW, int ldw, int M, int Nn, int n_groups, long long col_stride, long long strideW, hipStream_t stream);
/* tests: caption chunks per image of medmoe_local_pair3 (0 = automatic) */
int medmoe_local_pair3_chunks(int n);
/* 1: medmoe_local_pair3 has an instantiation for (HW regions, T words) */
int medmoe_local_pair3_supported(int HW, int T);

/* word-piece segment map + caption lengths on the device (text_encoder.py:45-76 token loop; cap_lens of medmoe_module.py:221-223): seg[b,t] = word
 * index of token t or -1 (dropped), cap[b] = #words not starting with '[' + 1; is_cont / starts_bracket: one byte per vocabulary id; ids int64
 * (ids64 != 0) or int32 */
int medmoe_segment_map(const void* ids, int ids64, const unsigned char* is_cont, const unsigned char* starts_bracket, int* seg, int* cap, int B, int T, int vocab, int sep_id, hipStream_t stream);

/* trainable text tower (reference freeze_bert: false, text_encoder.py:27-30): backward of the word-piece aggregation (text_encoder.py:32-117) -
 * every selected layer's hidden state receives dH[b,t] = d_word[b, seg[b,t]] + d_sent[b] / T (kept tokens), 0 (dropped) - and of the embedding
 * front-end y = LN(word[id] + pos[t] + type[tt]): dx (fp32 [B*T, D], for the position / token-type tables), fp32 atomics into the word table's
 * gradient rows, dgamma / dbeta accumulated */
int medmoe_text_aggregate_bwd(const float* d_word, const float* d_sent, const int* seg, void* dH, int B, int T, int D, hipStream_t stream);
int medmoe_text_embed_ln_bwd(const int* ids, const int* type_ids, const float* word, const float* pos, const float* type, const float* gamma, const void* dy, float* dx, float* dgamma, float* dbeta, float* g_word, int B, int T, int D, int vocab, float eps, hipStream_t stream);

/* Variable-length text batches (the frozen text tower on the tokens with attention mask 1 only, packed in (caption, position) order;
   reference text_encoder.py:97-117 computes all B x T positions): medmoe_text_pack builds tok_row[b*T+t] (packed row or -1),
   src_of_row[r] (= b*T+t), seq_off[B+1] and count[1] on the device (B <= 1024); the *_packed / *_rows / *_varlen entry points take them, so no
   row count ever travels to the host. */
int medmoe_text_pack(const unsigned char* mask, int* tok_row, int* src_of_row, int* seq_off, int* count, int B, int T, hipStream_t stream);
int medmoe_text_embed_ln_packed(const int* ids, const int* type_ids, const float* word, const float* pos, const float* type, const float* gamma, const float* beta, void* out, int B, int T, int D, int vocab, float eps, const int* src_of_row, const int* count, hipStream_t stream);
int medmoe_text_aggregate_packed(const void* h0, const void* h1, const void* h2, const void* h3, int n_layers, const int* seg, const int* tok_row, void* word_bf16, float* word_f32, float* sent, int B, int T, int D, hipStream_t stream);
int medmoe_layernorm_fwd_rows(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int D, float eps, int out_f32, const int* rows_dev, hipStream_t stream);
int medmoe_attn_fwd_varlen(const void* qkv, void* out, float* lse, const int* seq_off, int B, int Nmax, int H, int head_dim, hipStream_t stream);
/* medmoe_gemm_nt (plain operands) on the first *m_dev rows (device int, <= M) */
int medmoe_gemm_nt_rows(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, const void* residual, int ldr, void* aux, int ldaux, float alpha, int epi, int out_f32, const int* m_dev, hipStream_t stream);

/* Swin-T tower pieces (reference swin.py:119-149 = HF SwinModel, transformers modeling_swin.py):
   (shifted-)window attention on token-major qkv [B*H*W, 3C] (q | k | v, head h at columns h*32; 7x7 windows, head dim 32): the cyclic shift,
   window partition and their inverses are row arithmetic inside the kernels.  bias: [heads][64][64] fp32 = relative position bias of the
   49 x 49 token pairs, key columns >= 49 = -30000, rest 0; lse: [B*(H/7)*(W/7)*heads][64] fp32.  bwd: dqkv fully written; dbias (or NULL):
   [B*(H/7)*(W/7)*heads][64][64] fp32, dS of every (image, window, head) - its sum over images and windows is the bias gradient. */
int medmoe_win_attn_fwd(const void* qkv, const float* bias, void* out, float* lse, int B, int H, int W, int C, int heads, int shift, hipStream_t stream);
int medmoe_win_attn_bwd(const void* qkv, const float* bias, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int H, int W, int C, int heads, int shift, hipStream_t stream);
/* stochastic depth: out[b] = res[b] + scale[b] * branch[b] over per_sample bf16 elements per image (res == NULL: out = scale[b] * branch[b]) */
int medmoe_drop_path(const void* branch, const void* res, const float* scale, void* out, int B, long long per_sample, hipStream_t stream);
/* SwinPatchMerging's 2x2 concat: y[b, Y, X, q*C + c] = x[b, 2Y + (q & 1), 2X + (q >> 1), c] (scatter = 0) or its transpose (scatter = 1) */
int medmoe_patch_merge(const void* src, void* dst, int B, int H, int W, int C, int scatter, hipStream_t stream);

/* image preprocessing on the device: B uint8 HWC images (device pointers src_ptrs[b], sizes src_hw[2b], src_hw[2b+1]) ->
   bilinear resize (half-pixel centres) -> x rescale -> (x - mean) / std -> bf16 [B,3,Ho,Wo].  Replaces the per-step CPU
   AutoImageProcessor call of swin.py:131.  mean3 / std3 are HOST arrays. */
int medmoe_preprocess(const void* const* src_ptrs, const int* src_hw, void* dst, int B, int Ho, int Wo, float rescale, const float* mean3, const float* std3, hipStream_t stream);
/* the reference's own filter: Pillow BICUBIC resize on uint8 (HF image processor of swin-tiny, swin.py:131), horizontal then vertical
   pass in 22-bit fixed point, then x rescale, (x - mean) / std -> bf16 [B,3,Ho,Wo].  meta[b][9] = {Hs, Ws, ksize_h, ksize_v, offsets of the
   horizontal bounds / weights and vertical bounds / weights in coef, first row of image b in tmp}; tmp: uint8 [total_src_rows][Wo][3];
   u8_out (may be NULL): the resized uint8 image [B][Ho][Wo][3] before normalisation */
int medmoe_preprocess_bicubic(const void* const* src_ptrs, const long long* meta, const int* coef, void* tmp, void* dst, void* u8_out, int B, int Ho, int Wo, long long total_src_rows, float rescale, const float* mean3, const float* std3, hipStream_t stream);

/* padded geometry (HWp, Tp, Gm row width) the local-loss kernels were instantiated for */
int medmoe_local_geometry(int HW, int T, int* HWp, int* Tp, int* GW);
/* 1: the LDS-tiled pair kernels exist for (HW regions, T words); 0: the generic path below (any HW, T <= 80) */
int medmoe_local_fast_path(int HW, int T);

/* generic-geometry GLoRIA local loss (losses.py:698-736, 985-1012 as grouped GEMMs + these elementwise kernels over the uniform pair
   matrices [B*HWp, Bc*Tp]): region-softmax A from the word log-probabilities; cosine / exp / log-sum per pair from wctx = A^T ctx;
   d wctx after the CE over sim; dS from dA (in place); sum of two padded fp32 gradients -> bf16 */
int medmoe_local_gen_fwd_a(const void* lp, const int* cap_lens, void* A, int B, int Bc, int HW, int HWp, int T, int Tp, float temp1, long long ldp, hipStream_t stream);
int medmoe_local_gen_cos(const float* wc, const void* words, const float* wnorm, const int* cap_lens, float* sim, float* stats, float* sume, int B, int Bc, int T, int Tp, int D, float temp2, float eps, long long Kp, hipStream_t stream);
int medmoe_local_gen_dwctx(const float* wc, const void* words, const float* wnorm, const int* cap_lens, const float* gsim, const float* stats, const float* sume, void* dwc, int B, int Bc, int T, int Tp, int D, float temp2, float eps, long long Kp, hipStream_t stream);
int medmoe_local_gen_bwd_s(const void* lp, const void* A, void* dA_io, const int* cap_lens, int B, int Bc, int HW, int HWp, int T, int Tp, float temp1, long long ldp, hipStream_t stream);
int medmoe_unpad_cast2(const float* src, const float* src2, void* dst, int B, int HW, int HWp, int D, hipStream_t stream);

/* mean over tokens: router input (swin.py:137) and global feature (swin.py:112) */
int medmoe_mean_tokens(const void* x, float* out, int B, int Nt, int D, int t0, int cnt, hipStream_t stream);

/* backward of mean_tokens */
int medmoe_broadcast_tokens(const float* g, void* dy, int B, int Nt, int D, int t0, int cnt, float scale, hipStream_t stream);

/* MoE router softmax + top-k (swin.py:88-92,98-100); fixed fp32 order => bit-exact indices */
int medmoe_router_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* probs, int* idx, float* gates, int B, int Dv, int Hd, int E, int k, hipStream_t stream);

/* CE on router probabilities (medmoe_module.py:235-237) + gate gradients -> dlogits, dh */
int medmoe_router_bwd(const float* probs, const float* h, const float* w2, const int* idx, const float* dgates, const int* labels, const float* dprobs_ext, float ce_scale, float* dlogits, float* dh, float* loss_acc, int B, int Hd, int E, int k, hipStream_t stream);

/* small strided fp32 GEMM (router wgrad/dgrad, global-loss similarity and its gradients) */
int medmoe_sgemm(const float* A, const float* Bm, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn, long long ldc, float alpha, float beta, hipStream_t stream);

/* sort (sample,choice) pairs by expert; row maps + tile table for the grouped expert GEMMs (replaces the dense all-experts + gather of swin.py:105-108) */
int medmoe_dispatch(const int* idx, int B, int k, int E, int P, int Nt, int* slot_of, int* item_of_slot, int* expert_of_slot, int* row_off, int* tiles, int* tile_count, int max_tiles, int* rowmap, hipStream_t stream);

/* Expert scale attention + weighted sum (swin.py:62-80) */
int medmoe_scale_attn_fwd(const void* G, const void* H1, const float* w2, const float* b2, const int* expert_of_slot, int P, void* out, float* wts, int R, int Do, int Dh, hipStream_t stream);

/* combine selected experts into img_l (swin.py:106-113) */
int medmoe_combine_fwd(const void* expert_out, const int* slot_of, const float* gates, void* img_l, int B, int k, int P, int Do, hipStream_t stream);

/* backward of combine + scale attention */
int medmoe_scale_attn_bwd(const void* d_img_l, const float* d_img_g, const void* G, const void* H1, const float* wts, const float* w2, const void* expert_out, const int* expert_of_slot, const int* item_of_slot, const float* gates, int k, int P, void* dG, void* dH1, float* dw2, float* db2, float* dgate, int R, int Do, int Dh, hipStream_t stream);

/* pyramid-geometry experts: F.interpolate(size = Pout, mode = 'linear', align_corners = False) along the token axis (swin.py:42),
   bf16 [n, Pin, D] -> [n, Pout, D]; backward in gather form, optionally times ReLU'(relu_aux) of the interpolated projection's source */
int medmoe_lerp_tokens_fwd(const void* x, void* y, int n, int Pin, int Pout, int D, hipStream_t stream);
int medmoe_lerp_tokens_bwd(const void* dy, const void* relu_aux, void* dx, int n, int Pin, int Pout, int D, hipStream_t stream);
/* the same with the incoming gradient given as two summands dy + dy2 (Pin == Pout: dx = (dy + dy2) ReLU'(relu_aux)) */
int medmoe_lerp_tokens_bwd2(const void* dy, const void* dy2, const void* relu_aux, void* dx, int n, int Pin, int Pout, int D, hipStream_t stream);

/* scatter-add stage-feature gradients into the ViT residual-stream gradient */
int medmoe_stage_grad_add(const void* dF, const int* slot_of, void* dx, int B, int k, int P, int Nt, int D, hipStream_t stream);

/* Fp32LayerNorm forward (normalizations.py:8-19; transformer.py:78-79) */
int medmoe_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int D, float eps, int out_f32, hipStream_t stream);

/* Fp32LayerNorm backward (+ fused residual-gradient add) */
int medmoe_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, hipStream_t stream);

/* sum of squares of the flat gradient (clip_grad_norm_, pretraining_medmoe.yaml:23) */
int medmoe_sumsq(const float* g, long long n, float* out, hipStream_t stream);
/* the same with a fixed summation order (bit-identical on every data-parallel rank): out[0] = sum g^2; scratch >= 2049 floats, scratch[2048] zero on entry */
int medmoe_sumsq_det(const float* g, long long n, float* out, float* scratch, hipStream_t stream);

/* host scheduling aid, no reference counterpart (the reference trains on one stream under torch autograd): stream `to` waits for everything
   enqueued on `from` so far - how the hand-scheduled backward forks its weight-gradient GEMMs onto a second stream and joins them */
int medmoe_stream_fork(hipStream_t from, hipStream_t to);

/* fused clip + torch.optim.Adam step + bf16 down-cast (med-moe_pretraining.yaml:7-11) */
int medmoe_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, double lr, double beta1, double beta2, double eps, double weight_decay, int step, const float* grad_normsq, float max_norm, float grad_scale, hipStream_t stream);

/* fp32 -> bf16 copy of the master weights */
int medmoe_cast_bf16(const float* src, void* dst, long long n, hipStream_t stream);

/* batched bf16 transposes: W^T copies read by the dgrad GEMMs */
int medmoe_transpose_many(const void* src, void* dst, const long long* table, int n_entries, int max_tiles, hipStream_t stream);

/* kernel-selection switches for tests and measurements (the defaults are the fastest measured):
 *   1 gemm_nt256 on/off   2 256x256 NT tiles on/off   3 256x256 wgrad tiles on/off   4 rows per range of the grouped wgrad (>= 256)
 *   5 largest NT grid (1..256, experiments on fewer CUs)   6 scores512 on/off
 *   7 gemm_nt4w (four-wave NT GEMM; 0 = the eight-wave gemm_nt512)   8 gemm_tn4w (0 = gemm_tn512)
 *   9 fewest rows per M range of the plain wgrad (>= 64, default 2048)
 * Returns MM_ERR_ARG for an unknown key or a value out of range. */
int medmoe_set_option(int key, int value);
/* measurement aid (bench.py labels its per-launch timings with it): the kernel the most recent medmoe_gemm_nt / _rows / _tiles256 call of this
 * process launched - 0 gemm_nt_kernel (128x128 tile), 1 gemm_nt256_kernel, 2 gemm_nt512_kernel, 3 gemm_nt512_kernel GROUPED,
 * 4 gemm_nt4w_kernel, 5 gemm_nt4w_kernel GROUPED; -1 before the first call */
int medmoe_last_gemm_nt_kernel(void);

/* ---- fp8 (OCP e4m3fn) expert weights on the CDNA4 fp8 MFMA (BASELINE.json configs[4]; reference swin.py:18-30 projections) ---- */
/* bf16 rows (gathered through rowmap when given; times colscale[slot_expert[row / rows_per_slot]][k] when given) -> e4m3 rows q[M][K]
   + one dequantisation scale per row s[M] = amax / 448 */
int medmoe_quant_rows_e4m3(const void* x, int ldx, const int* rowmap, const float* colscale, const int* slot_expert, int rows_per_slot, void* q, float* s, int M, int K, hipStream_t stream);
/* fp32 master weights [G][N][K] -> e4m3 q[G][N][K], transposed qT[G][K][N] (may be NULL), scales s[G][N] = amax_k / 448 */
int medmoe_quant_weights_e4m3(const float* w, void* q, void* qT, float* s, int G, int N, int K, hipStream_t stream);
/* grouped C[m][n] = epi(sa[m] * sb[g][n] * sum_k Aq[m][k] Bq[g][n][k] (+ bias[g][n])) over the 128-row tile table {group, m0, m_end, -};
   epi 0 none, 1 ReLU, 2 (. + residual) * (aux > 0); C / residual / aux bf16 with row pitch ldc; sb may be NULL (scales folded into A) */
int medmoe_gemm_fp8_grouped(const void* Aq, const float* sa, const void* Bq, const float* sb, const float* bias, void* C, int ldc, const void* residual, const void* aux, const int* tiles, const int* tile_count, int max_tiles, int N, int K, long long strideB, long long strideSb, long long strideBias, int epi, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDMOE_HIP_H */
