/* phonlp_heads.h -- C-ABI of the MI355X (gfx950) task-head library `libphonlp_heads.so`.
 *
 * Drop-in boundary for PhoNLP's task heads (everything after the PhoBERT first-subword
 * gather).  The reference has no FFI: the seam is Python/torch, so each
 * entry point below names the reference Python call it replaces (paths under
 * /root/reference/phonlp).  The Python host in `phonlp_amd/` binds these with ctypes and
 * registers them as torch custom ops; INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'd / torch ROCm tensor storage) unless the
 *    name ends in `_host`; float = IEEE fp32, row-major, 16-byte aligned where a leading
 *    dimension is given;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue
 *    work, they never synchronise, allocate or free;
 *  - scratch memory is caller-provided: ask `phonlp_*_ws_bytes()` first, pass `ws`/`ws_bytes`;
 *  - return value: PHONLP_OK or a negative PHONLP_ERR_*; `phonlp_last_error()` gives text;
 *  - threads: every entry point may be called from several host threads at once provided the calls do not share
 *    output / workspace memory (error text is thread-local; the per-device launch-once setup is race-tolerant).  The
 *    process-wide SWITCHES are not synchronised -- phonlp_set_gemm_engine, phonlp_set_mst_fast_path and the profiling
 *    calls (phonlp_profile_*): set them before starting worker threads, not while calls are in flight;
 *  - `sentlens[b]` = number of real rows (words, incl. ROOT on the parser path) of sentence b;
 *    rows/tokens at index >= sentlens[b] are padding.
 */
#ifndef PHONLP_HEADS_H
#define PHONLP_HEADS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHONLP_OK 0
#define PHONLP_ERR_SHAPE (-1)       /* invalid shape / alignment / NULL argument */
#define PHONLP_ERR_WORKSPACE (-2)   /* workspace too small */
#define PHONLP_ERR_LAUNCH (-3)      /* HIP reported a launch error */
#define PHONLP_ERR_UNSUPPORTED (-4) /* size beyond what the kernels support (e.g. T > 64) */

/* per-sentence status written by phonlp_mst_one_root */
#define PHONLP_MST_OK 0
#define PHONLP_MST_NO_FINITE_TREE 1 /* ref chuliu_edmonds.py:162-168: the reference asserts here */

/* Padded feature width used by the biaffine kernels for a hidden size H: H real features, one
 * ones-column (ref biaffine.py:63-64), zero padding up to a multiple of 8. */
int phonlp_kp(int H);

int phonlp_version(void);
const char *phonlp_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Dense layer  Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N])    (fp32 MFMA, f32 accumulate)
 * Replaces nn.Linear call sites on the path: upos_hid / upos_clf (annotate_model.py:135-136,
 * 175-176), ner_tag_clf (:149), dep_hid (:186), DeepBiaffineScorer.W1/W2 (biaffine.py:83-85)
 * and PairwiseBilinear stage A (biaffine.py:27-29, with W pre-transposed by
 * phonlp_pack_bilinear).  If `sentlens` != NULL the rows are (b, i) = (m / L, m % L) and padded
 * rows are written as exact zeros (pad_packed_sequence semantics, annotate_model.py:150,187).
 * Requires K % 4 == 0, ldx % 4 == 0, ldw % 4 == 0, 16-byte aligned X and W.
 */
/* GEMM engine used by every matrix kernel of the library (process-wide; default 2, or $PHONLP_GEMM_ENGINE):
 *   0  v_mfma_f32_32x32x2_f32: exact fp32 multiply-add chain;
 *   1  "bf16x6": each fp32 operand split exactly into three bf16 pieces in registers, six
 *      v_mfma_f32_32x32x16_bf16 products, fp32 accumulate -- fp32-equivalent (dropped terms <= 3*2^-24 |a||b|),
 *      about twice as fast.  Same memory layouts, same entry points.
 *   2  "f16x3": as 1 for every entry point that takes fp32 operands; in addition the split-image entry points below
 *      (phonlp_split_rows / phonlp_linear_split) are the engine proper: operands scaled per row by a power of two
 *      and split exactly-to-2^-23 into two f16 pieces ONCE (weights at load, activations in one HBM-bound pass),
 *      three v_mfma_f32_32x32x16_f16 products, fp32 accumulate -- the same fp32-grade error with half the MFMA
 *      work of engine 1.  phonlp_amd/heads.py routes the large layers of annotate_batch through it. */
int phonlp_set_gemm_engine(int engine);
int phonlp_get_gemm_engine(void);

int phonlp_linear(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy,
                  int M, int N, int K, int relu, const int32_t *sentlens, int L, void *stream);

/* Engine 2 dense layer on split images (same nn.Linear call sites as phonlp_linear: biaffine.py:27-29 stage A,
 * biaffine.py:83-85 W1/W2).  Split image of a [rows, K] fp32 matrix = `planes` (phonlp_split_bytes(rows, K) bytes:
 * row pitch 4*K16, K16 = K rounded up to 16; per 8 consecutive k: h[8] f16 then l[8] f16 of the row scaled by a
 * power of two) + inv_scale [rows] fp32 (the inverse of that power of two).  K % 4 == 0, ldx % 4 == 0.
 *   Y[M,N] = act((X . W^T) + bias)   with X, W given as split images; sentlens / L / pad_bias as in
 * phonlp_linear and the folded form described under phonlp_deep_biaffine_arc (pad_bias may be NULL).
 * `kind` only names the launch for the profiler: 0 generic, 1 label stage A, 2 label W1/W2, 3 arc W1/W2. */
size_t phonlp_split_bytes(int rows, int K);
int phonlp_split_rows(const float *X, int ldx, int rows, int K, void *planes, float *inv_scale, void *stream);
int phonlp_linear_split(const void *Xp, const float *x_inv, const void *Wp, const float *w_inv, const float *bias,
                        float *Y, int ldy, int M, int N, int K, int relu, const int32_t *sentlens, int L,
                        const float *pad_bias, int kind, void *stream);
/* The same layer (biaffine.py:83-85 W1/W2) with its result written AS A SPLIT IMAGE (Yp: M rows of 4*N bytes, N % 16 == 0;
 * y_inv [M]) for the next engine-2 product, so that neither the fp32 result nor a split pass over it exist.  The row
 * scales are fixed before the product from a bound on the row's entries: |y[m][n]| <= bound_c * max_k |x[m][k]| +
 * bound_b with bound_c >= sqrt(K) * max_n ||W[n]||_2 and bound_b >= max |bias|, |pad value| (the caller derives both from
 * the weights at load); max_k |x[m][k]| is read off x_inv. */
int phonlp_linear_split_image(const void *Xp, const float *x_inv, const void *Wp, const float *w_inv, const float *bias,
                              void *Yp, float *y_inv, int M, int N, int K, int relu, const int32_t *sentlens, int L,
                              const float *pad_bias, float bound_c, float bound_b, int kind, void *stream);

/* ------------------------------------------------------------------------------------------
 * Weight packing (one-time, at load).  Device-to-device kernels.
 */

/* DeepBiaffineScorer.W1/W2 (biaffine.py:73-74) -> W12[2*KP, D], b12[2*KP]: rows 0..H-1 = W1,
 * row H = zeros with bias 1.0 (the ones column), rows H+1..KP-1 zero; then the same for W2. */
int phonlp_pack_scorer_linear(const float *W1, const float *b1, const float *W2, const float *b2, int H, int D,
                              float *W12, float *b12, void *stream);

/* PairwiseBilinear.weight (biaffine.py:18), stored (H+1, H+1, O) -> Wbt[O*KP, KP] with
 * Wbt[o*KP + d2][d1] = flat[d1*(H+1)*O + o*(H+1) + d2], zero elsewhere.  This bakes in the
 * reference's flat-reshape quirk (biaffine.py:27-35).  The module's
 * `bias` is never used by the reference (item 1) and has no counterpart here. */
int phonlp_pack_bilinear(const float *weight, int H, int O, float *Wbt, void *stream);

/* ------------------------------------------------------------------------------------------
 * POS head -- annotate_model.py:132-142 (tagger_forward) and :172-179 (dep_forward);
 * twin jointmodel/model.py:125-137.
 *   hid = relu(enc . Whid^T + bhid); logits = hid . Wclf^T + bclf           (per real token)
 *   pos_pred = argmax(logits) (lowest index on ties; 0 on padded rows)
 *   feat = [enc | softmax(logits) . emb]      (the 768+100 input of the next layer)
 * enc: [B*L, E] (ld_enc floats per row).  feat: [B*L, E+TE] (padded rows zero).
 * pos_pred [B*L] and upos_logits [B*L, U] are optional (NULL to skip).
 * emb == NULL (U <= TE <= 32): the embedding is the identity, feat = [enc | softmax(logits) | 0] -- what the scorers
 * read once upos_emb_matrix_dep is folded into their first layers (W[:, E:] . emb^T); upos_clf then runs inside the tail
 * kernel (one launch for two; the 400-term sums in a fixed order of their own, the same for every batch size).
 */
size_t phonlp_pos_head_ws_bytes(int B, int L, int E, int H, int U, int TE);
int phonlp_pos_head(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E,
                    const float *Whid, const float *bhid, const float *Wclf, const float *bclf, const float *emb,
                    int H, int U, int TE, float *feat, int32_t *pos_pred, float *upos_logits,
                    void *ws, size_t ws_bytes, void *stream);

/* The same head writing the feat rows as an engine-2 split image (feat_planes: phonlp_split_bytes(B*L, E+TE) bytes,
 * feat_inv_scale [B*L]; identical to phonlp_split_rows of feat) for phonlp_linear_split, so no split pass over feat is
 * needed; the fp32 feat itself is optional (feat_opt may be NULL). */
int phonlp_pos_head_split(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E,
                          const float *Whid, const float *bhid, const float *Wclf, const float *bclf, const float *emb,
                          int H, int U, int TE, float *feat_opt, void *feat_planes, float *feat_inv_scale,
                          int32_t *pos_pred, float *upos_logits, void *ws, size_t ws_bytes, void *stream);

/* Fused tagger head -- annotate_model.py:132-150 (tagger_forward after the gather) in one call, without forming
 * feat = [enc | pos_emb]: since ner_tag_clf is linear in feat,
 *   ner_logits = pad(relu(Wner[:, :E] . enc + Wner[:, E:] . pos_emb + bner)),
 * the enc part rides along the upos_hid GEMM as T extra output columns and the embedding part is added in the tail
 * kernel.  Whn [H+T, E] = upos_hid.weight rows followed by ner_tag_clf.weight[:, :E]; bhn [H+T] = upos_hid.bias followed
 * by zeros; Wner_emb [T, TE] = ner_tag_clf.weight[:, E:]; emb = upos_emb_matrix_ner [U, TE].  Same results as
 * phonlp_pos_head + phonlp_ner_emission to fp32 rounding (the 868-term dot product is summed in another order).
 * pos_pred [B*L], ner_logits [B*L, T]; upos_logits [B*L, U] optional (NULL to skip).
 * emb == NULL (TE == U <= 32): the embedding is the identity and Wner_emb [T, U] = ner_tag_clf.weight[:, E:] . emb^T, folded
 * by the caller at load; upos_clf then runs inside the tail kernel as in phonlp_pos_head. */
size_t phonlp_tagger_head_ws_bytes(int B, int L, int E, int H, int U, int TE, int T);
int phonlp_tagger_head(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E,
                       const float *Whn, const float *bhn, const float *Wclf, const float *bclf, const float *emb,
                       const float *Wner_emb, const float *bner, int H, int U, int TE, int T,
                       int32_t *pos_pred, float *ner_logits, float *upos_logits, void *ws, size_t ws_bytes, void *stream);

/* NER emissions -- annotate_model.py:149-150: logits[B,L,T] = pad(relu(feat . Wner^T + bner)). */
int phonlp_ner_emission(const float *feat, int D, const int32_t *sentlens, int B, int L,
                        const float *Wner, const float *bner, int T, float *logits, void *stream);

/* dep_hid -- annotate_model.py:186-187: hidden[B*L, D] = pad(feat . Wdep^T + bdep). */
int phonlp_dep_hidden(const float *feat, int D, const int32_t *sentlens, int B, int L,
                      const float *Wdep, const float *bdep, float *hidden, void *stream);

/* ------------------------------------------------------------------------------------------
 * Arc scores -- annotate_model.py:189,192-214 (twin jointmodel/model.py:185-207,239):
 *   unlabeled/linearization/distance DeepBiaffineScorers (O=1) on hidden (both sides),
 *   arc = unl + logsigmoid(lin*sign(j-i)) + (-log((|j-i| - 1 - softplus(dist))^2/2 + 1)),
 *   diagonal = -inf, log_softmax over ALL L head columns (padding included, no mask).
 * n_scorers = 1 (unlabeled only), or 3 (unlabeled, linearization, distance; either optional
 * one can be disabled with use_lin/use_dist = 0 but the pack keeps its slot).
 * W12 [n_scorers*2*KP, D], b12 [n_scorers*2*KP], Wbt [n_scorers*KP, KP] from the pack calls.
 * arc_logp [B,L,L].  raw_opt: NULL or [3,B,L,L] receiving the raw scorer outputs.
 *
 * Folded form (pad_sentlens and b12_pad both non-NULL; both NULL otherwise): `hidden` holds the rows BEFORE dep_hid
 * (feat = [x ; pos_emb]; padded rows may hold anything) and W12/b12 were packed from W1.Wdep, W1.bdep + b1 (resp.
 * W2): one GEMM instead of annotate_model.py:186-187 followed by biaffine.py:82-85's first layer.  Rows at or beyond
 * pad_sentlens[b] are written as relu(b12_pad) -- b12_pad is the pack of the ORIGINAL biases -- which is what the
 * reference's scorers make of pad_packed_sequence's all-zero hidden rows.  Same convention for the label entries.
 */
size_t phonlp_biaffine_arc_ws_bytes(int B, int L, int H, int n_scorers);
int phonlp_deep_biaffine_arc(const float *hidden, int D, int B, int L, int H, int n_scorers, int use_lin, int use_dist,
                             const float *W12, const float *b12, const float *Wbt,
                             float *arc_logp, float *raw_opt, void *ws, size_t ws_bytes, void *stream,
                             const int32_t *pad_sentlens, const float *b12_pad);

/* The same from the scorers' first-layer features XF [B*L, n_scorers*2*KP] (what the W1/W2 layer of
 * phonlp_deep_biaffine_arc writes: per scorer (relu(W1 x + b1) | 1 | 0.., relu(W2 x + b2) | 1 | 0..)), for callers
 * that compute that layer themselves (engine 2: phonlp_linear_split on split images).  biaffine.py:62-65,21-37 +
 * annotate_model.py:192-214. */
size_t phonlp_biaffine_arc_scores_ws_bytes(int B, int L, int H, int n_scorers);
int phonlp_biaffine_arc_scores(const float *XF, int B, int L, int H, int n_scorers, int use_lin, int use_dist,
                               const float *Wbt, float *arc_logp, float *raw_opt, void *ws, size_t ws_bytes,
                               void *stream);

/* Engine 2, stages A and B of the arc scorers in ONE kernel (biaffine.py:21-37 per scorer, annotate_model.py:192-214),
 * for sentences of up to 512 words and H + 1 <= 416: XFp / x_inv = the first-layer image phonlp_linear_split_image
 * wrote with N = n_scorers * 2 * KC columns, KC = phonlp_arc_kc(H) (per scorer the segments relu(W1 x + b1) | 1 | 0..
 * and relu(W2 x + b2) | 1 | 0.., KC columns each); Wbp / wb_inv = the split image of the packed bilinear weights with
 * every scorer's rows AND columns zero-padded to KC ([n_scorers * KC, KC]).  rows_total = rows of the image.  Packed
 * ragged batches: off [B] (first row of every sentence), sentlens [B], pad_row / pad_inv = the image row (and inverse
 * scale) of a padded head, i.e. of what the first layer gives a zero hidden row; else all four NULL / 0.
 * Same results as phonlp_biaffine_arc_scores on the fp32 features, to fp32 rounding. */
int phonlp_arc_kc(int H);
int phonlp_arc_scores_split(const void *XFp, const float *x_inv, int rows_total, int B, int L, int H, int n_scorers,
                            int use_lin, int use_dist, const void *Wbp, const float *wb_inv, const int32_t *off,
                            const int32_t *sentlens, const void *pad_row, float pad_inv, float *arc_logp, void *stream);
/* The same with a workspace: for FEW SHORT sentences (3 B <= 512, L <= 128, three scorers) one workgroup per sentence
 * would run the three scorers in a row on a fraction of the chip; given ws (phonlp_arc_scores_split_ws_bytes(B, L,
 * n_scorers) bytes, 0 = this shape does not use one; 16-byte aligned) the scorers run as separate workgroups that leave
 * their raw scores in ws, and a second launch adds the terms in the same order and takes the log-softmax: the same
 * bits as without ws (tests/test_gpu_small_batch.py), B=32, L=64: 0.113 -> 0.05 ms.  ws = NULL: exactly the entry above. */
size_t phonlp_arc_scores_split_ws_bytes(int B, int L, int n_scorers);
int phonlp_arc_scores_split_ws(const void *XFp, const float *x_inv, int rows_total, int B, int L, int H, int n_scorers,
                               int use_lin, int use_dist, const void *Wbp, const float *wb_inv, const int32_t *off,
                               const int32_t *sentlens, const void *pad_row, float pad_inv, float *arc_logp, void *ws,
                               size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Label scores -- annotate_model.py:190,215 (deprel DeepBiaffineScorer + .max(3)[1]):
 *   label_argmax[b,i,j] = argmax_o score[b,i,j,o], lowest o on ties (int32; the Python seam
 *   widens to int64).  scores_opt: NULL or [B,L,O,L] (the reference's memory order,
 *   biaffine.py:35) receiving every score.
 */
size_t phonlp_biaffine_label_ws_bytes(int B, int L, int H, int O);
int phonlp_deep_biaffine_label(const float *hidden, int D, int B, int L, int H, int O,
                               const float *W12, const float *b12, const float *Wbt,
                               int32_t *label_argmax, float *scores_opt, void *ws, size_t ws_bytes, void *stream,
                               const int32_t *pad_sentlens, const float *b12_pad);

/* The same scorer cut at the MST, for annotate(): the reference computes the whole label tensor
 * (annotate_model.py:190,215) but reads it only at j = heads[b,i] (annotate_model.py:263-266:
 * `deprel_seqs[i][j+1][h]` after the MST).  phonlp_label_project runs the scorer's two GEMMs into caller
 * buffers -- XF [B*L, 2*KP] = (relu(W1 x + b1) | 1 | 0.. , relu(W2 x + b2) | 1 | 0..), T [B*L, O*KP] with
 * T[b,i,o,:] = h1[b,i]^T W_o -- and can run beside the MST; phonlp_label_at_heads then gives
 *   rels[b,i] = argmax_o T[b,i,o,:] . h2[b,heads[b,i],:]   for 1 <= i < sentlens[b], 0 elsewhere
 * (lowest o on ties), i.e. phonlp_gather_labels(phonlp_deep_biaffine_label(...)) without the L x L x O
 * contraction.  best_opt: NULL or [B*L] receiving the winning score.
 */
int phonlp_label_project(const float *hidden, int D, int B, int L, int H, int O,
                         const float *W12, const float *b12, const float *Wbt, float *XF, float *T, void *stream,
                         const int32_t *pad_sentlens, const float *b12_pad);
int phonlp_label_at_heads(const float *T, const float *XF, const int32_t *heads, const int32_t *sentlens,
                          int B, int L, int H, int O, int32_t *rels, float *best_opt, void *stream);

/* Engine 2 form of the same pair.  phonlp_label_stage_a_split is stage A (biaffine.py:27-29) on split images
 * (X1p/x1_inv: phonlp_split_rows of XF[:, 0:KP]; Wbtp/wbt_inv: of the phonlp_pack_bilinear image) and writes T in the
 * "rows interleaved by 4" layout T4[(m/4)*O*KP + c][m%4] = T[m][c] (phonlp_label_t4_bytes bytes), which is the order
 * the MFMA accumulators hold it in; phonlp_label_at_heads_rows4 is phonlp_label_at_heads reading that layout
 * (same fp32 dot products in another summation order: equal to rounding).  ones_col: NULL, or [O*KP] = column H of the
 * phonlp_pack_bilinear image (Wbt[c][H], the weights that meet X1's ones column): the term is then added per column in
 * the epilogue instead of inside the product, which saves a k16 stage when H % 16 == 0. */
size_t phonlp_label_t4_bytes(int B, int L, int H, int O);
int phonlp_label_stage_a_split(const void *X1p, const float *x1_inv, const void *Wbtp, const float *wbt_inv,
                               const float *ones_col, int B, int L, int H, int O, float *T4, void *stream);
int phonlp_label_at_heads_rows4(const float *T4, const float *XF, const int32_t *heads, const int32_t *sentlens,
                                int B, int L, int H, int O, int32_t *rels, float *best_opt, void *stream);

/* The pair fused (engine 2): stage A's tiles are multiplied by the head's features and summed while they are still
 * in the accumulators, so T (1.8 GB at B=256, L=128) is neither written nor read; needs the heads first, i.e. runs
 * after the MST.  The product is taken transposed (weight rows x words: an accumulator lane then holds one word and
 * sixteen head features, and the contraction needs no cross-lane sums); the heads' features are gathered and transposed
 * into the workspace first.  Wbtcp/wbtc_inv = split image of the [O*KPC, KP] matrix whose row (blk*O + o)*32 + r is row
 * o*KP + 32*blk + r of the phonlp_pack_bilinear image (KPC = phonlp_label_kpc(H) = KP rounded up to 32, blk < KPC/32,
 * r < 32; rows past KP zero): relations interleaved by blocks of 32 head features, so that the relations a wavefront
 * scores share one block of the heads' features.  wbtc_inv and the workspace must be 16-byte aligned.  Same scores
 * as phonlp_label_at_heads to fp32 rounding (another, fixed, summation order).  best_opt may be NULL. */
int phonlp_label_kpc(int H);
size_t phonlp_label_at_heads_fused_ws_bytes(int B, int L, int H, int O);
int phonlp_label_at_heads_fused(const void *X1p, const float *x1_inv, const void *Wbtcp, const float *wbtc_inv,
                                const float *XF, const int32_t *heads, const int32_t *sentlens, int B, int L, int H,
                                int O, int32_t *rels, float *best_opt, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Viterbi -- models/common/crf.py:103-126 (viterbi_decode), per sentence on logits[b,:len].
 * trans[prev*T + cur].  tags [B,L] (entries >= len are set to 0), best [B] or NULL.
 * One wavefront per sentence; T <= 64.
 */
int phonlp_crf_viterbi(const float *logits, const float *trans, const int32_t *sentlens, int B, int L, int T,
                       int32_t *tags, float *best, void *stream);

/* CRF forward algorithm -- crf.py:79-100 (crf_log_norm) and the gold-path scores :48-77.
 * pad_mask [B,L] bytes (non-zero = padding, the reference's `masks`) and/or sentlens, either may
 * be NULL.  gold_tags (int32 [B,L]) optional: when given, unary[B] and binary[B] are written.
 * log_norm [B].  T <= 64.
 */
int phonlp_crf_log_partition(const float *logits, const float *trans, const uint8_t *pad_mask,
                             const int32_t *sentlens, const int32_t *gold_tags, int B, int L, int T,
                             float *log_norm, float *unary, float *binary, void *stream);

/* Gradient of the CRF negative log-likelihood -- what autograd derives from crf.py:24-100 (CRFLoss.forward) for the
 * emissions and the transition matrix, by forward-backward:
 *   d_logits[b,t,c] = scale * (P(tag_t = c) - [c == gold_t]), 0 on padded steps;
 *   d_trans[p,c]    = scale * sum_b sum_t (P(tag_{t-1} = p, tag_t = c) - [p == gold_{t-1}, c == gold_t]).
 * scale = 1/B for the reference's batch_average (crf.py:39-40), times the incoming gradient.  pad_mask / sentlens as
 * for phonlp_crf_log_partition.  d_logits [B,L,T], d_trans [T,T]; workspace phonlp_crf_nll_grad_ws_bytes.  T <= 64.
 */
size_t phonlp_crf_nll_grad_ws_bytes(int B, int T);
int phonlp_crf_nll_grad(const float *logits, const float *trans, const uint8_t *pad_mask, const int32_t *sentlens,
                        const int32_t *gold_tags, int B, int L, int T, float scale, float *d_logits, float *d_trans,
                        void *ws, size_t ws_bytes, void *stream);

/* The same gradient for callers that CANNOT ALLOCATE (code under HIP graph capture, tight training / serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The B x T x T scratch of phonlp_crf_nll_grad
 * lives in a file-scope device array of phonlp_crf_nll_grad_static_max_bytes() = 4 MiB per GPU (held until the
 * process ends); B*T*T*4 bytes beyond that is rejected with PHONLP_ERR_UNSUPPORTED.  The call allocates and frees
 * nothing; the first call on a device looks the array's address up (hipGetSymbolAddress, which may load the module's
 * device variables): make that one call OUTSIDE graph capture, every later one may be captured.  Every call on a GPU shares
 * that GPU's array: calls that overlap on one GPU (two streams, two threads) race -- serialise them.  Nothing in the
 * array needs clearing between calls (every entry read was written by the same call).  Same results as
 * phonlp_crf_nll_grad, bit for bit. */
size_t phonlp_crf_nll_grad_static_max_bytes(void);
int phonlp_crf_nll_grad_static(const float *logits, const float *trans, const uint8_t *pad_mask,
                               const int32_t *sentlens, const int32_t *gold_tags, int B, int L, int T, float scale,
                               float *d_logits, float *d_trans, void *stream);

/* ------------------------------------------------------------------------------------------
 * Chu-Liu-Edmonds single-root MST -- models/common/chuliu_edmonds.py:133-169, applied to
 * arc[b, :len, :len] (rows = dependents, cols = heads, node 0 = ROOT), float64 arithmetic,
 * the reference's tie-breaking and contraction order.  heads [B,L] int32 (heads[b,0] = 0;
 * entries >= len set to 0); status [B] int32 (PHONLP_MST_*).  One workgroup per sentence.
 *
 * Two decoders produce these trees.  The reference-order kernel replays the reference's one-cycle-per-level
 * recursion step by step.  The round-based kernel (L <= 128, on by default) contracts every cycle of a round together
 * and is used per sentence only where it provably returns the reference's tree: all of the reference's float64 sums
 * are exact for that sentence's scores and every selection has a strictly unique maximum (csrc/mst_fast.hip); any
 * other sentence (ties, integer-valued scores, denormals, NaN) is decoded by the reference-order kernel in the same
 * call.  phonlp_set_mst_fast_path(0) sends every sentence to the reference-order kernel (A/B and tests; process-wide,
 * not thread-safe); phonlp_mst_fallback_counter_offset gives the byte offset inside `ws` of an int32 that holds, after
 * the call, how many sentences were handed over (-1 if the round-based kernel does not apply to this L).
 * The workspace also holds the round-based decoder's expansion records (3 bytes per word and round, kept out of LDS
 * so that a dense-layer block fits a CU beside a decode): phonlp_mst_ws_bytes(256, 128) is 90 MB, 51 MB of it records.  Two calls
 * that overlap on one GPU need a workspace each.
 */
size_t phonlp_mst_ws_bytes(int B, int L);
int phonlp_mst_one_root(const float *arc, const int32_t *sentlens, int B, int L,
                        int32_t *heads, int32_t *status, void *ws, size_t ws_bytes, void *stream);
int phonlp_set_mst_fast_path(int on);
int phonlp_get_mst_fast_path(void);
long phonlp_mst_fallback_counter_offset(int B, int L);
long phonlp_mst_fallback_list_offset(int B, int L); /* int32 [B]: sentence | reason << 24 (1 inexact, 2/3 a tie) */

/* Decode glue -- annotate_model.py:263-266: rel[b,i] = label_argmax[b, i, heads[b,i]] for
 * 1 <= i < len, 0 elsewhere. */
int phonlp_gather_labels(const int32_t *label_argmax, const int32_t *heads, const int32_t *sentlens, int B, int L,
                         int32_t *rels, void *stream);

/* ------------------------------------------------------------------------------------------
 * First-sub-word gather -- annotate_model.py:119-123 / :159-163 (one index_select per sentence in a Python
 * loop): out[b, w, :] = enc[b, first_subword[b, w], :] for w < sentlens[b], zeros beyond (sentlens may be NULL:
 * then every row is gathered, like the reference, whose padded rows are dropped later by pack()).
 * enc [B,S,E] last hidden state, first_subword [B,L] int32 or int64 (index_is_int64), out [B,L,E].  E % 4 == 0.
 */
int phonlp_gather_first_subword(const float *enc, const void *first_subword, int index_is_int64,
                                const int32_t *sentlens, int B, int S, int L, int E, float *out, void *stream);

/* Device-side unsort -- annotate_model.py:270,277,287 (util.unsort within the batch) and :291-293 (over the corpus),
 * models/common/utils.py:189-195: the rows of a batch's result go straight to their sentences' places in a corpus-level
 * table, dst[dest_row[b]][0..n) = src[b][col0..col0+n) with n = min(Ls - col0, Ld) (col0 = 1 drops the parser's ROOT
 * column); dest_row[b] = the sentence's index in the input file (the two sort permutations composed on the host, which
 * made them).  status_opt / fail_count_opt: *fail_count += number of non-zero entries of status [B] (MST failures),
 * so a corpus needs one check at the end instead of one device->host round trip per batch. */
int phonlp_scatter_rows(const int32_t *src, int B, int Ls, int col0, const int32_t *dest_row, int32_t *dst, int Ld,
                        int n_dst, const int32_t *status_opt, int32_t *fail_count_opt, void *stream);

/* ------------------------------------------------------------------------------------------
 * Packed rows -- ragged batches without the padding's arithmetic.
 * The reference runs every dense layer on PackedSequence data, i.e. on the live words only (annotate_model.py:114-130,
 * 155-170: pack() = pack_padded_sequence, pad() = pad_packed_sequence), and pads again before the biaffine scorers
 * (:187); the padded tensors it scores (pad_packed_sequence zeros -> relu(b) rows) are constant rows.  Here the same
 * thing sentence-major: the live words (b, i), i < sentlens[b], of a [B, L, .] tensor are packed one sentence after the
 * other into m_live = sum(sentlens) rows, every row-wise entry point above (phonlp_pos_head_split, phonlp_tagger_head,
 * phonlp_linear_split, phonlp_split_rows) is then called with B = 1, L = m_live on the packed rows, and the three entry
 * points that need the sentence structure take the packing's descriptor:
 *   off[b]    = sentence b's first packed row (off[B] = m_live; usable as the 1-sentence "sentlens" of those calls),
 *   rowmap[p] = b*L + i of packed row p.
 * phonlp_pack_descriptor builds both on the device from sentlens (rowmap holds `cap` >= m_live entries);
 * phonlp_pack_rows gathers fp32 rows (dst[p] = src[rowmap[p]], W % 4 == 0); phonlp_unpack_rows scatters 4-byte rows back
 * (dst [n_dst, W] zero-filled first, dst[rowmap[p]] = src[p]): pos_pred / NER emissions of the padding are zeros as in
 * the unpacked entry points.
 * phonlp_biaffine_arc_scores_packed = phonlp_biaffine_arc_scores with XF packed; pad_row [n_scorers*2*KP] is the first
 * layer's output for a padded word (relu of the layer's own bias: the hidden row is zero).  The head columns
 * j >= sentlens[b] take pad_row, so every live dependent's log-softmax runs over all L columns exactly as in the
 * unpacked form; the rows of padded dependents are not written (the decoders never read them).
 * phonlp_label_at_heads_fused_packed = phonlp_label_at_heads_fused with X1p / XF packed; heads / rels / best_opt keep
 * their [B, L] shape (labels of ROOT and padding 0).  Workspace: phonlp_label_at_heads_fused_ws_bytes(1, m_live, H, O)
 * resp. phonlp_biaffine_arc_scores_ws_bytes(1, m_live, H, n_scorers). */
int phonlp_pack_descriptor(const int32_t *sentlens, int B, int L, int32_t *off, int32_t *rowmap, int cap, void *stream);
int phonlp_pack_rows(const float *src, int ld_src, int n_src, const int32_t *rowmap, int m, int W, float *dst, void *stream);
int phonlp_unpack_rows(const void *src, const int32_t *rowmap, int m, int W, void *dst, int n_dst, void *stream);
int phonlp_biaffine_arc_scores_packed(const float *XF, int m_live, const int32_t *off, const int32_t *sentlens,
                                      const float *pad_row, int B, int L, int H, int n_scorers, int use_lin, int use_dist,
                                      const float *Wbt, float *arc_logp, float *raw_opt, void *ws, size_t ws_bytes,
                                      void *stream);
int phonlp_label_at_heads_fused_packed(const void *X1p, const float *x1_inv, const void *Wbtcp, const float *wbtc_inv,
                                       const float *XF, const int32_t *heads, const int32_t *sentlens,
                                       const int32_t *rowmap, const int32_t *off, int m_live, int B, int L, int H, int O,
                                       int32_t *rels, float *best_opt, void *ws, size_t ws_bytes, void *stream);

/* Eval counts -- what the reference's scorers accumulate per corpus (models/pos/scorer.py:29-36 POS accuracy;
 * utils/conll18_ud_eval.py:531-532 UAS/LAS; tag-level NER agreement).  ADDS to counts[0..7] (int64, device):
 * sentences, words, pos_correct, ner_correct, uas_correct, las_correct, mst_failures, 0.  Tagger tensors are
 * [B,Lt], parser tensors [B,Ld] with column 0 = ROOT (skipped).  mst_status may be NULL.
 */
int phonlp_eval_counts(const int32_t *pos, const int32_t *ner, const int32_t *heads, const int32_t *rels,
                       const int32_t *gold_pos, const int32_t *gold_ner, const int32_t *gold_head,
                       const int32_t *gold_rel, const int32_t *lens_tag, const int32_t *lens_dep,
                       const int32_t *mst_status, int B, int Lt, int Ld, int64_t *counts, void *stream);

/* Entity-level NER counts -- models/ner/scorer.py:13-57 (score_by_entity) over ner/utils.py:85-128
 * (decode_from_bioes): counts[0..2] += {predicted entities that are also gold entities (same sentence, first index,
 * last index and type), predicted entities, gold entities}; precision/recall/F1 follow on the host from the
 * all-reduced sums.  pred/gold int32 [B,L] tag ids, sentlens int32 [B]; tag_kind int32 [T] = 0 'O', 1 'B-', 2 'I-',
 * 3 'E-', 4 'S-', any other value for tags the decoder ignores (<PAD>, <UNK>, ...); tag_type int32 [T] = id of the
 * type string after the prefix.  L <= 4096.
 */
int phonlp_ner_entity_counts(const int32_t *pred, const int32_t *gold, const int32_t *sentlens,
                             const int32_t *tag_kind, const int32_t *tag_type, int B, int L, int T,
                             int64_t *counts, void *stream);

/* ------------------------------------------------------------------------------------------
 * Parser training loss, forward value -- models/jointmodel/model.py:209-236 (dep_forward, eval=False branch) with
 * every dropout the identity (no autograd):
 *   loss[0] = arc cross-entropy (:211-216: rows 1 <= i < len, soft-max over the sentence's own columns),
 *   loss[1] = label cross-entropy at the gold heads (:218-223, same rows),
 *   loss[2] = linearization cross-entropy (:225-230) and loss[3] = distance term (:232-234) -- both over EVERY row
 *             i >= 1, padded rows included, as the reference sums them (it masks neither),
 *   loss[4] = (loss[0] + loss[1] + loss[2] + loss[3]) / number_of_words (:236).
 * Inputs are what the eval path produces: arc_logp [B,L,L] and raw [3,B,L,L] (raw_opt) of phonlp_deep_biaffine_arc /
 * phonlp_biaffine_arc_scores, T and XF of phonlp_label_project.  gold_head / gold_rel: int32 [B,L], column 0 (ROOT)
 * unused, i.e. the reference's B x (L-1) tensors shifted right by one; gold_rel -1 = ignored word.  raw may be NULL
 * when use_lin = use_dist = 0.  loss: 5 floats on the device.  Workspace: phonlp_dep_loss_ws_bytes.
 */
size_t phonlp_dep_loss_ws_bytes(int B, int L);
int phonlp_dep_loss(const float *arc_logp, const float *raw, const float *T, const float *XF,
                    const int32_t *gold_head, const int32_t *gold_rel, const int32_t *sentlens, int B, int L, int H,
                    int O, int use_lin, int use_dist, float number_of_words, float *loss, void *ws, size_t ws_bytes,
                    void *stream);

/* Backward of that loss w.r.t. the OUTPUTS of the four scorers -- what autograd derives from
 * models/jointmodel/model.py:209-236 (the linearization / distance terms are added to the unlabeled scores .detach()ed,
 * :197,:204, so the arc cross-entropy reaches the unlabeled scorer only).  Same inputs as phonlp_dep_loss.  Outputs, every
 * entry written: d_arc [B,L,L] = (soft-max over the sentence's own columns - one-hot(gold head)) / number_of_words on the
 * rows 1 <= i < len, zero elsewhere; d_label [B,L,O] = (soft-max over the relations at the gold head - one-hot(gold
 * relation)) / number_of_words on the same rows (the gradient of deprel_scores[b,i,head_i,:]; all other entries of that
 * tensor get none); d_lin, d_dist [B,L] = the gradient of lin / dist[b,i,head_i] on EVERY row i >= 1, padded rows
 * included (the reference masks neither term, :225-234).  No workspace. */
int phonlp_dep_loss_grad(const float *arc_logp, const float *raw, const float *T, const float *XF,
                         const int32_t *gold_head, const int32_t *gold_rel, const int32_t *sentlens, int B, int L, int H,
                         int O, int use_lin, int use_dist, float number_of_words, float *d_arc, float *d_label,
                         float *d_lin, float *d_dist, void *stream);

/* Backward THROUGH the layers (what autograd does behind the reference's modules; the gradients w.r.t. the scores come from
 * the *_grad entry points above).  All products run on the library's dense-layer kernels (fp32-grade on every engine);
 * reduction dimensions are zero-padded to multiples of 4 inside the workspace, so the shapes are free; column sums are
 * taken in a fixed order (no float atomics).
 *
 * phonlp_linear_backward: an nn.Linear (+ ReLU) layer Y = act(X W^T + b) -- upos_hid / upos_clf / ner_tag_clf / dep_hid
 * (annotate_model.py:43-60) and the scorers' W1 / W2 (biaffine.py:73-74, 83): X [M, K] (row pitch ldx), W [N, K], Yact =
 * the layer's OUTPUT [M, N] (read only when relu: the mask is Yact > 0), dY [M, N]  ->  dX [M, K] = dZ W, dW [N, K] =
 * dZ^T X, db [N] = column sums of dZ, with dZ = dY * [Yact > 0] (relu) or dY.  Any of dX / dW / db may be NULL. */
size_t phonlp_linear_backward_ws_bytes(int M, int N, int K);
int phonlp_linear_backward(const float *X, int ldx, const float *W, int ldw, const float *Yact, int ldy, const float *dY,
                           int lddy, int M, int N, int K, int relu, float *dX, int lddx, float *dW, int lddw, float *db,
                           void *ws, size_t ws_bytes, void *stream);

/* phonlp_bilinear_backward: a ONE-output PairwiseBilinear (biaffine.py:21-37, O = 1: the unlabeled, linearization and
 * distance scorers), S[b,i,j] = sum X1'[b,i,d1] Wbt[d2][d1] X2'[b,j,d2] with XF = (X1' | X2') [B*L, 2*KP] the scorer's
 * post-ReLU features incl. the ones column (as phonlp_label_project / the arc path produce them) and Wbt the
 * phonlp_pack_bilinear image [KP, KP].  Given G = dL/dS [B, L, L]:  dXF [B*L, 2*KP] = (dX1' | dX2') (columns >= H: the
 * ones / padding columns, to be ignored), dWbt [KP, KP] in the Wbt layout (dweight[d1][d2] of the reference's
 * (H+1) x (H+1) x 1 parameter = dWbt[d2][d1]).  The label scorer: phonlp_label_backward below. */
size_t phonlp_bilinear_backward_ws_bytes(int B, int L, int H);
int phonlp_bilinear_backward(const float *XF, const float *Wbt, const float *G, int B, int L, int H, float *dXF,
                             float *dWbt, void *ws, size_t ws_bytes, void *stream);

/* phonlp_label_backward: the label scorer (O relations), whose loss reads the scores at the GOLD heads only
 * (jointmodel/model.py:218-223): g [B, L, O] = dL/d deprel_scores[b, i, gold_head[b,i], :] as phonlp_dep_loss_grad's
 * d_label gives it (zero on ROOT / padding rows), XF [B*L, 2*KP] and Wbt [O*KP, KP] as for phonlp_label_project, T_opt
 * = phonlp_label_project's T [B*L, O*KP] when the caller kept it (NULL: recomputed).
 * dXF [B*L, 2*KP] = (dX1' | dX2') (dX2' summed over the dependents of each head, ascending), dWbt [O*KP, KP].  L <= 1024. */
size_t phonlp_label_backward_ws_bytes(int B, int L, int H, int O);
int phonlp_label_backward(const float *XF, const float *Wbt, const float *T_opt, const float *g, const int32_t *gold_head,
                          const int32_t *sentlens, int B, int L, int H, int O, float *dXF, float *dWbt, void *ws,
                          size_t ws_bytes, void *stream);

/* The soft POS distribution pos_dis = softmax(upos logits) that feeds the tag embedding of the NER and parser heads
 * (jointmodel/model.py:135,150,171), rows of [M, U] (U <= 1024), and its backward dZ = P * (dP - sum_u P dP). */
int phonlp_softmax_rows(const float *Z, int M, int U, float *P, void *stream);
int phonlp_softmax_rows_backward(const float *P, const float *dP, int M, int U, float *dZ, void *stream);

/* POS training loss, forward value -- models/jointmodel/model.py:128-135 (pos_forward with use_soft_pos=False; crit_pos
 * :94 = CrossEntropyLoss(ignore_index=1), mean reduction), dropout the identity:
 *   loss[0] = mean over the words i < sentlens[b] with gold_upos != ignore_index of
 *             logsumexp(upos_logits[b,i,:]) - upos_logits[b,i,gold]  (NaN when no word counts, as torch),
 *   loss[1] = the sum, loss[2] = the number of words counted.
 * upos_logits [B,L,U] as phonlp_pos_head writes them, gold_upos int32 [B,L].  The reference passes ignore_index = 1.
 */
size_t phonlp_pos_loss_ws_bytes(int B, int L);
int phonlp_pos_loss(const float *upos_logits, const int32_t *gold_upos, const int32_t *sentlens, int B, int L, int U,
                    int ignore_index, float *loss, void *ws, size_t ws_bytes, void *stream);

/* Gradient of that loss w.r.t. the logits, as autograd derives it from crit_pos (jointmodel/model.py:94,134):
 * d_logits[b,i,:] = (softmax(upos_logits[b,i,:]) - onehot(gold)) / count on the words that count, 0 elsewhere.
 * count: DEVICE pointer to the number of counted words (loss + 2 of phonlp_pos_loss), read by the kernel. */
int phonlp_pos_loss_grad(const float *upos_logits, const int32_t *gold_upos, const int32_t *sentlens,
                         const float *count, int B, int L, int U, int ignore_index, float *d_logits, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing (no counterpart in the reference, which has no profiling: SURVEY.md section 5).
 * Between _begin and _end every kernel launch of the library is bracketed by a hipEvent pair recorded
 * on its launch stream (up to max_samples launches).  _end synchronises those events and copies the
 * durations (milliseconds) and a tag id per launch to HOST arrays; returns the number of samples or a
 * negative error.  phonlp_profile_tag_name(id) names a tag ("linear:label_stage_a", "mst", ...).
 * phonlp_profile_only(tag) restricts the sampling to launches with that tag (NULL or "" = every launch): an event
 * pair costs a few microseconds of queue time per launch, so a timed run brackets only the kernel it reports.
 */
int phonlp_profile_begin(int max_samples);
int phonlp_profile_only(const char *tag);
int phonlp_profile_end(float *ms_out_host, int32_t *tag_out_host, int cap);
const char *phonlp_profile_tag_name(int id);

#ifdef __cplusplus
}
#endif
#endif /* PHONLP_HEADS_H */
