/*
 * libconvdr_hip.so -- C ABI of the MI355X (gfx950) kernels behind ConvDR's hot path.
 *
 * The reference (thunlp/ConvDR) is pure Python; it has no FFI of its own.  Every entry point
 * below replaces a *third-party arithmetic call site* of the reference and is
 * what a ctypes binding on the reference side would bind (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (tensor.data_ptr()) unless the name says host;
 *   - plain C types only, row-major contiguous arrays, explicit sizes;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no allocation and no synchronisation inside: the caller owns outputs and workspace;
 *   - return 0 on success, negative on error; convdr_last_error() gives the message
 *     (thread-local).  Kernels are enqueued asynchronously.
 */
#ifndef CONVDR_HIP_H
#define CONVDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* convdr_stream_t;

int convdr_version(void);
/* "domain:bus:device.function" of HIP device `device` (hipDeviceGetPCIBusId through the runtime this library is bound
 * to): the host side looks up the GPU's NUMA node with it and places its pinned staging buffers there. */
int convdr_device_pci_bus_id(int device, char* out, int len);
const char* convdr_last_error(void);

/* Optional per-kernel timing with hipEvents recorded on the launch stream (bench.py's roofline
 * leg).  enable(1) clears the recorded spans; collect() synchronises on the spans named `name`
 * ("ip_scan_emit", "ip_scan_sample", "ip_rescore", ...) and returns their summed duration. */
int convdr_prof_enable(int on);
int convdr_prof_collect(const char* name, float* total_ms, int* launches);

/* ------------------------------------------------------------------------------------------
 * Flat inner-product index: replaces faiss.IndexFlatIP(768) as driven by
 *   /root/reference/drivers/run_convdr_inference.py:353 (ctor), :180 (.add), :182 (.search), :202 (.reset)
 * ------------------------------------------------------------------------------------------ */

/* Column mean of an fp32 block [n, d] (the centring vector of the scan copies).  scratch: >= 1024 * d floats. */
int convdr_ip_column_mean(const float* p_f32, int64_t n, int d, float* scratch, float* mean, convdr_stream_t stream);

/* .add(block): build the bf16 scan copy of an fp32 block [n, d]:  p_bf16 = bf16(p - centre)  (centre: device fp32 [d]
 * or NULL).  Subtracting a fixed vector from every passage shifts all scores of a query by the same constant, so the
 * ranking is unchanged while the rounding-error bound of the scan shrinks from |q| max|p| to |q| max|p - centre|
 * (embeddings of one encoder share a large common component).  p_bf16_lo (nullable): bf16 of the rounding remainder
 * (p - centre) - hi, needed by the split-bf16 scan.  max_i ||p_i - centre||_2 is folded into *max_norm (device float,
 * caller zero-initialises it on reset).  d % 64 == 0. */
int convdr_ip_prepare_block(const float* p_f32, int64_t n, int d, const float* centre, void* p_bf16, void* p_bf16_lo,
                            float* max_norm, convdr_stream_t stream);

/* The same for the fp16 scan (the default first rung: v_mfma_f32_32x32x16_f16 runs at the bf16 rate with 11 significand
 * bits instead of 8, i.e. an 8x tighter error band per pass):  p_f16 = half(scale * (p - centre)),  p_f16_lo (nullable)
 * = half of the remainder.  `scale` is a power of two (exact) that moves the block's norms to ~2^12, away from both
 * ends of the half range; convdr_ip_f16_scale(max norm seen so far) proposes it.  *max_norm stays UNSCALED.
 * p_f16 == NULL: only fold the norms into *max_norm (the pass that finds the scale of a first block). */
int convdr_ip_prepare_block_f16(const float* p_f32, int64_t n, int d, const float* centre, float scale, void* p_f16,
                                void* p_f16_lo, float* max_norm, convdr_stream_t stream);
float convdr_ip_f16_scale(float max_norm);

/* Bytes of device workspace convdr_ip_search needs for these sizes. */
size_t convdr_ip_workspace_bytes(int nq, int64_t n, int d, int k, int cap);

/* per-query status written by convdr_ip_search */
#define CONVDR_IP_OK 0        /* result is the exact top-k (certified)                                  */
#define CONVDR_IP_OVERFLOW 1  /* more than `cap` candidates passed tau: retry with tau_retry (raise cap
                                 when tau_retry does not exceed the tau that was used)                    */
#define CONVDR_IP_TOO_FEW 2   /* fewer than min(k, n) candidates passed tau: retry with tau_retry         */
#define CONVDR_IP_UNCERTAIN 3 /* the re-score band reaches below tau: retry with tau_retry               */
#define CONVDR_IP_RANGE 4     /* fp16 scan only: scale * max norm > 60000, elements of the half copy may be
                                 inf -- rebuild the copy with convdr_ip_f16_scale(current max norm), search again */

/* .search(Q, k): exact inner-product top-k of nq fp32 queries against one resident block.
 *   scan     bf16 MFMA GEMM  S~ = P_bf16 * Q_bf16^T  with a fused per-query threshold test: passages with
 *            S~ >= tau[q] are appended to a candidate list (a [nq, n] score matrix is never written);
 *            tau comes from a bf16 scan of a 1/32 sample of the block, aimed at `rank_target` hits per
 *            query (or tau_in when given);
 *   cut      with eps = (2u + u^2 + d 2^-23) * ||q|| * max||p||, u = 2^-8 (rigorous bound on |S~ - exact|; 0.00792 at d = 768): the band
 *            {S~ >= S~(k) - 2 eps} provably contains the exact top-k;
 *   rescore  the band is re-scored from the fp32 originals in fp64 with the canonical summation order
 *            documented in oracle/search.py, and sorted by (score desc, index asc).
 * status is OK only if the candidate list is complete over the band (cut >= tau, no overflow), so an
 * OK result equals the exhaustive exact top-k; otherwise tau_retry[q] is the threshold to pass as
 * tau_in for that query.
 * Outputs (device): D [nq, k] fp32 scores (descending), I [nq, k] int64 row indices into the block
 * (-1 / -FLT_MAX padding when n < k, as FAISS does), status [nq] int32, tau_retry [nq] fp32.
 * tau_in: NULL, or device [nq] thresholds (retry path).  cap: candidate capacity per query
 * (power of two, 1024..8192).  rank_target: expected candidates per query (0 -> 16*k, at most cap/2).
 * p_bf16_lo: NULL = plain bf16 scan (eps = 0.00792 |q| max|p'| at d = 768); non-NULL = split-bf16 scan S~ = Ph Qh + Ph Ql + Pl Qh
 * (three MFMA passes, eps = (3 u^2 + 3 d 2^-23) |q| max|p'| = 3.2e-4 at d = 768): the second rung for clustered embeddings whose top scores are closer
 * together than the bf16 error band. */
int convdr_ip_search(const float* q_f32, int nq, const float* p_f32, const void* p_bf16, const void* p_bf16_lo, int64_t n, int d,
                     int k, const float* p_max_norm, const float* tau_in, int cap, int rank_target,
                     void* workspace, size_t workspace_bytes, float* D, int64_t* I, int32_t* status,
                     float* tau_retry, convdr_stream_t stream);

/* convdr_ip_search over the fp16 scan copy built by convdr_ip_prepare_block_f16 with the same p_scale.  Queries are
 * scaled per row by a power of two inside (norm -> [2^11, 2^12)); thresholds (tau_in / tau_retry) are in those scaled
 * units and are only meaningful for a retry of the same query against the same copy.
 *   eps = (2u + u^2 + d 2^-23) |q| max|p'| + eta (1 + u) sqrt(d) (|q| + max|p'|) + d eta^2,  u = 2^-11, eta = 2^-14
 *   (1.07e-3 |q| max|p'| at d = 768; the absolute term ~1e-6 of it);  p_f16_lo non-NULL: the split scan
 *   S~ = Ph Qh + Ph Ql + Pl Qh with 3 u^2 + 3 d 2^-23 = 2.8e-4.
 * status may also be CONVDR_IP_RANGE (see above).  D / I are defined exactly as for convdr_ip_search -- the rung only
 * decides which candidates are re-scored, never the result. */
int convdr_ip_search_f16(const float* q_f32, int nq, const float* p_f32, const void* p_f16, const void* p_f16_lo, float p_scale,
                         int64_t n, int d, int k, const float* p_max_norm, const float* tau_in, int cap, int rank_target,
                         void* workspace, size_t workspace_bytes, float* D, int64_t* I, int32_t* status,
                         float* tau_retry, convdr_stream_t stream);

/* Instrumentation of the last convdr_ip_search on this workspace (device uint32 [nq] each):
 * candidates emitted by the scan / size of the exactly re-scored band. */
const uint32_t* convdr_ip_debug_counts(const void* workspace, int nq, int64_t n, int d, int k, int cap);
const uint32_t* convdr_ip_debug_band(const void* workspace, int nq, int64_t n, int d, int k, int cap);

/* Two-way merge of per-query result lists: replaces the Python pointer walk of
 *   /root/reference/drivers/run_convdr_inference.py:213-229
 * Da/Ia [nq, na] is the running result (earlier blocks), Db/Ib [nq, nb] the new block's, every row sorted by score
 * descending.  Writes the first n_out (<= na + nb) entries of the complete merge; on equal scores the entry of list
 * A comes first (`>=`, :218) and each list keeps its own order -- the permutation a stable descending sort of the
 * concatenation [A, B] produces.  lda / ldb / ldo: row pitches in elements.  na, nb <= 4096. */
int convdr_topk_merge(const float* Da, const int64_t* Ia, int na, int64_t lda, const float* Db, const int64_t* Ib, int nb,
                      int64_t ldb, int nq, int n_out, float* Dout, int64_t* Iout, int64_t ldo, convdr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dual-encoder forward: replaces the HuggingFace RobertaModel / BertModel forward + pooling + head
 * behind  /root/reference/model/models.py:140-148 (RobertaDot_NLL_LN.query_emb / body_emb) and
 * :205-211, :227-235 (HFBertEncoder.forward, BiEncoder.query_emb / body_emb).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t kind;        /* 0 = RoBERTa position ids (cumsum(ids != pad_idx) * (ids != pad_idx) + pad_idx), 1 = BERT (0..L-1) */
  int32_t hidden, heads, layers, intermediate;   /* head_dim = hidden / heads must be 64 */
  int32_t vocab, max_pos, pad_idx;
  int32_t out_dim;     /* 768 for rdot_nll (embeddingHead + norm, models.py:136-137); 0 = raw CLS (dpr, models.py:210) */
  float ln_eps;        /* encoder LayerNorms (1e-5 RoBERTa, 1e-12 BERT) */
  float head_ln_eps;   /* nn.LayerNorm(768) default 1e-5 */
  int32_t pool_mean;   /* 0 = CLS pooling emb_all[0][:, 0] (every registered config, models.py:43); 1 = masked mean over the
                        * sequence's tokens (EmbeddingMixin.masked_mean, models.py:32-35, use_mean = True) */
} convdr_encoder_config;

typedef struct {       /* device pointers; w* are bf16 [out, in] row-major (nn.Linear layout), the rest fp32 */
  const void* wqkv;    /* [3H, H] = rows of attention.self.{query,key,value}.weight stacked */
  const float* bqkv;   /* [3H] */
  const void* wo;      /* attention.output.dense.weight [H, H] */
  const float* bo;
  const float *ln1_g, *ln1_b; /* attention.output.LayerNorm */
  const void* w1;      /* intermediate.dense.weight [I, H] */
  const float* b1;
  const void* w2;      /* output.dense.weight [H, I] */
  const float* b2;
  const float *ln2_g, *ln2_b; /* output.LayerNorm */
  /* Optional (NULL = absent): wo / w2 again in K-slice-major order [in / 32][H][32] (convdr_pack_kslice), read by the
   * fused projection + residual + LayerNorm kernel of the inference forward (hidden == 768, >= 24576 packed rows):
   * a 32-wide K slice of all H output rows is then one contiguous 48 KB run of whole cache lines.  Ignored by the
   * training entry points. */
  const void* wo_ks;
  const void* w2_ks;
} convdr_layer_weights;

typedef struct {
  const float *word_emb, *pos_emb, *type_emb; /* fp32 [vocab, H], [max_pos, H], [>=1, H] (row 0 used) */
  const float *emb_ln_g, *emb_ln_b;
  const convdr_layer_weights* layers;         /* HOST array of `layers` entries */
  const void* head_w;                         /* bf16 [out_dim, H] (out_dim > 0) */
  const float *head_b, *head_ln_g, *head_ln_b;
} convdr_encoder_weights;

/* fp32 -> bf16 (round to nearest even); n % 4 == 0.  Used to pack weights at load time / after an optimizer step. */
int convdr_cast_f32_bf16(const float* x, void* y, int64_t n, convdr_stream_t stream);

/* bf16 [n, k] row-major -> K-slice-major bf16 [k / 32][n][32]  (out[(s * n + r) * 32 + c] = w[r * k + 32 s + c]);
 * k % 32 == 0.  See convdr_layer_weights.wo_ks / w2_ks. */
int convdr_pack_kslice(const void* w_bf16, int n, int k, void* out, convdr_stream_t stream);

size_t convdr_encoder_workspace_bytes(const convdr_encoder_config* cfg, int64_t rows, int B);

/* Status word of an encoder forward: the first int32 of `workspace` (inference and training forward alike), zeroed at
 * the start of the call and OR-ed by the packing kernel.  The reference raises IndexError from nn.Embedding for a token
 * id outside the table (model/models.py:141-142); a kernel cannot raise, so the id is clamped to 0 (no out-of-bounds
 * read, no out-of-bounds atomic in the backward), the flag is set, and the host raises when it next reads the word
 * (the outputs of a flagged batch are meaningless). */
#define CONVDR_ENC_STATUS_BAD_TOKEN 1  /* a token id < 0 or >= cfg->vocab under the mask                       */
#define CONVDR_ENC_STATUS_BAD_LENS 2   /* seq_lens[b] != number of unmasked tokens of row b (or > L)          */
#define CONVDR_ENC_STATUS_BAD_MASK 4   /* attention_mask[b, 0] == 0: the CLS position must be a real token    */

/* out[b, :] = embedding of sequence b.  input_ids / attention_mask: device int64 [B, L] exactly as the reference
 * drivers pass them (gen_passage_embeddings.py:105-112); with ids_are_int32 != 0 input_ids is int32 [B, L] (token-cache
 * records, data/tokenizing.py:116) and attention_mask may be NULL = "l < seq_lens[b]" (what GetProcessingFn builds,
 * data/tokenizing.py:138-140).  Only mask == 1 tokens are computed ("packed rows"):
 * cu_seqlens (device int32 [B+1]) gives each sequence's first row, multiples of 8, cu[B] == rows;
 * seq_lens (device int32 [B]) = mask.sum(1) >= 1; mask[b, 0] must be 1 (the CLS position).  max_len = max(seq_lens).
 * out: device fp32 [B, out_dim or hidden]. */
int convdr_encoder_forward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                           const void* input_ids, int ids_are_int32, const int64_t* attention_mask, int B, int L,
                           const int32_t* cu_seqlens, const int32_t* seq_lens, int64_t rows, int max_len,
                           void* workspace, size_t workspace_bytes, float* out, convdr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training step of the student encoder: replaces, for /root/reference/drivers/run_convdr_train.py:109-191,
 *   embs = model(concat_ids, concat_id_mask)          -> convdr_encoder_train_forward (keeps activations)
 *   loss_fn(embs, teacher_embs)           :115, :460   -> convdr_mse_fwd_bwd
 *   logits / loss_fn_2(logits, labels)    :160-170     -> convdr_rank_ce_fwd_bwd
 *   loss.backward()                       :178         -> convdr_encoder_backward
 *   clip_grad_norm_(.., max_grad_norm)    :188-189     -> convdr_grad_norm_clip
 *   optimizer.step()  (HF AdamW, utils/dpr_utils.py:80-87) -> convdr_adamw_step
 * Dropout (run_convdr_train.py:107 model.train(): hidden / attention-probability dropout inside the HF encoder): torch's
 * RNG stream cannot be matched, so the mask is a documented counter-based function of (seed, site, layer, element)
 * (csrc/dropout.hpp, restated in oracle/dropout.py); the backward regenerates it.  NULL / p = 0: identity.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  float p_hidden;      /* config.hidden_dropout_prob: embeddings output, attention-output dense, FFN-output dense */
  float p_attention;   /* config.attention_probs_dropout_prob: softmax probabilities */
  uint32_t seed;       /* the SAME value for a forward and its backward */
} convdr_dropout;

typedef struct {   /* bf16 transposed weights for the data-gradient GEMMs (device pointers) */
  const void* wqkv_t;  /* [H, 3H] */
  const void* wo_t;    /* [H, H]  */
  const void* w1_t;    /* [H, I]  */
  const void* w2_t;    /* [I, H]  */
} convdr_layer_weights_t;

typedef struct {   /* fp32 gradient buffers, ACCUMULATED into (+=); same shapes as the parameters */
  float *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
} convdr_layer_grads;

typedef struct {
  float *word_emb, *pos_emb, *type_emb;   /* only row 0 of type_emb receives gradient */
  float *emb_ln_g, *emb_ln_b;
  const convdr_layer_grads* layers;       /* HOST array */
  float *head_w, *head_b, *head_ln_g, *head_ln_b;
} convdr_encoder_grads;

size_t convdr_encoder_train_workspace_bytes(const convdr_encoder_config* cfg, int64_t rows, int B);

/* Same contract as convdr_encoder_forward; additionally leaves every activation the backward needs in `workspace`
 * (which must stay untouched until convdr_encoder_backward has been enqueued). */
int convdr_encoder_train_forward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                                 const void* input_ids, int ids_are_int32, const int64_t* attention_mask, int B, int L,
                                 const int32_t* cu_seqlens, const int32_t* seq_lens, int64_t rows, int max_len,
                                 void* workspace, size_t workspace_bytes, float* out, const convdr_dropout* dropout,
                                 convdr_stream_t stream);

/* d_out: fp32 [B, out_dim or hidden] gradient of the loss w.r.t. the embeddings.  wt: HOST array of `layers`
 * entries; head_w_t: bf16 [hidden, out_dim] (NULL when out_dim == 0). */
int convdr_encoder_backward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                            const convdr_layer_weights_t* wt, const int32_t* cu_seqlens, const int32_t* seq_lens,
                            const void* head_w_t, int B, int64_t rows, int max_len, void* workspace,
                            size_t workspace_bytes, const float* d_out, const convdr_encoder_grads* grads,
                            const convdr_dropout* dropout, convdr_stream_t stream);

/* The same backward for gradient buffers NOBODY HAS WRITTEN YET (what autograd hands loss.backward() on the first backward after
 * zero_grad(): /root/reference/drivers/run_convdr_train.py:178,190): every gradient except the embedding tables is STORED by the one
 * kernel that completes it (weight-gradient tiles, the partial-sum reductions) instead of added to the buffer's contents, so
 * only grads->word_emb / pos_emb / type_emb / emb_ln_g / emb_ln_b must be zero on entry (the tables are scatter-added: token rows
 * repeat) and the other buffers may hold anything.  Bit-identical to convdr_encoder_backward on all-zero buffers; saves the
 * fill of, and one read of, every weight gradient (0.7 GB per step for roberta-base). */
int convdr_encoder_backward_fresh(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                                  const convdr_layer_weights_t* wt, const int32_t* cu_seqlens, const int32_t* seq_lens,
                                  const void* head_w_t, int B, int64_t rows, int max_len, void* workspace,
                                  size_t workspace_bytes, const float* d_out, const convdr_encoder_grads* grads,
                                  const convdr_dropout* dropout, convdr_stream_t stream);

/* One weight gradient of the backward above, exposed for parity tests at arbitrary shapes:
 *   dW[n, k] += sum_t dy[t, n] * x[t, k]     (what autograd's Linear backward computes for
 *   /root/reference/drivers/run_convdr_train.py:178; dy / x bf16 row-major with row strides ld_dy / ld_x, fp32 out)
 * straight from the token-major operands (TN MFMA engine, csrc/gemm_tn.hpp).  N, K and the strides are multiples of 8.
 * slab: fp32 scratch of slab_elems >= N * K elements (more lets the contraction be split across more workgroups; the
 * slices are summed in a fixed order: deterministic). */
int convdr_wgrad(const void* dy, int N, int64_t ld_dy, const void* x, int K, int64_t ld_x, int64_t rows, float* slab,
                 size_t slab_elems, float* dW, convdr_stream_t stream);

/* Gradient all-reduce under the backward (replaces what DistributedDataParallel's bucket hooks do for
 * /root/reference/drivers/run_convdr_train.py:52,178): makes `stream` wait until every gradient of encoder layer
 * `layer` written by the most recent convdr_encoder_backward on the current device is complete (the layers finish in
 * the order layers-1 .. 0; embeddings and head only with the whole call).  The caller then enqueues the collective for
 * that layer's slice of the gradient arena on `stream` while the backward of the layers below is still running.
 * layer = -1: the embedding tables and the embedding LayerNorm -- written by the call's last kernels on its own stream,
 * while the last weight-gradient branch may still be running (the call's stream only joins it afterwards). */
int convdr_backward_wait_layer(int layer, convdr_stream_t stream);

/* fp32 [n, k] row-major -> bf16 [k, n] (packing of the transposed weights) */
int convdr_transpose_f32_bf16(const float* x, int n, int k, void* y, convdr_stream_t stream);

/* Batched form: for i < count, fp32 [n[i], k[i]] at base + src_off[i] -> bf16 [k[i], n[i]] at out + dst_off[i]
 * (offsets in elements; src_off / n / k / dst_off are HOST arrays). */
int convdr_pack_transposed(const float* base, int count, const int64_t* src_off, const int32_t* n, const int32_t* k,
                           const int64_t* dst_off, void* out, convdr_stream_t stream);

/* The same from a bf16 source: bf16 [n[i], k[i]] at (bf16*)base + src_off[i] -> bf16 [k[i], n[i]].  For hosts that keep a bf16 copy
 * of the weights current (convdr_adamw_step_packed does): a third less traffic per training step than re-rounding the fp32 master
 * weights, same bits. */
int convdr_pack_transposed_bf16(const void* base, int count, const int64_t* src_off, const int32_t* n, const int32_t* k,
                                const int64_t* dst_off, void* out, convdr_stream_t stream);

/* loss[0] = mean((s - t)^2) over n elements (nn.MSELoss); ds (nullable) = grad_scale * 2 (s - t) / n */
int convdr_mse_fwd_bwd(const float* s, const float* t, int64_t n, float grad_scale, float* loss, float* ds,
                       convdr_stream_t stream);

/* logits[b, k] = <embs[b], docs[b, k]>, k = 0 is the positive; loss_per_query[b] = -log_softmax(logits[b])[0]
 * (nn.CrossEntropyLoss = their mean); d_embs (nullable) (+)= grad_scale / B * sum_k (softmax - onehot0) docs[b, k].
 * embs [B, E], docs [B, K, E] fp32, K <= 64. */
int convdr_rank_ce_fwd_bwd(const float* embs, const float* docs, int B, int K, int E, float grad_scale,
                           float* loss_per_query, float* d_embs, int accumulate, convdr_stream_t stream);

/* In-batch-negative form of the ranking loss (BASELINE configs[4]: the B x K teacher document embeddings of every rank
 * are all-gathered; NOT in the reference, whose formula above stays the default -- defined by
 * oracle/train.py:inbatch_rank_loss): logits[b, n] = <embs[b], docs[n]> over ALL N gathered documents,
 * loss_per_query[b] = -log_softmax(logits[b])[pos[b]] with pos[b] the row of query b's own positive document;
 * d_embs (nullable) (+)= grad_scale / B * sum_n (softmax - onehot(pos[b])) docs[n].  embs [B, E], docs [N, E] fp32,
 * pos device int32 [B], N <= 16384. */
/* The pairwise NLL of the model surface -- NLL.forward(q, a, b) (/root/reference/model/models.py:66-75; BiEncoder.forward
 * :254-262) and its MaxP form NLL_MultiChunk.forward (:92-126) -- with gradients for ALL three inputs (a and b are student
 * outputs here, not frozen teacher embeddings):  s_x[i] = max_c(<q[i], x[i, c]> + bias_x[i, c]) for x in {a, b} (C chunks per
 * document; C = 1 and NULL biases: the plain pairwise form; the first maximal chunk wins, like torch.max),
 * loss_per_query[i] = -log_softmax([s_a, s_b])[0]; d_q / d_a / d_b (each nullable) = grad_scale / B * d(sum_i loss_i) / d(.),
 * overwritten (non-selected chunks get zeros).  q [B, E], a / b [B, C, E], biases [B, C] fp32; C <= 32. */
int convdr_pair_nll_fwd_bwd(const float* q, const float* a, const float* b, const float* bias_a, const float* bias_b, int B,
                            int C, int E, float grad_scale, float* loss_per_query, float* d_q, float* d_a, float* d_b,
                            convdr_stream_t stream);

int convdr_inbatch_ce_fwd_bwd(const float* embs, const float* docs, int B, int N, int E, const int32_t* pos,
                              float grad_scale, float* loss_per_query, float* d_embs, int accumulate,
                              convdr_stream_t stream);

/* torch.nn.utils.clip_grad_norm_ over one flat fp32 gradient buffer whose entries still have to be multiplied by
 * pre_scale (1 / world size after a SUM all-reduce; 1 otherwise):  norm_and_coef[0] = ||pre_scale * g||_2,
 * norm_and_coef[1] = pre_scale * min(1, max_norm / (norm + 1e-6)) = the factor the stored gradients are multiplied by;
 * apply != 0 does that in place, otherwise the caller hands norm_and_coef + 1 to convdr_adamw_step (grad_scale).
 * scratch: >= 1024 floats. */
int convdr_grad_norm_clip(float* grads, int64_t n, float max_norm, float pre_scale, float* scratch, float* norm_and_coef,
                          int apply, convdr_stream_t stream);

/* The two halves of convdr_grad_norm_clip, for callers that sum a gradient arena piece by piece -- e.g. every encoder
 * layer's slice on a side stream as soon as convdr_backward_wait_layer says it is complete, under the backward of the
 * layers below, so that only the last pieces are summed after the backward:
 *   convdr_grad_sumsq:       partials[b] = sum of squares of block b's share of x[0, n), b < nblocks (<= 1024)
 *   convdr_grad_norm_finish: norm_and_coef as convdr_grad_norm_clip, from `count` partial sums (any order of pieces: the
 *                            finish folds them in index order in fp64 -- deterministic for a fixed layout) */
int convdr_grad_sumsq(const float* x, int64_t n, float* partials, int nblocks, convdr_stream_t stream);
int convdr_grad_norm_finish(const float* partials, int count, float max_norm, float pre_scale, float* norm_and_coef,
                            convdr_stream_t stream);

/* The stream convdr_encoder_backward runs its weight-gradient branches on (default: one it creates itself).  HIP multiplexes
 * streams onto GPU_MAX_HW_QUEUES hardware queues in order of first use; a caller that has verified that `stream` runs
 * concurrently with its compute stream hands it in here.  Per device (the stream must belong to the CURRENT device, else
 * an error is returned); may be called again between two backward calls to move the branch to another stream (every
 * backward ends with the caller's stream waiting for the branch, so nothing is pending in between). */
int convdr_train_set_side_stream(convdr_stream_t stream);

/* x[i] *= scale[0] (device scalar), e.g. the clip coefficient */
int convdr_scale_f32(float* x, int64_t n, const float* scale, convdr_stream_t stream);

/* transformers==2.3.0 AdamW on flat fp32 buffers (NOT torch.optim.AdamW: eps is added to sqrt(v) before the bias
 * correction, decoupled weight decay is applied after the update).  grad_scale: optional device scalar multiplied
 * into g first (the clip coefficient).  step >= 1. */
int convdr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, int correct_bias, const float* grad_scale,
                      convdr_stream_t stream);
/* The same update, also refreshing the packed bf16 copy of the weights that the encoder GEMMs read (what
 * convdr_cast_f32_bf16 over p[bf16_first, n) would produce afterwards): bf16_copy[i - bf16_first] = bf16(p[i]) for
 * i >= bf16_first, written from the registers that hold the new weight -- the training step then has no cast pass over
 * the parameter arena.  bf16_copy == NULL: exactly convdr_adamw_step.  bf16_first % 4 == 0. */
int convdr_adamw_step_packed(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int step, int correct_bias, const float* grad_scale,
                             void* bf16_copy, int64_t bf16_first, convdr_stream_t stream);

/* Tuning / test knobs: "fused_ln_min_rows" = minimum packed rows for the fused GEMM + residual + LayerNorm kernel
 * (default 24576; tests lower it to exercise that kernel on small inputs); "fused_ln_max_k" = largest contraction
 * length it is used for; "hm_blocked" = 0 / 1: row-major / blocked layout of the FFN activation between FFN1 and the
 * fused FFN2 + LayerNorm kernel (default 1; a workspace-internal choice, results are identical); "attn_bwd_fused" = 1 / 0:
 * training backward of the attention in one workgroup per (sequence, head) for sequences of at most 256 tokens / always the
 * dQ kernel + the dK, dV kernel (default 1); "gelu_gp" = 1 / 0: gelu' evaluated in the training forward's FFN1 epilogue / the
 * pre-activation saved and a separate pass in the backward; "ip_fused_finish" = 1 / 0: one / three launches behind a scan;
 * "ffn2_splitk" = 1 / 0: the K = 3072 projection of at most 5,376 packed rows as 4 / 2 contraction slices + a finishing row
 * kernel / as whole-contraction tiles (default 1; fp32 summation order differs);
 * "ln_rows", "ln_bwd_rows" = straight-line LayerNorm forward (1 / 0) and backward (2 / 1 / 0) kernels for hidden size 768 against
 * the general ones (same formulas, rounding-level differences); "gemm_trace" / "gemm_trace_ln" = device buffer for the s_memtime phase stamps of a
 * `make TRACE=1` build (tools/gemm_trace*.py; 0 = off). */
int convdr_set_option(const char* name, int64_t value);

/* Test aid: byte offsets of the activation buffers inside the encoder workspace, in the order tok_id, tok_pos, X, Q, K,
 * Vt, ctx, Hm, Y, cls_b, cls_y, cls_f, head_y; out[13] = leading dimension of Vt. */
int convdr_encoder_debug_layout(const convdr_encoder_config* cfg, int64_t rows, int B, int64_t* out);

/* ------------------------------------------------------------------------------------------
 * Collectives of the N > 1 paths for hosts that are not Python: thin wrappers over RCCL (xGMI).
 * The reference's own interface for these exchanges is torch.distributed (gen_passage_embeddings.py:314, DDP /
 * nn.DataParallel run_convdr_train.py:77-78, faiss IndexShards run_convdr_inference.py:355-368), which
 * convdr_amd/parallel.py keeps using; a torch-free host gets the same three steps here:
 *   query all-gather            convdr_comm_allgather(comm, Q_local, Q_all, nq_local * d * 4, stream)
 *   per-rank top-k all-gather   convdr_comm_allgather(comm, packed (score, offset) lists, all lists, nq * k * 12, stream)
 *                               followed by W - 1 convdr_topk_merge calls
 *   gradient all-reduce (sum)   convdr_comm_allreduce_f32(comm, arena slice, arena slice, count, stream) behind
 *                               convdr_backward_wait_layer; 1 / W rides on convdr_grad_norm_clip(pre_scale)
 * One communicator per process and device (one process per GPU).  librccl.so is looked up at the first call (the copy
 * the process has already loaded, e.g. torch's; else the system one): it is not a link-time dependency.
 * convdr_comm_unique_id fills CONVDR_COMM_ID_BYTES HOST bytes on one rank; the host distributes them out of band (file, MPI,
 * socket) and every rank calls convdr_comm_init with them.  send / recv are device pointers; in-place use
 * (recv + rank * bytes == send, or recv == send for the all-reduce) is allowed as in RCCL.
 * ------------------------------------------------------------------------------------------ */
#define CONVDR_COMM_ID_BYTES 128
typedef struct convdr_comm_opaque* convdr_comm_t;
int convdr_comm_unique_id(void* id_out_host);
int convdr_comm_init(convdr_comm_t* comm, int nranks, int rank, const void* unique_id_host);
int convdr_comm_ranks(convdr_comm_t comm, int* nranks, int* rank);
int convdr_comm_allgather(convdr_comm_t comm, const void* send, void* recv, size_t bytes_per_rank, convdr_stream_t stream);
int convdr_comm_allreduce_f32(convdr_comm_t comm, const float* send, float* recv, size_t count, convdr_stream_t stream);
int convdr_comm_destroy(convdr_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* CONVDR_HIP_H */
