// Host orchestration + C ABI of the dual-encoder forward (inference path).
// Reference call sites replaced:  model/models.py:140-148 (RobertaDot_NLL_LN.query_emb/body_emb),
// :205-211 + :227-235 (HFBertEncoder / BiEncoder.query_emb/body_emb).
#include "gemm_launch.hpp"
#include "gemm_ln.hpp"

#include "../../include/convdr_hip.h"

namespace convdr {

struct EncBufs {
  int32_t* status;   // workspace offset 0: CONVDR_ENC_STATUS_* flags of the last forward
  int32_t *tok_id, *tok_pos;
  bf16_t *X, *Q, *K, *Vt, *ctx, *Hm, *cls_b, *cls_ctx, *cls_x, *cls_x1;
  float *Y, *cls_y, *cls_f, *head_y;
  float* slab;       // split-contraction partial sums of the few-rows FFN2 (null above SPLITK_MAX_ROWS rows)
  int64_t ldt;
  size_t total;
};

// Few rows: the K = 3072 projection is cut into up to SPLITK_MAX_SPLIT contraction slices (gemm_resid_ln below) while its
// 128 x 128 tiles fill at most half of the chip's 512 workgroup slots -- 42 token tiles x 6 feature tiles.
constexpr int64_t SPLITK_MAX_ROWS = 42 * 128;
constexpr int SPLITK_MAX_SPLIT = 4;
static int64_t g_ffn2_splitk = 1;   // convdr_set_option("ffn2_splitk", 0): whole-contraction tiles (A/B, tests)

static EncBufs enc_plan(const convdr_encoder_config* c, int64_t rows, int B, char* base) {
  EncBufs p;
  size_t o = 0;
  // (integer arithmetic: the size-only call plans from a null base, and pointer arithmetic on null is undefined -- UBSan, make SAN=1)
  auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return (char*)((uintptr_t)base + at); };
  const int H = c->hidden, I = c->intermediate;
  const int64_t rs = rows + 128;  // slack: attention K tiles / clamped reads
  p.ldt = align_up((size_t)rows + 64, 8);
  p.status = (int32_t*)take(256);
  p.tok_id = (int32_t*)take(rs * 4);
  p.tok_pos = (int32_t*)take(rs * 4);
  p.X = (bf16_t*)take(rs * H * 2);
  p.Q = (bf16_t*)take(rs * H * 2);
  p.K = (bf16_t*)take(rs * H * 2);
  p.Vt = (bf16_t*)take((size_t)H * p.ldt * 2);
  p.ctx = (bf16_t*)take(rs * H * 2);
  p.Hm = (bf16_t*)take(rs * I * 2);
  p.Y = (float*)take(rs * H * 4);
  p.slab = rows <= SPLITK_MAX_ROWS ? (float*)take((size_t)SPLITK_MAX_SPLIT * rs * H * 4) : nullptr;
  const int64_t Bp = B + 128;
  p.cls_b = (bf16_t*)take(Bp * H * 2);
  p.cls_ctx = (bf16_t*)take(Bp * H * 2);
  p.cls_x = (bf16_t*)take(Bp * H * 2);
  p.cls_x1 = (bf16_t*)take(Bp * H * 2);
  p.cls_y = (float*)take(Bp * H * 4);
  p.cls_f = (float*)take(Bp * H * 4);
  p.head_y = (float*)take(Bp * (c->out_dim > 0 ? c->out_dim : 1) * 4);
  p.total = o;
  return p;
}

static int64_t g_fused_ln_max_k = 1 << 30;
static int64_t g_hm_blocked = 1;    // blocked activation hand-offs; convdr_set_option("hm_blocked", 0) gives the row-major paths back
static int64_t g_fused_ln_min_rows = 128 * 192;   // below this the 128-token tiles cannot fill the 256 CUs

// Y = A W^T + bias + R, X = LayerNorm(Y): fused row-complete kernel for hidden size 768 and enough rows to fill the
// chip, else GEMM (fp32 sums) + LayerNorm kernel.  `Yf` is the fp32 scratch of the unfused path.
static bool fused_ln_applies(int64_t rows, int H, int K) {
  return H == 768 && rows >= g_fused_ln_min_rows && K % LN_SLICE == 0 && K <= g_fused_ln_max_k;
}
// a_blocked: A is in the EPI_GELU_BLK layout (only ever set when fused_ln_applies)
static int gemm_resid_ln(const bf16_t* W, const bf16_t* Wks, const bf16_t* A, int64_t rows, int H, int K, const float* bias,
                         const bf16_t* R, const float* gamma, const float* beta, float eps, float* Yf, bf16_t* X,
                         const char* name, hipStream_t st, bool a_blocked = false, float* slab = nullptr) {
  // measured at 262k rows: K = 768: 0.52 ms fused vs 0.49 + 0.19 ms (GEMM + LayerNorm); K = 3072: 1.33 vs 1.17 + 0.19 ms
  if (fused_ln_applies(rows, H, K)) {
    static DeviceOnce attr_done;
    if (attr_done.first())
      CONVDR_CHECK_HIP(
          hipFuncSetAttribute((const void*)k_gemm_resid_ln, hipFuncAttributeMaxDynamicSharedMemorySize, LN_SMEM_BYTES));
    GemmLnArgs a{W, A, rows, K, bias, R, gamma, beta, eps, X,
                 K == 768 ? (unsigned long long*)g_gemm_trace_ln : nullptr, Wks, a_blocked ? 1 : 0, 0};
#ifdef CONVDR_ENABLE_TRACE   // timing experiment that produces garbage results: only in the `make TRACE=1` library
    static const int dbg_epi = getenv("CONVDR_DBG_SKIP_EPI") ? atoi(getenv("CONVDR_DBG_SKIP_EPI")) : 0;
    a.dbg_skip_epi = dbg_epi == 1;
#endif
    ProfScope prof(name, st);
    hipLaunchKernelGGL(k_gemm_resid_ln, dim3((unsigned)ceil_div64(rows, TileLN::TL)), dim3(512), LN_SMEM_BYTES, st, a);
    CONVDR_CHECK_LAUNCH("k_gemm_resid_ln");
    return 0;
  }
  GemmArgs g{};
  // Few rows, long contraction (the frozen teacher's targets in a training step, a query batch of the evaluation loop): the
  // 128 x 128 tiles of a whole-contraction launch fill a quarter of the chip and each is a 48-step latency chain (126
  // workgroups, 48 us at 2.6 k rows); cut into 4 (2) slices while tiles x slices fit the 512 workgroup slots, and finished --
  // partial sums + bias + residual + LayerNorm -- by one row kernel: 17 + 6 us.  The sum order differs from the whole-
  // contraction tile's (fp32 rounding level).
  int ns = 1;
  if (slab && g_ffn2_splitk && H == 768 && K >= 2048 && rows <= SPLITK_MAX_ROWS) {
    const int64_t tiles = ceil_div64(rows, 128) * (H / 128), slots = (int64_t)device_cu_count() * 2;
    for (int c = SPLITK_MAX_SPLIT; c >= 2; c >>= 1)
      if (tiles * c <= slots && K % (c * GEMM_BK) == 0) { ns = c; break; }
  }
  if (ns > 1) {
    g.rows = rows; g.W = W; g.X = A; g.N = H; g.K = K; g.k_split_len = K / ns; g.Cf = slab;
    if (int e = launch_gemm<EPI_SLAB_F32>(g, st, name)) return e;
    ProfScope prof("layernorm", st);
    hipLaunchKernelGGL(k_slab_finish_ln<3>, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, slab, ns, rows, bias, R, gamma, beta,
                       eps, X);
    CONVDR_CHECK_LAUNCH("k_slab_finish_ln");
    return 0;
  }
  g.rows = rows; g.W = W; g.X = A; g.N = H; g.K = K; g.bias = bias; g.Cf = Yf; g.R = R;
  if (int e = launch_gemm<EPI_RESID_F32>(g, st, name)) return e;
  ProfScope prof("layernorm", st);
  launch_layernorm_bf16(Yf, rows, H, gamma, beta, eps, X, st);
  CONVDR_CHECK_LAUNCH("k_layernorm");
  return 0;
}

static int check_config(const convdr_encoder_config* c) {
  CONVDR_REQUIRE(c->hidden % 128 == 0 && c->hidden <= 1024, "encoder: hidden must be a multiple of 128, <= 1024 (got %d)",
                 c->hidden);
  CONVDR_REQUIRE(c->heads * 64 == c->hidden, "encoder: head_dim must be 64 (hidden=%d heads=%d)", c->hidden, c->heads);
  CONVDR_REQUIRE(c->intermediate % 64 == 0, "encoder: intermediate %% 64 != 0 (%d)", c->intermediate);
  CONVDR_REQUIRE(c->out_dim == 0 || (c->out_dim % 4 == 0 && c->out_dim <= 1024), "encoder: bad out_dim %d", c->out_dim);
  return 0;
}

// One transformer layer on packed rows; buffers may be shared across layers (inference).
// cls_only (last layer): only the CLS rows of the output are live (models.py:43), so after attention the CLS rows of
// ctx and of the layer input are gathered into compact [B, H] buffers and the output projection, LayerNorm and FFN run
// on B rows instead of `rows`; K and V still need every token, but Q only the CLS rows (a B-row GEMM) and the attention
// runs its CLS_Q form: saves ~10/12 of the last layer's GEMM work and half of its attention traffic.
// On return: p.X holds the layer output (LayerNorm2 applied); when cls_only, p.Y[0..B) holds the pre-LayerNorm2 sums of
// the B CLS rows instead.
int encoder_layer_forward(const convdr_encoder_config* c, const convdr_layer_weights* w, const EncBufs& p,
                          const int32_t* cu, const int32_t* lens, int64_t rows, int B, int max_len, float* lse,
                          bool cls_only, hipStream_t st) {
  const int H = c->hidden, I = c->intermediate;
  GemmArgs g{};
  g.rows = rows;
  // fused QKV projection: Q, K token-major, V feature-major
  g.W = (const bf16_t*)w->wqkv; g.X = p.X; g.N = 3 * H; g.K = H; g.bias = w->bqkv;
  g.Qo = p.Q; g.Ko = p.K; g.Vt = p.Vt; g.H = H; g.ldt = p.ldt;
  const bf16_t *xin = p.X, *ctx = p.ctx;
  bf16_t* x1 = p.X;
  int64_t n = rows;
  // blocked activation layouts between a producer's registers and the row-complete projection + LayerNorm kernel
  // (EPI_GELU_BLK, k_attention_fwd<CTX_BLK>): on whenever that kernel serves the shape.
  // convdr_set_option("hm_blocked", 0): the row-major round-2 paths
  const bool blk_on = g_hm_blocked != 0;
  const bool ctx_blocked = blk_on && !cls_only && fused_ln_applies(rows, H, H);
  const bool qk_blocked = blk_on && fused_ln_applies(rows, H, H);   // Q / K between the QKV projection and the attention
  g.qk_blocked = qk_blocked ? 1 : 0;
  if (!cls_only) {
    if (int e = launch_gemm<EPI_QKV>(g, st, "gemm_qkv")) return e;
    AttnArgs a{p.Q, p.K, p.Vt, p.ldt, cu, lens, H, (int64_t)H, p.ctx, lse, 0.125f, (unsigned long long*)g_attn_trace};
    ProfScope prof("attention", st);
    if (ctx_blocked && qk_blocked) {
      hipLaunchKernelGGL((k_attention_fwd<false, true, true>), dim3((max_len + 127) / 128, c->heads, B), dim3(256), ATT_SMEM_BYTES, st, a);
    } else {
      hipLaunchKernelGGL(k_attention_fwd<false>, dim3((max_len + 127) / 128, c->heads, B), dim3(256), ATT_SMEM_BYTES, st, a);
    }
    CONVDR_CHECK_LAUNCH("k_attention_fwd");
  } else {
    // K and V of every token, Q of the B CLS rows only (a third of the projection and the Q / ctx traffic of the
    // attention saved); the CLS queries land at the head of the otherwise unused Q buffer
    hipLaunchKernelGGL(k_gather_cls, dim3((B + 3) / 4), dim3(256), 0, st, cu, B, H, p.X, (const float*)nullptr, p.cls_x,
                       (float*)nullptr);
    CONVDR_CHECK_LAUNCH("k_gather_cls");
    g.W = (const bf16_t*)w->wqkv + (size_t)H * H; g.bias = w->bqkv + H; g.N = 2 * H; g.third0 = 1;
    if (int e = launch_gemm<EPI_QKV>(g, st, "gemm_qkv")) return e;
    GemmArgs gq{};
    gq.rows = B; gq.W = (const bf16_t*)w->wqkv; gq.X = p.cls_x; gq.N = H; gq.K = H; gq.bias = w->bqkv; gq.Cb = p.Q;
    if (int e = launch_gemm<EPI_BF16>(gq, st, "gemm_qkv")) return e;
    AttnArgs a{p.Q, p.K, p.Vt, p.ldt, cu, lens, H, (int64_t)H, p.cls_ctx, nullptr, 0.125f, nullptr};
    ProfScope prof("attention", st);
    if (qk_blocked) hipLaunchKernelGGL((k_attention_fwd<true, false, true>), dim3(1, c->heads, B), dim3(256), ATT_SMEM_BYTES, st, a);
    else hipLaunchKernelGGL(k_attention_fwd<true>, dim3(1, c->heads, B), dim3(256), ATT_SMEM_BYTES, st, a);
    CONVDR_CHECK_LAUNCH("k_attention_fwd<cls>");
    xin = p.cls_x; ctx = p.cls_ctx; x1 = p.cls_x1; n = B;
  }
  // attention output dense + residual + LayerNorm -> X1
  if (int e = gemm_resid_ln((const bf16_t*)w->wo, (const bf16_t*)w->wo_ks, ctx, n, H, H, w->bo, xin, w->ln1_g, w->ln1_b, c->ln_eps, p.Y, x1,
                            "gemm_attn_out", st, ctx_blocked))
    return e;
  // FFN
  GemmArgs g2{};
  g2.rows = n; g2.W = (const bf16_t*)w->w1; g2.X = x1; g2.N = I; g2.K = H; g2.bias = w->b1; g2.Cb = p.Hm;
  // FFN1 -> FFN2 through the blocked activation layout (EPI_GELU_BLK) whenever FFN2 is the row-complete kernel that can
  // read it: the 1.6 GB tile output of FFN1 then leaves its registers in whole lines without passing through LDS.
  // (CONVDR_HM_BLOCKED=0: row-major Hm, the round-2 path; A/B switch)
  const bool blocked = blk_on && !cls_only && fused_ln_applies(n, H, I) && I % 256 == 0 && (I / 256) * ceil_div64(n, 256) >= 192;
  if (blocked) {
    if (int e = launch_gemm<EPI_GELU_BLK>(g2, st, "gemm_ffn1")) return e;
    return gemm_resid_ln((const bf16_t*)w->w2, (const bf16_t*)w->w2_ks, p.Hm, n, H, I, w->b2, x1, w->ln2_g, w->ln2_b, c->ln_eps, p.Y,
                         p.X, "gemm_ffn2", st, true);
  }
  if (int e = launch_gemm<EPI_GELU_BF16>(g2, st, "gemm_ffn1")) return e;
  if (cls_only) {   // final embedding: fp32 pre-LN sums for the B CLS rows, LayerNorm'ed by the caller
    g2 = GemmArgs{};
    g2.rows = n; g2.W = (const bf16_t*)w->w2; g2.X = p.Hm; g2.N = H; g2.K = I; g2.bias = w->b2; g2.Cf = p.Y; g2.R = x1;
    return launch_gemm<EPI_RESID_F32>(g2, st, "gemm_ffn2");
  }
  return gemm_resid_ln((const bf16_t*)w->w2, (const bf16_t*)w->w2_ks, p.Hm, n, H, I, w->b2, x1, w->ln2_g, w->ln2_b, c->ln_eps, p.Y, p.X, "gemm_ffn2", st,
                       false, p.slab);
}

}  // namespace convdr

using namespace convdr;

extern "C" size_t convdr_encoder_workspace_bytes(const convdr_encoder_config* cfg, int64_t rows, int B) {
  return enc_plan(cfg, rows, B, nullptr).total;
}

extern "C" int convdr_cast_f32_bf16(const float* x, void* y, int64_t n, convdr_stream_t stream) {
  CONVDR_REQUIRE(n % 4 == 0, "convdr_cast_f32_bf16: n %% 4 != 0 (%lld)", (long long)n);
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)(ceil_div64(n4, 256) < 4096 ? ceil_div64(n4, 256) : 4096);
  hipLaunchKernelGGL(k_cast_f32_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n4);
  CONVDR_CHECK_LAUNCH("k_cast_f32_bf16");
  return 0;
}

extern "C" int convdr_encoder_forward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                                      const void* input_ids, int ids_are_int32, const int64_t* attention_mask, int B, int L,
                                      const int32_t* cu_seqlens, const int32_t* seq_lens, int64_t rows, int max_len,
                                      void* workspace, size_t workspace_bytes, float* out, convdr_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int e = check_config(cfg)) return e;
  CONVDR_REQUIRE(B > 0 && L > 0 && rows > 0 && rows % 8 == 0 && max_len > 0 && max_len <= L,
                 "convdr_encoder_forward: bad sizes B=%d L=%d rows=%lld max_len=%d", B, L, (long long)rows, max_len);
  const EncBufs p = enc_plan(cfg, rows, B, (char*)workspace);
  CONVDR_REQUIRE(workspace_bytes >= p.total, "convdr_encoder_forward: workspace too small (%zu < %zu)", workspace_bytes,
                 p.total);
  const int H = cfg->hidden;
  // V^T columns past the last row are read (never used) by the last key tile: keep them finite
  CONVDR_CHECK_HIP(hipMemset2DAsync(p.Vt + rows, p.ldt * 2, 0, (p.ldt - rows) * 2, H, st));
  CONVDR_CHECK_HIP(hipMemsetAsync(p.status, 0, 256, st));
  hipLaunchKernelGGL(k_seq_pack, dim3((B + 3) / 4), dim3(256), 0, st, input_ids, ids_are_int32, attention_mask, seq_lens, B, L, cu_seqlens,
                     cfg->kind, cfg->pad_idx, cfg->max_pos, cfg->vocab, p.tok_id, p.tok_pos, p.status);
  CONVDR_CHECK_LAUNCH("k_seq_pack");
  {
    ProfScope prof("embed_ln", st);
    hipLaunchKernelGGL(k_embed_ln, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, p.tok_id, p.tok_pos, rows, H,
                       w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, cfg->ln_eps, p.X, DropSite{0u, 0u, 1.f});
    CONVDR_CHECK_LAUNCH("k_embed_ln");
  }
  for (int l = 0; l < cfg->layers; ++l) {
    const convdr_layer_weights* lw = &w->layers[l];
    const bool last = l + 1 == cfg->layers;
    if (int e = encoder_layer_forward(cfg, lw, p, cu_seqlens, seq_lens, rows, B, max_len, nullptr, last && !cfg->pool_mean, st))
      return e;
    if (last && cfg->pool_mean) {
      // use_mean = True (models.py:32-35, :40-41): the whole last layer is live; p.X holds its LayerNorm'ed output
      hipLaunchKernelGGL(k_masked_mean, dim3(B), dim3(256), 0, st, p.X, cu_seqlens, seq_lens, H, p.cls_b,
                         cfg->out_dim > 0 ? p.cls_f : out);
      CONVDR_CHECK_LAUNCH("k_masked_mean");
    } else if (last) {
      // the last layer ran its tail on the CLS rows only: p.Y[0..B) are their pre-LN sums
      float* cls_out = cfg->out_dim > 0 ? p.cls_f : out;
      hipLaunchKernelGGL(k_layernorm, dim3((B + 3) / 4), dim3(256), 0, st, p.Y, (int64_t)B, H, lw->ln2_g, lw->ln2_b,
                         cfg->ln_eps, p.cls_b, cls_out);
      CONVDR_CHECK_LAUNCH("k_layernorm(cls)");
    }
  }
  if (cfg->out_dim > 0) {  // rdot_nll head: LayerNorm(Linear(H, out_dim)(cls))   models.py:144
    GemmArgs g{};
    g.rows = B; g.W = (const bf16_t*)w->head_w; g.X = p.cls_b; g.N = cfg->out_dim; g.K = H; g.bias = w->head_b;
    g.Cf = p.head_y;
    if (int e = launch_gemm<EPI_F32>(g, st, "gemm_head")) return e;
    hipLaunchKernelGGL(k_layernorm, dim3((B + 3) / 4), dim3(256), 0, st, p.head_y, (int64_t)B, cfg->out_dim,
                       w->head_ln_g, w->head_ln_b, cfg->head_ln_eps, (bf16_t*)nullptr, out);
    CONVDR_CHECK_LAUNCH("k_layernorm(head)");
  }
  return 0;
}

// Debug / test aid: byte offsets of the activation buffers inside the workspace, in the order
// tok_id, tok_pos, X, Q, K, Vt, ctx, Hm, Y, cls_b, cls_y, cls_f, head_y; out[13] = ldt.
extern "C" int convdr_encoder_debug_layout(const convdr_encoder_config* cfg, int64_t rows, int B, int64_t* out) {
  const EncBufs p = enc_plan(cfg, rows, B, nullptr);
  const void* v[13] = {p.tok_id, p.tok_pos, p.X, p.Q, p.K, p.Vt, p.ctx, p.Hm, p.Y, p.cls_b, p.cls_y, p.cls_f, p.head_y};
  for (int i = 0; i < 13; ++i) out[i] = (int64_t)(size_t)v[i];
  out[13] = p.ldt;
  return 0;
}

// Tuning / test knobs.  "fused_ln_min_rows": minimum packed rows for the fused GEMM + LayerNorm kernel.
namespace convdr {
// 16 bytes per thread: chunk c (8 elements) of row r in slice s
static __global__ void __launch_bounds__(256) k_pack_kslice(const bf16_t* __restrict__ w, int n, int k, bf16_t* __restrict__ out) {
  const int64_t total = (int64_t)n * (k / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i & 3);
    const int64_t sr = i >> 2;                 // s * n + r
    const int r = (int)(sr % n), s = (int)(sr / n);
    *(uint4*)(out + i * 8) = *(const uint4*)(w + (int64_t)r * k + s * 32 + c * 8);
  }
}
}  // namespace convdr

extern "C" int convdr_pack_kslice(const void* w_bf16, int n, int k, void* out, convdr_stream_t stream) {
  CONVDR_REQUIRE(n > 0 && k > 0 && k % 32 == 0, "convdr_pack_kslice: bad shape %d x %d", n, k);
  const int64_t total = (int64_t)n * (k / 8);
  hipLaunchKernelGGL(k_pack_kslice, dim3((unsigned)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)w_bf16, n, k, (bf16_t*)out);
  CONVDR_CHECK_LAUNCH("k_pack_kslice");
  return 0;
}

extern "C" int convdr_set_option(const char* name, int64_t value) {
  if (strcmp(name, "fused_ln_min_rows") == 0) {
    g_fused_ln_min_rows = value;
    return 0;
  }
  if (strcmp(name, "hm_blocked") == 0) {   // FFN1 -> FFN2 activation layout (EPI_GELU_BLK): 1 blocked, 0 row-major
    g_hm_blocked = value;
    return 0;
  }
  if (strcmp(name, "gemm_tile_policy") == 0) {   // 0 cost model, 1 / 2 / 3 force 256 x 256 / 256 x 128 / 128 x 128 (gemm_launch.hpp)
    g_gemm_tile_policy = value;
    return 0;
  }
  if (strcmp(name, "attn_bwd_fused") == 0) {   // training: 1 = one-workgroup attention backward for sequences <= 256 tokens
    g_attn_bwd_fused = value;
    return 0;
  }
  if (strcmp(name, "embed_bwd_deterministic") == 0) {   // training: 1 = embedding-table gradients without atomics (k_embed_scatter_det)
    g_embed_bwd_det = value;
    return 0;
  }
  if (strcmp(name, "ip_fused_finish") == 0) {   // search: 1 = cut + re-score + select in one launch per search (k_ip_finish), 0 = three launches
    convdr::g_ip_fused_finish = value;
    return 0;
  }
  if (strcmp(name, "gelu_gp") == 0) {   // training: 1 = gelu' evaluated in the forward's FFN1 epilogue, multiplied in the FFN2 dgrad epilogue
    g_gelu_gp = value;
    return 0;
  }
  if (strcmp(name, "ffn2_splitk") == 0) {   // few-rows FFN2: 1 (default) split contraction + finishing row kernel, 0 whole-contraction tiles
    g_ffn2_splitk = value;
    return 0;
  }
  if (strcmp(name, "ln_rows") == 0) {   // forward LayerNorm of H = 768 rows: 1 (default) straight-line kernel, 0 the general one
    g_ln_rows = value;
    return 0;
  }
  if (strcmp(name, "ln_bwd_rows") == 0) {   // training: LayerNorm backward kernel of the H = 768 encoder layers (gemm_launch.hpp)
    g_ln_bwd_rows = value;
    return 0;
  }
  if (strcmp(name, "fused_ln_max_k") == 0) {
    g_fused_ln_max_k = value;
    return 0;
  }
  if (strcmp(name, "clock_probe") == 0) {   // measurement: device buffer of 4 uint64 for the FFN1 kernel's clock stamps (0 = off)
    g_clock_probe = (void*)(uintptr_t)value;
    g_clock_probe_next = 0;
    return 0;
  }
  if (strcmp(name, "clock_probe_slots") == 0) {   // number of 4 x uint64 slots behind "clock_probe" (set it first)
    g_clock_probe_slots = value > 0 ? value : 1;
    return 0;
  }
  if (strcmp(name, "attn_trace") == 0) {
    g_attn_trace = (void*)(uintptr_t)value;
    return 0;
  }
  if (strcmp(name, "gemm_trace_ln") == 0) {   // same for k_gemm_resid_ln (the K = 768 launches)
    g_gemm_trace_ln = (void*)(uintptr_t)value;
    return 0;
  }
  if (strcmp(name, "gemm_trace") == 0) {   // experiment: device buffer for k_gemm phase stamps (0 = off)
    g_gemm_trace = (void*)(uintptr_t)value;
    return 0;
  }
  set_error("convdr_set_option: unknown option %s", name);
  return -1;
}
