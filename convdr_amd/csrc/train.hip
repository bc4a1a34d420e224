// Training path of the dual encoder: forward that keeps activations, full backward, losses, grad-norm clip and
// HF-AdamW.  Replaces, for the student model, what
//   embs = model(concat_ids, concat_id_mask) ... loss.backward() ... clip_grad_norm_ ... optimizer.step()
// (/root/reference/drivers/run_convdr_train.py:109-191) execute through torch autograd + HF AdamW.
#include "gemm_launch.hpp"
#include "gemm_tn.hpp"
#include "train_kernels.hpp"
#include "attention_train.hpp"

#include "../../include/convdr_hip.h"

namespace convdr {

struct LayerSave {
  bf16_t *Xin, *QKV, *ctx, *X1, *Hpre, *Hm;
  float *LSE, *Y1, *Y2;
  uint32_t* Mbits;   // [heads][ATTM_PIECES][ldt]: dropout keep bits of the attention probabilities (attention_train.hpp)
};

// bf16 activation gradients of one layer that its weight-gradient products read (kept per layer: the weight-gradient
// branch runs on its own stream, whole layers behind the activation-gradient chain, and must never be waited for)
struct LayerBwd {
  bf16_t *dYb, *dHpre, *dYb2, *dQKV;
  // per-block partial sums, finished by one k_reduce_multi launch on the weight-gradient stream:
  // LayerNorm 2 / 1 backward [blocks][3][H] (dense bias, gamma, beta), bias column sums of FFN1 [chunks][I] and QKV [chunks][3H]
  float *part_ln2, *part_ln1, *part_b1, *part_bqkv;
};

struct TrainBufs {
  int32_t* status;
  int32_t *tok_id, *tok_pos;
  int32_t* order;   // [B] sequences in descending length: the dispatch order of the attention kernels (k_len_order)
  LayerSave* L;  // host array (inside the plan object)
  LayerBwd* G;   // host array
  bf16_t *Xout, *cls_b;
  float *cls_y, *cls_f, *head_y;
  // backward scratch
  float *G0, *G1, *Drow, *slab, *part, *dcls_y, *dcls_f, *dhead_y;
  bf16_t *dctx, *dhead_yb, *dclsb;
  bf16_t *dXb1, *dXb2;   // dgrad outputs of FFN1 / the QKV projection (added to the fp32 stream by the next LayerNorm backward)
  // Last layer, CLS pooling: only the B CLS rows are live after the attention (models.py:43), so the output projection,
  // LayerNorm and FFN of that layer -- forward and backward -- run on compact [B, .] copies of those rows (c_*).
  bf16_t *c_ctx, *c_xin, *c_X1, *c_Hpre, *c_Hm;          // forward, kept for the backward
  float* c_Y1;
  float* c_ctx32;   // the CLS rows of the last layer's attention output in fp32 (AttnBwdArgs::cls32)
  bf16_t *c_dYb, *c_dHpre, *c_dYb2, *c_dctx;             // backward
  float *c_dX1f, *c_dY1, *c_slab;
  size_t c_slab_elems;
  int64_t ldt, Tp;
  size_t slab_elems;
  size_t total;
};

constexpr int TRAIN_MAX_LAYERS = 48;
constexpr int CLS_MAX_SPLIT = 16;   // contraction slices of the compact (CLS-row) projections of the last layer
#ifndef CONVDR_LN_BWD_BLOCKS
#define CONVDR_LN_BWD_BLOCKS 768
#endif
constexpr int LN_BWD_BLOCKS = CONVDR_LN_BWD_BLOCKS;
constexpr int EMB_BWD_BLOCKS = 512;   // (k_embed_bwd keeps a 48 KB staging image per workgroup: three per CU)
static_assert(EMB_BWD_BLOCKS <= LN_BWD_BLOCKS, "p.part is sized by LN_BWD_BLOCKS");
constexpr int COLSUM_CHUNKS = 64;
constexpr size_t SLAB_ELEMS = (size_t)16 * 3072 * 768;  // >= splits * N * K for every weight of a base-size model

struct TrainPlan {
  TrainBufs b;
  LayerSave layers[TRAIN_MAX_LAYERS];
  LayerBwd bwd[TRAIN_MAX_LAYERS];
};

static void train_plan(const convdr_encoder_config* c, int64_t rows, int B, char* base, TrainPlan& P) {
  TrainBufs& p = P.b;
  p.L = P.layers;
  p.G = P.bwd;
  size_t o = 0;
  // (integer arithmetic: the size-only call plans from a null base, and pointer arithmetic on null is undefined -- UBSan, make SAN=1)
  auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return (char*)((uintptr_t)base + at); };
  const int H = c->hidden, I = c->intermediate;
  const int64_t rs = rows + 128;
  p.Tp = (int64_t)align_up((size_t)rows, 64);
  p.ldt = p.Tp + 64;
  p.status = (int32_t*)take(256);   // workspace offset 0, as in the inference plan
  p.tok_id = (int32_t*)take(rs * 4);
  p.tok_pos = (int32_t*)take(rs * 4);
  p.order = (int32_t*)take((size_t)B * 4);
  for (int l = 0; l < c->layers; ++l) {
    LayerSave& s = P.layers[l];
    s.Xin = (bf16_t*)take(rs * H * 2);
    s.QKV = (bf16_t*)take(rs * 3 * H * 2);
    s.ctx = (bf16_t*)take(rs * H * 2);
    s.LSE = (float*)take((size_t)c->heads * p.ldt * 4);
    s.Mbits = (uint32_t*)take((size_t)c->heads * ATTM_PIECES * p.ldt * 4);
    if (l + 1 == c->layers && !c->pool_mean) {   // CLS-only tail: these live in the compact c_* buffers
      s.X1 = s.Hpre = s.Hm = nullptr;
      s.Y1 = s.Y2 = nullptr;
      continue;
    }
    s.X1 = (bf16_t*)take(rs * H * 2);
    s.Hpre = (bf16_t*)take(rs * I * 2);
    s.Hm = (bf16_t*)take(rs * I * 2);
    s.Y1 = (float*)take(rs * H * 4);
    s.Y2 = (float*)take(rs * H * 4);
  }
  const int64_t Bp = B + 128;
  const int E = c->out_dim > 0 ? c->out_dim : 4;
  p.Xout = (bf16_t*)take(rs * H * 2);
  p.cls_b = (bf16_t*)take(Bp * H * 2);
  p.cls_y = (float*)take(Bp * H * 4);
  p.cls_f = (float*)take(Bp * H * 4);
  p.head_y = (float*)take(Bp * E * 4);
  p.G0 = (float*)take(rs * H * 4);
  p.G1 = (float*)take(rs * H * 4);
  p.Drow = (float*)take((size_t)c->heads * p.ldt * 4);
  p.slab_elems = SLAB_ELEMS;
  p.slab = (float*)take(p.slab_elems * 4);
  p.part = (float*)take((size_t)LN_BWD_BLOCKS * 3 * 1024 * 4);   // head / CLS LayerNorm, embeddings
  p.dcls_y = (float*)take(Bp * H * 4);
  p.dcls_f = (float*)take(Bp * H * 4);
  p.dhead_y = (float*)take(Bp * E * 4);
  for (int l = 0; l < c->layers; ++l) {
    LayerBwd& g = P.bwd[l];
    g.dYb = (bf16_t*)take(rs * H * 2);
    g.dHpre = (bf16_t*)take(rs * I * 2);
    g.dYb2 = (bf16_t*)take(rs * H * 2);
    g.dQKV = (bf16_t*)take(rs * 3 * H * 2);
    g.part_ln2 = (float*)take((size_t)LN_BWD_BLOCKS * 3 * H * 4);
    g.part_ln1 = (float*)take((size_t)LN_BWD_BLOCKS * 3 * H * 4);
    g.part_b1 = (float*)take((size_t)COLSUM_CHUNKS * I * 4);
    g.part_bqkv = (float*)take((size_t)COLSUM_CHUNKS * 3 * H * 4);
  }
  p.dXb1 = (bf16_t*)take(rs * H * 2);
  p.dXb2 = (bf16_t*)take(rs * H * 2);
  p.c_ctx = (bf16_t*)take(Bp * H * 2);
  p.c_xin = (bf16_t*)take(Bp * H * 2);
  p.c_X1 = (bf16_t*)take(Bp * H * 2);
  p.c_Hpre = (bf16_t*)take(Bp * I * 2);
  p.c_Hm = (bf16_t*)take(Bp * I * 2);
  p.c_Y1 = (float*)take(Bp * H * 4);
  p.c_ctx32 = (float*)take(Bp * H * 4);
  p.c_dYb = (bf16_t*)take(Bp * H * 2);
  p.c_dHpre = (bf16_t*)take(Bp * I * 2);
  p.c_dYb2 = (bf16_t*)take(Bp * H * 2);
  p.c_dctx = (bf16_t*)take(Bp * H * 2);
  p.c_dX1f = (float*)take(Bp * H * 4);
  p.c_dY1 = (float*)take(Bp * H * 4);
  p.c_slab_elems = (size_t)CLS_MAX_SPLIT * Bp * H;
  p.c_slab = (float*)take(p.c_slab_elems * 4);
  p.dctx = (bf16_t*)take(rs * H * 2);
  p.dhead_yb = (bf16_t*)take(Bp * E * 2);
  p.dclsb = (bf16_t*)take(Bp * H * 2);
  p.total = o;
}

// The weight-gradient branch of a layer -- the bias column sums of the QKV / FFN1 projections and ONE batched TN GEMM
// launch for the four weight matrices -- has no consumer before the optimizer.  It runs on a private stream:
// fork() makes the side stream wait for what the main stream has enqueued so far (the layer's activation gradients);
// nothing on the main stream ever waits for the side stream before the final join, because every operand the branch
// reads lives in a per-layer buffer (LayerBwd / LayerSave) and its scratch (slab, the bias part of `part`) is its own.
static hipStream_t g_ext_side[16] = {};   // per device, like WgradFork::get(): convdr_train_set_side_stream
static int cur_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  return dev;
}

// Timing-only experiments (`make TRACE=1` library only: the results are garbage): CONVDR_DBG_SKIP drops whole classes of
// launches -- 1 gelu', 2 LayerNorm backward, 4 forward attention, 8 weight gradients, 16 attention backward, 32 bias column sums (zero partials instead), 64 forward
// LayerNorm, 128 the per-layer gradient-norm partials (zeros instead), 256 the transposed-weight refresh (512: that launch twice) -- to measure what each class costs the STEP (tools/dbg/skip_probe.sh, profiles/r04_train_kd_sensitivity.txt).
// CONVDR_DBG_DOUBLE (same library; tools/dbg/double_probe.sh): a class of IDEMPOTENT launches goes out twice, results unchanged --
// the step's difference is the class's marginal cost with the product's own operands (a skipped class leaves stale operands behind,
// and the GEMMs' time depends on their operands: profiles/r06_ab_transpose.txt).  1 forward GEMMs, 2 data-gradient GEMMs (both in
// gemm_launch.hpp), 4 weight gradients, 8 forward attention, 16 attention backward, 32 forward LayerNorm, 64 LayerNorm backward,
// 128 bias column sums + their reductions (storing backward only), 256 gradient-norm partials.
#ifdef CONVDR_ENABLE_TRACE
static int dbg_skip() { static const int v = getenv("CONVDR_DBG_SKIP") ? atoi(getenv("CONVDR_DBG_SKIP")) : 0; return v; }
static int dbg_double() { static const int v = getenv("CONVDR_DBG_DOUBLE") ? atoi(getenv("CONVDR_DBG_DOUBLE")) : 0; return v; }
#else
static constexpr int dbg_skip() { return 0; }
static constexpr int dbg_double() { return 0; }
#endif   // convdr_train_set_side_stream (before the first backward of the process)

// convdr_encoder_backward_fresh: every parameter gradient except the embedding tables is STORED by the one kernel that completes it
// (weight-gradient tiles, k_reduce_multi jobs) instead of added to the buffer's contents; set for the duration of that call only
static thread_local bool t_bwd_overwrite = false;

struct WgradFork {
  hipStream_t main, side;
  hipEvent_t prod, fin;
  hipEvent_t head_fin;   // the head's parameter-gradient jobs on the side stream have read p.part
  hipEvent_t emb_main;   // the embedding gradients are written (recorded on the main stream, before the join)
  // "every gradient of layer l is written": one event on each stream (convdr_backward_wait_layer)
  hipEvent_t layer_main[TRAIN_MAX_LAYERS], layer_side[TRAIN_MAX_LAYERS];
  int layers_recorded;
  bool ok;
  static WgradFork& get() {   // one per device (the side stream belongs to the device that is current at creation)
    static WgradFork f[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    return f[dev];
  }
  int init(hipStream_t st) {
    main = st;
    if (!ok) {
      // (a lowest-priority side stream -- so that this branch would only take the compute units the critical chain leaves
      //  idle -- measured no different on MI355X, round 3: 11.63 vs 11.66 ms per configs[2] step; round 4: streams of
      //  other priorities bring their own hardware queues, and with more than GPU_MAX_HW_QUEUES = 4 queues alive the step
      //  takes 17-19 ms.)  Which hardware queue a stream shares with which other stream is decided by HIP in order of first
      // use, and a branch that shares the MAIN stream's queue runs serialised with the chain it is meant to run beside
      // (10.7 -> 12.4 ms per step depending on nothing but how many streams the process had used before,
      // tools/dbg/stream_queue_probe.py): the host may hand in a stream it has verified to run concurrently with the main
      // one (convdr_train_set_side_stream; train.py:_aux_streams does that once per process).
      if (g_ext_side[cur_device_slot()]) side = g_ext_side[cur_device_slot()];
      else CONVDR_CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      // These events order streams of ONE device against each other; nothing on the host or on another device ever inspects them
      // (the collectives of parallel.py wait for them on a stream of the same device).  hipEventDisableSystemFence: the record
      // then does not carry a system-scope release (cache write-back + invalidate) of its own -- the kernels' own agent-scope
      // fences order the data.  CONVDR_EVENT_SYSTEM_FENCE=1: the default flags of rounds 1-5 (A/B).
      static const bool sys_fence = getenv("CONVDR_EVENT_SYSTEM_FENCE") && atoi(getenv("CONVDR_EVENT_SYSTEM_FENCE"));
      const unsigned ef = hipEventDisableTiming | (sys_fence ? 0u : hipEventDisableSystemFence);
      CONVDR_CHECK_HIP(hipEventCreateWithFlags(&prod, ef));
      CONVDR_CHECK_HIP(hipEventCreateWithFlags(&fin, ef));
      CONVDR_CHECK_HIP(hipEventCreateWithFlags(&head_fin, ef));
      CONVDR_CHECK_HIP(hipEventCreateWithFlags(&emb_main, ef));
      for (int i = 0; i < TRAIN_MAX_LAYERS; ++i) {
        CONVDR_CHECK_HIP(hipEventCreateWithFlags(&layer_main[i], ef));
        CONVDR_CHECK_HIP(hipEventCreateWithFlags(&layer_side[i], ef));
      }
      ok = true;
    }
    return 0;
  }
  int fork() {
    CONVDR_CHECK_HIP(hipEventRecord(prod, main));
    CONVDR_CHECK_HIP(hipStreamWaitEvent(side, prod, 0));
    return 0;
  }
  int join() {   // the main stream waits for everything enqueued on the side stream
    CONVDR_CHECK_HIP(hipEventRecord(fin, side));
    CONVDR_CHECK_HIP(hipStreamWaitEvent(main, fin, 0));
    return 0;
  }
  int layer_done(int l, hipStream_t side_or_main) {   // both chains have enqueued the last gradient writes of layer l
    CONVDR_CHECK_HIP(hipEventRecord(layer_main[l], main));
    CONVDR_CHECK_HIP(hipEventRecord(layer_side[l], side_or_main));
    return 0;
  }
};

static int check_dropout(const convdr_dropout* d, const convdr_encoder_config* c, int64_t rows) {
  if (!d) return 0;
  CONVDR_REQUIRE(d->p_hidden >= 0.f && d->p_hidden < 1.f && d->p_attention >= 0.f && d->p_attention < 1.f,
                 "dropout probabilities must be in [0, 1) (hidden %g, attention %g)", d->p_hidden, d->p_attention);
  // 32-bit element counters (csrc/dropout.hpp): rows * H / 2 and (rows * heads) << 9 must not wrap
  CONVDR_REQUIRE((double)rows * c->hidden / 2 < 4.0e9 && (double)rows * c->heads * 512 < 4.0e9,
                 "dropout: %lld packed rows overflow the 32-bit mask counters", (long long)rows);
  return 0;
}

static int check_train_config(const convdr_encoder_config* c) {
  CONVDR_REQUIRE(c->hidden % 128 == 0 && c->hidden <= 1024 && c->heads * 64 == c->hidden,
                 "train: hidden must be a multiple of 128 (<= 1024) with head_dim 64 (hidden=%d heads=%d)", c->hidden,
                 c->heads);
  CONVDR_REQUIRE(c->intermediate % 64 == 0 && c->intermediate <= 3072 * 2 && c->layers <= TRAIN_MAX_LAYERS,
                 "train: unsupported intermediate/layers (%d, %d)", c->intermediate, c->layers);
  CONVDR_REQUIRE(c->out_dim == 0 || (c->out_dim % 64 == 0 && c->out_dim <= 1024), "train: out_dim %% 64 != 0 (%d)",
                 c->out_dim);
  return 0;
}

// dW[n, k] += sum_t dY[t, n] X[t, k] for up to TN_MAX_PROBLEMS weight matrices in one launch, straight from the
// token-major operands (TN engine, gemm_tn.hpp).  Every output tile runs the whole contraction and adds its result into
// the gradient itself; only when the contraction is long and the tiles too few to fill the chip is it cut into slices,
// each writing an fp32 slab that k_reduce_partials then adds into the gradient in slice order.  Deterministic either way.
struct WgradItem {
  const bf16_t* dY; int N; int64_t ld_dy;
  const bf16_t* X; int K; int64_t ld_x;
  float* dW;
};
template <class T>
static int wgrad_launch(const WgradItem* it, int count, int64_t rows, float* slab, size_t slab_elems, hipStream_t st,
                        int tail_split) {
  static DeviceOnce attr_done;
  if (attr_done.first())
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_tn<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         TnCfg<T>::SMEM_BYTES));
  GemmTnArgs g{};
  g.count = count;
  g.rows = rows;
  int tiles = 0;
  size_t elems = 0;
  for (int i = 0; i < count; ++i) {
    GemmTnProblem& q = g.p[i];
    q.Rm = it[i].X; q.ldr = it[i].ld_x; q.NR = it[i].K;
    q.Lm = it[i].dY; q.ldl = it[i].ld_dy; q.NL = it[i].N;
    q.tilesR = (q.NR + T::TR - 1) / T::TR;
    tiles += q.tilesR * ((q.NL + T::TL - 1) / T::TL);
    q.tile_end = tiles;
    q.out = it[i].dW;
    elems += (size_t)q.NL * q.NR;
  }
  const int steps = (int)ceil_div64(rows, 64);
  int nsplit = 1;
  bool ordered = false;
  // Ordered in-place slices are OFF by default: at the configs[2] size they cut the launch from 3.5 to 2.2 ms of kernel
  // time per step (216 instead of 108 workgroups) and make the STEP 2 % slower (12.09 vs 11.85 ms, A/B on one box, round
  // 3) -- the branch runs beside the activation-gradient chain, which is the critical path and loses the compute units
  // the wider launch takes.  CONVDR_WGRAD_SPLIT=1 turns them on (convdr_wgrad callers with nothing running beside).
  static const bool no_ordered = !(getenv("CONVDR_WGRAD_SPLIT") && atoi(getenv("CONVDR_WGRAD_SPLIT")));
  if (tail_split > 1 && slab && steps < 512) {
    // the last branch of a backward pass has nothing left to hide behind: one tile's whole contraction (~250 us at
    // configs[2]) is the tail of the step, so it is cut into slab slices that fill the idle chip
    nsplit = tail_split;
    if (nsplit > steps / 32) nsplit = steps / 32;
    if ((size_t)nsplit * elems > slab_elems) nsplit = (int)(slab_elems / elems);
    if (nsplit < 1) nsplit = 1;
  } else if (steps >= 512 && slab) {   // long contraction, few tiles: slices of >= 256 K steps until the chip is full
    nsplit = (int)ceil_div64(device_cu_count(), tiles);
    if (nsplit > steps / 256) nsplit = steps / 256;
    if ((size_t)nsplit * elems > slab_elems) nsplit = (int)(slab_elems / elems);
    if (nsplit < 1) nsplit = 1;
  } else if (slab && !no_ordered && steps >= 64 && 2 * tiles <= device_cu_count() * (TnCfg<T>::SMEM_BYTES > 80 * 1024 ? 1 : 2) &&
             (size_t)tiles * sizeof(int) <= slab_elems * sizeof(float)) {
    // short contraction, the tiles fill at most half of the chip (configs[2]: 141 K steps, 108 tiles on 256 CUs):
    // ordered in-place slices of >= 32 K steps, all workgroups resident at once (see GemmTnArgs::flags)
    const int slots = device_cu_count() * (TnCfg<T>::SMEM_BYTES > 80 * 1024 ? 1 : 2);
    nsplit = slots / tiles;
    if (nsplit > steps / 32) nsplit = steps / 32;
    if (nsplit > 4) nsplit = 4;
    ordered = nsplit > 1;
    if (!ordered) nsplit = 1;
  }
  // the operand windows are addressed with 32-bit byte offsets (buffer descriptors): a contraction slice must stay below
  // 2 GiB -- very long row counts are cut into more slab slices by themselves (when there is a slab to hold them)
  int64_t max_ld = 0;
  for (int i = 0; i < count; ++i) max_ld = std::max(max_ld, std::max(it[i].ld_x, it[i].ld_dy));
  const int64_t window_steps = (((int64_t)1 << 31) - 1) / (64 * max_ld * 2);
  if (window_steps >= 1 && steps > window_steps * nsplit && slab) {
    const int need = (int)ceil_div64(steps, window_steps);
    if ((size_t)need * elems <= slab_elems) { nsplit = need; ordered = false; }
  }
  g.steps_per_split = (steps + nsplit - 1) / nsplit;
  if (g.steps_per_split < 1) g.steps_per_split = 1;
  nsplit = steps > 0 ? (steps + g.steps_per_split - 1) / g.steps_per_split : 1;
  g.nsplit = nsplit;
  CONVDR_REQUIRE((int64_t)g.steps_per_split * 64 * max_ld * 2 < ((int64_t)1 << 31),
                 "wgrad: a contraction slice of %d x 64 rows x %lld columns exceeds the 2 GiB operand window (pass a slab so "
                 "that it can be split)", g.steps_per_split, (long long)max_ld);
  g.flags = nullptr;
  g.overwrite = t_bwd_overwrite ? 1 : 0;
  if (ordered && nsplit > 1) {
    g.flags = (int*)slab;
    CONVDR_CHECK_HIP(hipMemsetAsync(g.flags, 0, (size_t)tiles * sizeof(int), st));
  } else if (nsplit > 1) {
    size_t o = 0;
    for (int i = 0; i < count; ++i) {
      g.p[i].out = slab + o;
      o += (size_t)nsplit * g.p[i].NL * g.p[i].NR;
    }
  }
  {
    ProfScope prof("gemm_wgrad", st);
    if (!(dbg_skip() & 8))
      hipLaunchKernelGGL((k_gemm_tn<T>), dim3((unsigned)tiles, (unsigned)nsplit), dim3(T::THREADS), TnCfg<T>::SMEM_BYTES, st, g);
    if ((dbg_double() & 4) && t_bwd_overwrite && !g.flags)   // (stores or slabs: idempotent)
      hipLaunchKernelGGL((k_gemm_tn<T>), dim3((unsigned)tiles, (unsigned)nsplit), dim3(T::THREADS), TnCfg<T>::SMEM_BYTES, st, g);
    CONVDR_CHECK_LAUNCH("k_gemm_tn");
  }
  if (nsplit > 1 && !g.flags)
    for (int i = 0; i < count; ++i) {
      const int64_t n = (int64_t)g.p[i].NL * g.p[i].NR;
      hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)(ceil_div64(n / 4, 256) < 2048 ? ceil_div64(n / 4, 256) : 2048)),
                         dim3(256), 0, st, g.p[i].out, nsplit, n, n, it[i].dW, t_bwd_overwrite ? 0 : 1);
      CONVDR_CHECK_LAUNCH("k_reduce_partials");
    }
  return 0;
}
static int wgrad_batch(const WgradItem* it, int count, int64_t rows, float* slab, size_t slab_elems, hipStream_t st,
                       int tail_split = 0) {
  CONVDR_REQUIRE(count >= 1 && count <= TN_MAX_PROBLEMS, "wgrad: %d problems in one batch", count);
  bool big = true;
  for (int i = 0; i < count; ++i) {
    CONVDR_REQUIRE(it[i].N % 8 == 0 && it[i].K % 8 == 0 && it[i].ld_dy % 8 == 0 && it[i].ld_x % 8 == 0,
                   "wgrad: N, K and the row strides must be multiples of 8 (N=%d K=%d)", it[i].N, it[i].K);
    big = big && it[i].N >= 256 && it[i].K >= 256;
  }
  // 256 x 256 tiles (half the L2 -> LDS bytes per FLOP of 128 x 128: at two 128-tiles per CU the operand stream, not the
  // matrix pipe, bounded the round-1 kernel) unless a matrix is smaller than a tile
  // (round 5: 128 x 128 tiles here too -- 64 KB of LDS, so that a weight-gradient workgroup SHARES a compute unit with a
  //  128 x 128 workgroup of the activation-gradient chain instead of owning it -- measured worse: step 10.3-10.6 vs 9.78 ms,
  //  the data-gradient GEMMs 3.0 -> 3.6-4.1 ms per step; profiles/r05_ab_wgrad_tile128.txt)
  if (big) return wgrad_launch<Tile256>(it, count, rows, slab, slab_elems, st, tail_split);
  return wgrad_launch<Tile128>(it, count, rows, slab, slab_elems, st, tail_split);
}
static int wgrad(const bf16_t* dY, int N, int64_t ld_dy, const bf16_t* X, int K, int64_t ld_x, int64_t rows, const TrainBufs& p,
                 float* dW, hipStream_t st) {
  const WgradItem it{dY, N, ld_dy, X, K, ld_x, dW};
  return wgrad_batch(&it, 1, rows, p.slab, p.slab_elems, st);
}

static int colsum_chunks(int64_t rows) { return rows >= 2048 ? COLSUM_CHUNKS : 8; }

// LayerNorm backward of `rows` rows, incoming gradient dY (fp32) + dYadd (bf16, optional).  Partial sums of
// (dense bias = column sums of dX, dgamma, dbeta) go to part[blocks][3][H]; returns the number of blocks.
static int ln_bwd_kernel(const float* dY, const bf16_t* dYadd, const float* Yin, int64_t rows, int H, const float* g, float eps,
                         float* dXf, bf16_t* dXb, float* part, int* blocks_out, hipStream_t st,
                         const DropSite drop = DropSite{0u, 0u, 1.f}, const int32_t* row_map = nullptr) {
  int blocks = (int)(ceil_div64(rows, 4) < LN_BWD_BLOCKS ? ceil_div64(rows, 4) : LN_BWD_BLOCKS);
  ProfScope prof("layernorm_bwd", st);
  if (!(dbg_skip() & 2))
  for (int rep = 0; rep < ((dbg_double() & 64) ? 2 : 1); ++rep)
  {
    // g_ln_bwd_rows (option "ln_bwd_rows" / CONVDR_LN_BWD_ROWS): 0 = the general kernel everywhere (A/B), 1 = straight-line form without, 2 (default) = with the
    // register prefetch of the next row; CONVDR_LN_BWD_GRID: workgroups of the straight-line form.  configs[2] step, medians of
    // three alternations on one box: 9.16 ms (0) -> 8.96 (2), 8.93-8.96 for (1) and for grids 384 / 444 / 512 -- the 24
    // launches of a step 0.94 -> 0.74 ms (profiles/r05_ab_ln_bwd_rows.txt).
    const int mode = (int)g_ln_bwd_rows;
    static const int grid_env = getenv("CONVDR_LN_BWD_GRID") ? atoi(getenv("CONVDR_LN_BWD_GRID")) : 0;
    if (mode && H == 768 && dY && dYadd && dXf && dXb && !row_map) {
      if (grid_env > 0 && grid_env < blocks) blocks = grid_env;
      if (mode == 2)
        hipLaunchKernelGGL((k_layernorm_bwd_rows<3, true, 3>), dim3(blocks), dim3(256), 0, st, dY, dYadd, Yin, rows, g, eps, dXf, dXb, part, drop);
      else
        hipLaunchKernelGGL((k_layernorm_bwd_rows<3, false, 4>), dim3(blocks), dim3(256), 0, st, dY, dYadd, Yin, rows, g, eps, dXf, dXb, part, drop);
    } else
      hipLaunchKernelGGL(k_layernorm_bwd, dim3(blocks), dim3(256), 0, st, dY, dYadd, Yin, rows, H, g, eps, dXf, dXb, part, drop, row_map);
  }
  CONVDR_CHECK_LAUNCH("k_layernorm_bwd");
  *blocks_out = blocks;
  return 0;
}

// Projection of the B compact CLS rows of the last layer, A[B, K] W[N, K]^T -> c_slab[slices][B][N]: the contraction is cut
// into slices of 3 K steps so that the launch has N / 128 x slices tiles -- as one whole-contraction tile per 128 features it
// would be a 12- (K = 768) or 48-step (K = 3072) latency chain on 6 compute units, as long as the full-size GEMM it replaces.
static int cls_projection(const bf16_t* W, const bf16_t* A, int B, int N, int K, const TrainBufs& p, hipStream_t st, int* nsplit) {
  const int nk = K / GEMM_BK;
  int ns = nk / 3 < CLS_MAX_SPLIT ? nk / 3 : CLS_MAX_SPLIT;
  while (ns > 1 && nk % ns) --ns;
  if (ns < 1) ns = 1;
  CONVDR_REQUIRE((size_t)ns * B * N <= p.c_slab_elems, "train: CLS-row slab too small (%d x %d x %d)", ns, B, N);
  GemmArgs g{};
  g.rows = B; g.W = W; g.X = A; g.N = N; g.K = K; g.k_split_len = K / ns; g.Cf = p.c_slab;
  *nsplit = ns;
  return launch_gemm<EPI_SLAB_F32>(g, st, "gemm_cls");
}

struct ReduceList {
  ReduceJobs a{};
  int blocks = 0;
  // out[0 .. n) += sum over nparts of part[p * stride + .]
  void add(const float* part, int nparts, int64_t stride, int n, float* out) {
    ReduceJob& q = a.j[a.count++];
    q.part = part; q.out = out; q.stride = stride; q.nparts = nparts; q.n = n;
    blocks += (n + 15) / 16;
    q.block_end = blocks;
  }
  // the three parameter gradients of a LayerNorm backward: one job when they are adjacent in the gradient arena
  // (dbias, dgamma, dbeta -- train.py:_tower_params order), else one job each
  void add_ln(const float* part, int blocks_ln, int H, float* dbias, float* dgamma, float* dbeta) {
    if (dbias && dgamma == dbias + H && dbeta == dgamma + H) {
      add(part, blocks_ln, (int64_t)3 * H, 3 * H, dbias);
    } else {
      float* outs[3] = {dbias, dgamma, dbeta};
      for (int k = 0; k < 3; ++k)
        if (outs[k]) add(part + (size_t)k * H, blocks_ln, (int64_t)3 * H, H, outs[k]);
    }
  }
  int launch(hipStream_t st) {
    if (!a.count) return 0;
    a.overwrite = t_bwd_overwrite ? 1 : 0;
    hipLaunchKernelGGL(k_reduce_multi, dim3(blocks), dim3(256), 0, st, a);
    CONVDR_CHECK_LAUNCH("k_reduce_multi");
    return 0;
  }
};

}  // namespace convdr

using namespace convdr;

extern "C" size_t convdr_encoder_train_workspace_bytes(const convdr_encoder_config* cfg, int64_t rows, int B) {
  if (cfg->layers > TRAIN_MAX_LAYERS) return 0;
  TrainPlan P;
  train_plan(cfg, rows, B, nullptr, P);
  return P.b.total;
}

extern "C" int convdr_encoder_train_forward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                                            const void* input_ids, int ids_are_int32, const int64_t* attention_mask, int B,
                                            int L, const int32_t* cu_seqlens, const int32_t* seq_lens, int64_t rows,
                                            int max_len, void* workspace, size_t workspace_bytes, float* out,
                                            const convdr_dropout* dropout, convdr_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int e = check_train_config(cfg)) return e;
  if (int e = check_dropout(dropout, cfg, rows)) return e;
  const float p_hid = dropout ? dropout->p_hidden : 0.f, p_att = dropout ? dropout->p_attention : 0.f;
  const uint32_t dseed = dropout ? dropout->seed : 0u;
  CONVDR_REQUIRE(B > 0 && L > 0 && rows > 0 && rows % 8 == 0 && max_len > 0 && max_len <= L,
                 "convdr_encoder_train_forward: bad sizes B=%d L=%d rows=%lld max_len=%d", B, L, (long long)rows, max_len);
  TrainPlan P;
  train_plan(cfg, rows, B, (char*)workspace, P);
  const TrainBufs& p = P.b;
  CONVDR_REQUIRE(workspace_bytes >= p.total, "convdr_encoder_train_forward: workspace too small (%zu < %zu)",
                 workspace_bytes, p.total);
  const int H = cfg->hidden, I = cfg->intermediate;
  CONVDR_CHECK_HIP(hipMemsetAsync(p.status, 0, 256, st));
  hipLaunchKernelGGL(k_seq_pack, dim3((B + 3) / 4), dim3(256), 0, st, input_ids, ids_are_int32, attention_mask, seq_lens, B, L,
                     cu_seqlens, cfg->kind, cfg->pad_idx, cfg->max_pos, cfg->vocab, p.tok_id, p.tok_pos, p.status);
  CONVDR_CHECK_LAUNCH("k_seq_pack");
  hipLaunchKernelGGL(k_len_order, dim3((B + 255) / 256), dim3(256), 0, st, seq_lens, B, p.order);
  CONVDR_CHECK_LAUNCH("k_len_order");
  hipLaunchKernelGGL(k_embed_ln, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, p.tok_id, p.tok_pos, rows, H,
                     w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, cfg->ln_eps, P.layers[0].Xin,
                     drop_site(dseed, DROP_SITE_EMB, 0, p_hid));
  CONVDR_CHECK_LAUNCH("k_embed_ln");
  for (int l = 0; l < cfg->layers; ++l) {
    const convdr_layer_weights* lw = &w->layers[l];
    const LayerSave& s = P.layers[l];
    GemmArgs g{};
    g.rows = rows; g.W = (const bf16_t*)lw->wqkv; g.X = s.Xin; g.N = 3 * H; g.K = H; g.bias = lw->bqkv; g.Cb = s.QKV;
    if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_qkv")) return e;
    // Last layer with CLS pooling: only the B CLS rows are live from here on (models.py:43).  The attention runs its first
    // query tile only (it holds row 0 of every sequence), and output projection, LayerNorm and FFN run on compact [B, .]
    // copies of those rows (TrainBufs::c_*; dropout masks indexed by the packed row, so the result is the full layer's).
    const bool cls_tail = l + 1 == cfg->layers && !cfg->pool_mean;
    {
      // (keep bits for the one-workgroup backward: whenever that kernel CAN take this batch -- the "attn_bwd_fused" option is
      //  looked at by the backward only, so the bits are there whichever way it is set then)
      AttnTrainArgs a{s.QKV, rows, cu_seqlens, seq_lens, H, s.ctx, drop_site(dseed, DROP_SITE_ATT_PROBS, l, p_att), s.LSE, p.ldt, 0.125f,
                      p.order, max_len <= ATTF_MAX_LEN ? s.Mbits : nullptr, cls_tail ? p.c_ctx32 : nullptr};
      ProfScope prof("attention", st);
      const dim3 grid(cls_tail ? 1 : (max_len + 127) / 128, cfg->heads, B);
      if (!(dbg_skip() & 4))
        for (int rep = 0; rep < ((dbg_double() & 8) ? 2 : 1); ++rep) {
          if (a.drop.thresh) hipLaunchKernelGGL(k_attention_train_fwd<true>, grid, dim3(256), 4 * ATT_TILE, st, a);
          else hipLaunchKernelGGL(k_attention_train_fwd<false>, grid, dim3(256), 4 * ATT_TILE, st, a);
        }
      CONVDR_CHECK_LAUNCH("k_attention_train_fwd");
    }
    if (cls_tail) {
      hipLaunchKernelGGL(k_gather_cls2, dim3((B + 3) / 4), dim3(256), 0, st, cu_seqlens, B, H, s.ctx, s.Xin, p.c_ctx, p.c_xin);
      CONVDR_CHECK_LAUNCH("k_gather_cls2");
      const unsigned fin_blocks = (unsigned)ceil_div64((int64_t)B * H / 4, 256);
      int ns = 1;
      if (int e = cls_projection((const bf16_t*)lw->wo, p.c_ctx, B, H, H, p, st, &ns)) return e;
      hipLaunchKernelGGL(k_slab_finish, dim3(fin_blocks), dim3(256), 0, st, p.c_slab, ns, B, H, lw->bo, p.c_xin, cu_seqlens,
                         drop_site(dseed, DROP_SITE_ATTN_OUT, l, p_hid), p.c_Y1);
      hipLaunchKernelGGL(k_layernorm, dim3((B + 3) / 4), dim3(256), 0, st, p.c_Y1, (int64_t)B, H, lw->ln1_g, lw->ln1_b, cfg->ln_eps, p.c_X1,
                         (float*)nullptr);
      g = GemmArgs{};
      g.rows = B; g.W = (const bf16_t*)lw->w1; g.X = p.c_X1; g.N = I; g.K = H; g.bias = lw->b1; g.Cb = p.c_Hm; g.Cb2 = p.c_Hpre;
      if (int e = launch_gemm<EPI_GELU_SAVE>(g, st, "gemm_cls")) return e;
      if (int e = cls_projection((const bf16_t*)lw->w2, p.c_Hm, B, H, I, p, st, &ns)) return e;
      hipLaunchKernelGGL(k_slab_finish, dim3(fin_blocks), dim3(256), 0, st, p.c_slab, ns, B, H, lw->b2, p.c_X1, cu_seqlens,
                         drop_site(dseed, DROP_SITE_FFN_OUT, l, p_hid), p.cls_y);   // = the CLS rows of the pre-LayerNorm2 sums
      CONVDR_CHECK_LAUNCH("cls tail");
      float* cls_out = cfg->out_dim > 0 ? p.cls_f : out;
      hipLaunchKernelGGL(k_layernorm, dim3((B + 3) / 4), dim3(256), 0, st, p.cls_y, (int64_t)B, H, lw->ln2_g, lw->ln2_b,
                         cfg->ln_eps, p.cls_b, cls_out);
      CONVDR_CHECK_LAUNCH("k_layernorm(cls)");
      continue;
    }
    g = GemmArgs{};
    g.rows = rows; g.W = (const bf16_t*)lw->wo; g.X = s.ctx; g.N = H; g.K = H; g.bias = lw->bo; g.Cf = s.Y1; g.R = s.Xin;
    g.drop = drop_site(dseed, DROP_SITE_ATTN_OUT, l, p_hid);
    if (int e = launch_gemm<EPI_RESID_F32>(g, st, "gemm_attn_out")) return e;
    if (!(dbg_skip() & 64))
      for (int rep = 0; rep < ((dbg_double() & 32) ? 2 : 1); ++rep)
        launch_layernorm_bf16(s.Y1, rows, H, lw->ln1_g, lw->ln1_b, cfg->ln_eps, s.X1, st);
    CONVDR_CHECK_LAUNCH("k_layernorm");
    g = GemmArgs{};
    g.rows = rows; g.W = (const bf16_t*)lw->w1; g.X = s.X1; g.N = I; g.K = H; g.bias = lw->b1; g.Cb = s.Hm; g.Cb2 = s.Hpre;
    // (gelu_gp: s.Hpre holds gelu'(pre-activation) in the blocked layout instead of the pre-activation itself)
    if (g_gelu_gp) {
      if (int e = launch_gemm<EPI_GELU_GP>(g, st, "gemm_ffn1")) return e;
    } else if (int e = launch_gemm<EPI_GELU_SAVE>(g, st, "gemm_ffn1")) return e;
    g = GemmArgs{};
    g.rows = rows; g.W = (const bf16_t*)lw->w2; g.X = s.Hm; g.N = H; g.K = I; g.bias = lw->b2; g.Cf = s.Y2; g.R = s.X1;
    g.drop = drop_site(dseed, DROP_SITE_FFN_OUT, l, p_hid);
    if (int e = launch_gemm<EPI_RESID_F32>(g, st, "gemm_ffn2")) return e;
    if (l + 1 < cfg->layers) {
      if (!(dbg_skip() & 64))
        for (int rep = 0; rep < ((dbg_double() & 32) ? 2 : 1); ++rep)
          launch_layernorm_bf16(s.Y2, rows, H, lw->ln2_g, lw->ln2_b, cfg->ln_eps, P.layers[l + 1].Xin, st);
      CONVDR_CHECK_LAUNCH("k_layernorm");
    } else if (cfg->pool_mean) {   // use_mean = True: masked mean of the whole last layer's output
      launch_layernorm_bf16(s.Y2, rows, H, lw->ln2_g, lw->ln2_b, cfg->ln_eps, p.Xout, st);
      hipLaunchKernelGGL(k_masked_mean, dim3(B), dim3(256), 0, st, p.Xout, cu_seqlens, seq_lens, H, p.cls_b,
                         cfg->out_dim > 0 ? p.cls_f : out);
      CONVDR_CHECK_LAUNCH("k_masked_mean");
    }   // (CLS pooling: the last layer took the cls_tail branch above)
  }
  if (cfg->out_dim > 0) {
    GemmArgs g{};
    g.rows = B; g.W = (const bf16_t*)w->head_w; g.X = p.cls_b; g.N = cfg->out_dim; g.K = H; g.bias = w->head_b; g.Cf = p.head_y;
    if (int e = launch_gemm<EPI_F32>(g, st, "gemm_head")) return e;
    hipLaunchKernelGGL(k_layernorm, dim3((B + 3) / 4), dim3(256), 0, st, p.head_y, (int64_t)B, cfg->out_dim, w->head_ln_g,
                       w->head_ln_b, cfg->head_ln_eps, (bf16_t*)nullptr, out);
    CONVDR_CHECK_LAUNCH("k_layernorm(head)");
  }
  return 0;
}

static int encoder_backward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w, const convdr_layer_weights_t* wt,
                            const int32_t* cu_seqlens, const int32_t* seq_lens, const void* head_w_t, int B, int64_t rows, int max_len,
                            void* workspace, size_t workspace_bytes, const float* d_out, const convdr_encoder_grads* gr,
                            const convdr_dropout* dropout, convdr_stream_t stream);

extern "C" int convdr_encoder_backward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                                       const convdr_layer_weights_t* wt, const int32_t* cu_seqlens, const int32_t* seq_lens,
                                       const void* head_w_t, int B, int64_t rows, int max_len, void* workspace,
                                       size_t workspace_bytes, const float* d_out, const convdr_encoder_grads* gr,
                                       const convdr_dropout* dropout, convdr_stream_t stream) {
  t_bwd_overwrite = false;
  return encoder_backward(cfg, w, wt, cu_seqlens, seq_lens, head_w_t, B, rows, max_len, workspace, workspace_bytes, d_out, gr, dropout,
                          stream);
}

extern "C" int convdr_encoder_backward_fresh(const convdr_encoder_config* cfg, const convdr_encoder_weights* w,
                                             const convdr_layer_weights_t* wt, const int32_t* cu_seqlens, const int32_t* seq_lens,
                                             const void* head_w_t, int B, int64_t rows, int max_len, void* workspace,
                                             size_t workspace_bytes, const float* d_out, const convdr_encoder_grads* gr,
                                             const convdr_dropout* dropout, convdr_stream_t stream) {
  t_bwd_overwrite = true;
  const int e = encoder_backward(cfg, w, wt, cu_seqlens, seq_lens, head_w_t, B, rows, max_len, workspace, workspace_bytes, d_out, gr,
                                 dropout, stream);
  t_bwd_overwrite = false;
  return e;
}

static int encoder_backward(const convdr_encoder_config* cfg, const convdr_encoder_weights* w, const convdr_layer_weights_t* wt,
                            const int32_t* cu_seqlens, const int32_t* seq_lens, const void* head_w_t, int B, int64_t rows, int max_len,
                            void* workspace, size_t workspace_bytes, const float* d_out, const convdr_encoder_grads* gr,
                            const convdr_dropout* dropout, convdr_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int e = check_train_config(cfg)) return e;
  if (int e = check_dropout(dropout, cfg, rows)) return e;
  const float p_hid = dropout ? dropout->p_hidden : 0.f, p_att = dropout ? dropout->p_attention : 0.f;
  const uint32_t dseed = dropout ? dropout->seed : 0u;
  TrainPlan P;
  train_plan(cfg, rows, B, (char*)workspace, P);
  const TrainBufs& p = P.b;
  CONVDR_REQUIRE(workspace_bytes >= p.total, "convdr_encoder_backward: workspace too small (%zu < %zu)", workspace_bytes,
                 p.total);
  const int H = cfg->hidden, I = cfg->intermediate, NL = cfg->layers;
  const convdr_layer_grads* lg_last = &gr->layers[NL - 1];
  const float* dcls = d_out;  // gradient w.r.t. LayerNorm2(cls rows) of the last layer

  WgradFork& wf = WgradFork::get();
  static const bool fork_wgrad = !(getenv("CONVDR_NO_WGRAD_FORK") && atoi(getenv("CONVDR_NO_WGRAD_FORK")));
  if (int e = wf.init(st)) return e;
  hipStream_t ss = fork_wgrad ? wf.side : st;   // stream of the weight-gradient branches
  // The head and the CLS-row LayerNorm are a chain of ~6 launches of a few microseconds each in front of the first big kernel:
  // what only FINISHES parameter gradients (the two partial-sum reductions, the head's weight gradient) goes to the
  // weight-gradient stream, off that chain (round 5; CONVDR_HEAD_WGRAD_INLINE=1: on the main stream as in rounds 1-4).
  static const bool head_inline = getenv("CONVDR_HEAD_WGRAD_INLINE") && atoi(getenv("CONVDR_HEAD_WGRAD_INLINE"));
  // (the two LayerNorm backwards keep their partial sums apart: the second one starts while the first one's reduction may
  //  still be pending on the other stream)
  const int64_t part2_off = (int64_t)ceil_div64(B, 4) * 3 * 1024;
  const bool side_ok = fork_wgrad && !head_inline && 2 * ceil_div64(B, 4) <= LN_BWD_BLOCKS;
  hipStream_t hs = side_ok ? ss : st;
  float* part2 = side_ok ? p.part + part2_off : p.part;

  // ---- head: out = LayerNorm(head_y), head_y = cls_b . head_w^T + head_b ----
  if (cfg->out_dim > 0) {
    const int E = cfg->out_dim;
    int blocks = 0;
    if (int e = ln_bwd_kernel(d_out, nullptr, p.head_y, B, E, w->head_ln_g, cfg->head_ln_eps, p.dhead_y, p.dhead_yb, p.part, &blocks, st))
      return e;
    if (side_ok)
      if (int e = wf.fork()) return e;
    {
      ReduceList r;
      r.add_ln(p.part, blocks, E, gr->head_b, gr->head_ln_g, gr->head_ln_b);
      if (int e = r.launch(hs)) return e;
    }
    // d head_w [E, H] = d head_y^T . cls_b
    if (int e = wgrad(p.dhead_yb, E, E, p.cls_b, H, H, B, p, gr->head_w, hs)) return e;
    // d cls = d head_y . head_w : dgrad with the transposed head weight [H, E]
    GemmArgs g{};
    g.rows = B; g.W = (const bf16_t*)head_w_t; g.X = p.dhead_yb; g.N = H; g.K = E; g.Cf = p.dcls_f;
    if (int e = launch_gemm<EPI_F32>(g, st, "gemm_dgrad")) return e;
    dcls = p.dcls_f;
  }
  // ---- last layer's LayerNorm2 on the CLS rows only, scattered into a zero [rows, H] gradient ----
  const bool pool_mean = cfg->pool_mean != 0;
  if (pool_mean) {
    // use_mean = True: d(last layer output)[row] = d pooled[b] / len[b]; the last layer is then an ordinary layer
    hipLaunchKernelGGL(k_masked_mean_bwd, dim3(B), dim3(256), 0, st, dcls, cu_seqlens, seq_lens, H, p.G0);
    CONVDR_CHECK_LAUNCH("k_masked_mean_bwd");
  } else {
    // (its dbias output is the last layer's FFN2 bias gradient: only the CLS rows of that layer carry gradient)
    int blocks = 0;
    if (int e = ln_bwd_kernel(dcls, nullptr, p.cls_y, B, H, w->layers[NL - 1].ln2_g, cfg->ln_eps, p.dcls_y, (bf16_t*)nullptr, part2,
                              &blocks, st))
      return e;
    if (side_ok)
      if (int e = wf.fork()) return e;
    ReduceList r;
    r.add_ln(part2, blocks, H, p_hid > 0.f ? nullptr : lg_last->b2, lg_last->ln2_g, lg_last->ln2_b);
    if (int e = r.launch(hs)) return e;
    // (p.dcls_y [B, H]: the gradient of the CLS rows' pre-LayerNorm2 sums; the last layer's tail stays on those B rows)
  }
  if (side_ok) CONVDR_CHECK_HIP(hipEventRecord(wf.head_fin, ss));

  // The gradient flowing down the residual stream is cur_f (fp32) + cur_b (bf16 dgrad tile output, or null)
  float* cur_f = p.G0;
  const bf16_t* cur_b = nullptr;
  const int chunks = colsum_chunks(rows);
  CONVDR_REQUIRE(I % 8 == 0 && H % 8 == 0, "train: hidden / intermediate %% 8 != 0");
  float* other = p.G1;
  for (int l = NL - 1; l >= 0; --l) {
    const convdr_layer_weights* lw = &w->layers[l];
    const convdr_layer_weights_t* lt = &wt[l];
    const convdr_layer_grads* lg = &gr->layers[l];
    const LayerSave& s = P.layers[l];
    const LayerBwd& d = P.bwd[l];
    const bool last = l == NL - 1 && !pool_mean;   // the CLS-only tail of the last layer (see the forward's cls_tail)
    int blocks_ln2 = 0, blocks_ln1 = 0;
    const int chunks_c = colsum_chunks(B);         // column-sum chunks of the compact [B, .] matrices
    GemmArgs g{};
    float* dY1;
    if (!last) {
      // d(pre-LN2 sum Y2) = LayerNorm2 backward of the incoming gradient
      float* dY2 = other;
      if (int e = ln_bwd_kernel(cur_f, cur_b, s.Y2, rows, H, lw->ln2_g, cfg->ln_eps, dY2, d.dYb, d.part_ln2, &blocks_ln2, st,
                                drop_site(dseed, DROP_SITE_FFN_OUT, l, p_hid)))
        return e;
      other = cur_f;
      // ---- FFN2: Y2 = Hm W2^T + b2 + X1:  dHpre = (dY2 W2) * gelu'(Hpre) ----
      g.rows = rows; g.W = (const bf16_t*)lt->w2_t; g.X = d.dYb; g.N = I; g.K = H; g.Cb = d.dHpre;
      if (g_gelu_gp) {
        // the forward left gelu'(Hpre) behind (EPI_GELU_GP): multiplied in this GEMM's epilogue; the FFN1 bias gradient --
        // column sums of dHpre -- joins the QKV bias sums on the weight-gradient stream (below)
        g.Gp = s.Hpre;
        if (int e = launch_gemm<EPI_MUL_GP>(g, st, "gemm_dgrad")) return e;
      } else {
        if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_dgrad")) return e;
        ProfScope prof("dgelu_colsum", st);
        if (!(dbg_skip() & 1))
          hipLaunchKernelGGL(k_dgelu_colsum, dim3((I + 255) / 256, chunks), dim3(256), 0, st, d.dHpre, s.Hpre, rows, I, d.part_b1);
        CONVDR_CHECK_LAUNCH("k_dgelu_colsum");
      }
      // ---- FFN1: Hpre = X1 W1^T + b1;  dX1 = dHpre W1 (bf16 tile output) + dY2 (residual branch, fp32) ----
      g = GemmArgs{};
      g.rows = rows; g.W = (const bf16_t*)lt->w1_t; g.X = d.dHpre; g.N = H; g.K = I; g.Cb = p.dXb1;
      {   // Round 5: this long-K data-gradient GEMM takes 256 x 256 tiles when they fit beside the weight-gradient branch.  The
          // launcher's cost model counts every CU, but the branch of the layer above (one 160 KB-LDS workgroup per 256 x 256
          // tile of the four weight matrices: 108 for roberta-base) owns its CUs while this kernel runs: at 9.2 k rows 108
          // big tiles fit the ~148 free CUs in one round where 432 small ones need two.  Step 9.36 -> 9.21 ms (three
          // alternations, profiles/r05_ab_dgrad_tiles.txt); the QKV data-gradient GEMM the same way: -0.10 alone, nothing
          // on top of this one.  CONVDR_DGRAD_FFN1_256=0: the cost model's choice (A/B).
        static const bool off = getenv("CONVDR_DGRAD_FFN1_256") && !atoi(getenv("CONVDR_DGRAD_FFN1_256"));
        const int64_t wg_tiles = (int64_t)((I + 255) / 256) * ((H + 255) / 256) * 2 + (int64_t)((3 * H + 255) / 256 + (H + 255) / 256) * ((H + 255) / 256);
        const int64_t free_cus = device_cu_count() - (wg_tiles < device_cu_count() / 2 ? wg_tiles : device_cu_count() / 2);
        if (!off && fork_wgrad && H % 256 == 0 && (int64_t)(H / 256) * ceil_div64(rows, 256) <= free_cus) g.tile_hint = 256;
      }
      if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_dgrad")) return e;
      // ---- LayerNorm1: dY1 = LN'(dY2 + dXb1) ----
      dY1 = other;
      if (int e = ln_bwd_kernel(dY2, p.dXb1, s.Y1, rows, H, lw->ln1_g, cfg->ln_eps, dY1, d.dYb2, d.part_ln1, &blocks_ln1, st,
                                drop_site(dseed, DROP_SITE_ATTN_OUT, l, p_hid)))
        return e;
      other = dY2;
      // ---- attention output projection: Y1 = ctx Wo^T + bo + Xin ----
      g = GemmArgs{};
      g.rows = rows; g.W = (const bf16_t*)lt->wo_t; g.X = d.dYb2; g.N = H; g.K = H; g.Cb = p.dctx;
      if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_dgrad")) return e;
    } else {
      // ---- the same chain on the B CLS rows: p.dcls_y [B, H] is d(pre-LN2 sum) of those rows, every other row's is zero ----
      const DropSite dr_ffn = drop_site(dseed, DROP_SITE_FFN_OUT, l, p_hid), dr_att = drop_site(dseed, DROP_SITE_ATTN_OUT, l, p_hid);
      if (p_hid > 0.f) {
        // (the CLS-row LayerNorm backward above indexed its rows 0..B-1, not by packed row: with dropout the mask is applied
        //  here, and the FFN2 bias gradient -- column sums of the MASKED gradient -- comes from a column-sum pass)
        hipLaunchKernelGGL(k_cast_drop_f32_bf16, dim3(64), dim3(256), 0, st, p.dcls_y, p.c_dYb, (int64_t)B, H, dr_ffn, cu_seqlens);
        CONVDR_CHECK_LAUNCH("k_cast_drop_f32_bf16");
      } else {
        hipLaunchKernelGGL(k_cast_f32_bf16, dim3(64), dim3(256), 0, st, p.dcls_y, p.c_dYb, (int64_t)B * H / 4);
        CONVDR_CHECK_LAUNCH("k_cast_f32_bf16");
      }
      g.rows = B; g.W = (const bf16_t*)lt->w2_t; g.X = p.c_dYb; g.N = I; g.K = H; g.Cb = p.c_dHpre;
      if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_cls")) return e;
      hipLaunchKernelGGL(k_dgelu_colsum, dim3((I + 255) / 256, chunks_c), dim3(256), 0, st, p.c_dHpre, p.c_Hpre, (int64_t)B, I, d.part_b1);
      CONVDR_CHECK_LAUNCH("k_dgelu_colsum");
      // dX1 = dHpre W1 + dY2: the contraction slices are added straight onto the fp32 residual-branch gradient
      int ns = 1;
      if (int e = cls_projection((const bf16_t*)lt->w1_t, p.c_dHpre, B, H, I, p, st, &ns)) return e;
      hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)ceil_div64((int64_t)B * H / 4, 256)), dim3(256), 0, st, p.c_slab, ns,
                         (int64_t)B * H, (int64_t)B * H, p.c_dX1f, 0, (const float*)p.dcls_y);
      CONVDR_CHECK_LAUNCH("k_reduce_partials");
      if (int e = ln_bwd_kernel(p.c_dX1f, nullptr, p.c_Y1, B, H, lw->ln1_g, cfg->ln_eps, p.c_dY1, p.c_dYb2, d.part_ln1, &blocks_ln1, st,
                                dr_att, cu_seqlens))
        return e;
      g = GemmArgs{};
      g.rows = B; g.W = (const bf16_t*)lt->wo_t; g.X = p.c_dYb2; g.N = H; g.K = H; g.Cb = p.c_dctx;
      if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_cls")) return e;
      // d ctx: zero but for the CLS rows; the attention backward below reads the first query tile only.  The Q third of dQKV
      // beyond that tile is never written: zero.  The residual-branch gradient handed to the layer below: likewise.
      dY1 = cur_f;
      hipLaunchKernelGGL(k_cls_tail_scatter, dim3(B, 8), dim3(256), 0, st, cu_seqlens, B, H, p.c_dctx, p.dctx, d.dQKV, p.c_dY1, dY1,
                         (g_attn_bwd_fused && max_len <= ATTF_MAX_LEN) ? 64 : 128);
      CONVDR_CHECK_LAUNCH("k_cls_tail_scatter");
    }
    // ---- attention ----
    {
      // (D[h, t] = dO . O per head is computed by the dQ kernel for its own queries and handed to the dK / dV kernel
      //  through p.Drow: no separate row-dot pass)
      AttnBwdArgs a{s.QKV, p.dctx, rows, s.LSE, cfg->heads, p.Drow, s.ctx, p.Drow, p.ldt, cu_seqlens, seq_lens, H, d.dQKV, 0.125f,
                    drop_site(dseed, DROP_SITE_ATT_PROBS, l, p_att), last ? 64 : 0, p.order, s.Mbits, last ? p.c_ctx32 : nullptr,
                    (l == 5) ? (unsigned long long*)g_attn_trace : nullptr};   // (TRACE builds, layer 5: tools/dbg/attn_bwd_trace.py)
      ProfScope prof("attention_bwd", st);
      if (g_attn_bwd_fused && max_len <= ATTF_MAX_LEN) {   // (last layer: q_limit = 64, one query step)
        static DeviceOnce attr_done;
        if (attr_done.first()) {
          CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_attention_bwd_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               ATTB_FUSED_SMEM));
          CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_attention_bwd_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               ATTB_FUSED_SMEM));
        }
        if (!(dbg_skip() & 16))
          for (int rep = 0; rep < ((dbg_double() & 16) ? 2 : 1); ++rep) {
            if (a.drop.thresh) hipLaunchKernelGGL(k_attention_bwd_fused<true>, dim3(cfg->heads, B), dim3(512), ATTB_FUSED_SMEM, st, a);
            else hipLaunchKernelGGL(k_attention_bwd_fused<false>, dim3(cfg->heads, B), dim3(512), ATTB_FUSED_SMEM, st, a);
          }
      } else {
        const dim3 grid((max_len + 127) / 128, cfg->heads, B);
        hipLaunchKernelGGL(k_attention_bwd_dq, last ? dim3(1, cfg->heads, B) : grid, dim3(256), ATTB_DQ_SMEM, st, a);
        hipLaunchKernelGGL(k_attention_bwd_dkv, grid, dim3(256), ATTB_DKV_SMEM, st, a);
      }
      CONVDR_CHECK_LAUNCH("k_attention_bwd");
    }
    // ---- the layer's weight-gradient branch: every operand is complete now; it runs beside the layers below ----
    if (fork_wgrad)
      if (int e = wf.fork()) return e;
    for (int rep = 0; rep < ((dbg_double() & 128) && t_bwd_overwrite ? 2 : 1); ++rep) {
      if (dbg_skip() & 32) {   // (timing bound only: zero partials instead of the column sums)
        CONVDR_CHECK_HIP(hipMemsetAsync(d.part_bqkv, 0, sizeof(float) * (size_t)chunks * 3 * H, ss));
        if (!last && g_gelu_gp) CONVDR_CHECK_HIP(hipMemsetAsync(d.part_b1, 0, sizeof(float) * (size_t)chunks * I, ss));
      } else {
        hipLaunchKernelGGL(k_colsum_bf16, dim3((3 * H + 255) / 256, chunks), dim3(256), 0, ss, d.dQKV, rows, 3 * H, d.part_bqkv);
        if (!last && g_gelu_gp)
          hipLaunchKernelGGL(k_colsum_bf16, dim3((I + 255) / 256, chunks), dim3(256), 0, ss, d.dHpre, rows, I, d.part_b1);
      }
      CONVDR_CHECK_LAUNCH("k_colsum_bf16");
      ReduceList r;
      if (!last) r.add_ln(d.part_ln2, blocks_ln2, H, lg->b2, lg->ln2_g, lg->ln2_b);
      if (last && p_hid > 0.f) {   // see the k_cast_drop_f32_bf16 branch above (part_ln2 is free in the last layer)
        hipLaunchKernelGGL(k_colsum_bf16, dim3((H + 255) / 256, chunks_c), dim3(256), 0, ss, p.c_dYb, (int64_t)B, H, d.part_ln2);
        CONVDR_CHECK_LAUNCH("k_colsum_bf16");
        r.add(d.part_ln2, chunks_c, H, H, lg->b2);
      }
      r.add_ln(d.part_ln1, blocks_ln1, H, lg->bo, lg->ln1_g, lg->ln1_b);
      r.add(d.part_b1, last ? chunks_c : chunks, I, I, lg->b1);
      r.add(d.part_bqkv, chunks, 3 * H, 3 * H, lg->bqkv);
      if (int e = r.launch(ss)) return e;
    }
    if (!last) {
      const WgradItem items[4] = {{d.dHpre, I, I, s.X1, H, H, lg->w1},          // Hpre = X1 W1^T
                                  {d.dYb, H, H, s.Hm, I, I, lg->w2},            // Y2 = Hm W2^T
                                  {d.dQKV, 3 * H, 3 * H, s.Xin, H, H, lg->wqkv},   // QKV = Xin Wqkv^T
                                  {d.dYb2, H, H, s.ctx, H, H, lg->wo}};         // Y1 = ctx Wo^T
      static const int tail_split = getenv("CONVDR_WGRAD_TAIL_SPLIT") ? atoi(getenv("CONVDR_WGRAD_TAIL_SPLIT")) : 2;
      if (int e = wgrad_batch(items, 4, rows, p.slab, p.slab_elems, ss, l == 0 && fork_wgrad ? tail_split : 0)) return e;
    } else {
      // three of the four products contract over the B CLS rows only (one 64-token K step); the QKV projection saw every row
      const WgradItem items_c[3] = {{p.c_dHpre, I, I, p.c_X1, H, H, lg->w1}, {p.c_dYb, H, H, p.c_Hm, I, I, lg->w2},
                                    {p.c_dYb2, H, H, p.c_ctx, H, H, lg->wo}};
      if (int e = wgrad_batch(items_c, 3, B, p.slab, p.slab_elems, ss)) return e;
      const WgradItem item_qkv{d.dQKV, 3 * H, 3 * H, s.Xin, H, H, lg->wqkv};
      if (int e = wgrad_batch(&item_qkv, 1, rows, p.slab, p.slab_elems, ss)) return e;
    }
    // ---- QKV projection: QKV = Xin Wqkv^T + bqkv;  dXin = dQKV Wqkv (bf16 tile output) + dY1 (residual branch, fp32) ----
    g = GemmArgs{};
    g.rows = rows; g.W = (const bf16_t*)lt->wqkv_t; g.X = d.dQKV; g.N = H; g.K = 3 * H; g.Cb = p.dXb2;
    if (int e = launch_gemm<EPI_BF16>(g, st, "gemm_dgrad")) return e;
    cur_f = dY1; cur_b = p.dXb2;   // d(output of layer l - 1) = dY1 + dXb2
    if (int e = wf.layer_done(l, ss)) return e;
  }
  wf.layers_recorded = NL;
  // ---- embeddings ----
  {
    const int blocks = (int)(ceil_div64(rows, 4) < EMB_BWD_BLOCKS ? ceil_div64(rows, 4) : EMB_BWD_BLOCKS);
    ProfScope prof("embed_bwd", st);
    if (side_ok) CONVDR_CHECK_HIP(hipStreamWaitEvent(st, wf.head_fin, 0));   // p.part is rewritten here (long complete: first jobs of that stream)
    // "embed_bwd_deterministic": the rows' gradients go to the free fp32 stream buffer, then owners of table rows add them
    // in token-row order (train_kernels.hpp: k_embed_scatter_det)
    float* o_rows = g_embed_bwd_det ? other : nullptr;
    hipLaunchKernelGGL(k_embed_bwd, dim3(blocks), dim3(256), 0, st, cur_f, cur_b, p.tok_id, p.tok_pos, rows, H, w->word_emb, w->pos_emb,
                       w->type_emb, w->emb_ln_g, cfg->ln_eps, gr->word_emb, gr->pos_emb, p.part,
                       drop_site(dseed, DROP_SITE_EMB, 0, p_hid), o_rows);
    CONVDR_CHECK_LAUNCH("k_embed_bwd");
    if (o_rows) {
      hipLaunchKernelGGL(k_embed_scatter_det, dim3(1024), dim3(256), 0, st, o_rows, p.tok_id, p.tok_id, rows, H, cfg->vocab, gr->word_emb);
      hipLaunchKernelGGL(k_embed_scatter_det, dim3(cfg->max_pos < 1024 ? cfg->max_pos : 1024), dim3(256), 0, st, o_rows, p.tok_pos, p.tok_id,
                         rows, H, cfg->max_pos, gr->pos_emb);
      CONVDR_CHECK_LAUNCH("k_embed_scatter_det");
    }
    float* outs[3] = {gr->emb_ln_g, gr->emb_ln_b, gr->type_emb};
    ReduceList r;
    for (int k = 0; k < 3; ++k) r.add(p.part + (size_t)k * H, blocks, (int64_t)3 * H, H, outs[k]);
    if (int e = r.launch(st)) return e;
  }
  CONVDR_CHECK_HIP(hipEventRecord(wf.emb_main, st));   // convdr_backward_wait_layer(-1): the embedding gradients, ahead of the join
  if (fork_wgrad)
    if (int e = wf.join()) return e;   // every weight gradient is complete for whatever follows on `stream`
  return 0;
}

extern "C" int convdr_wgrad(const void* dy, int N, int64_t ld_dy, const void* x, int K, int64_t ld_x, int64_t rows, float* slab,
                            size_t slab_elems, float* dW, convdr_stream_t stream) {
  CONVDR_REQUIRE(N > 0 && K > 0 && rows >= 0 && ld_dy >= N && ld_x >= K && (slab == nullptr || slab_elems >= (size_t)N * K),
                 "convdr_wgrad: bad sizes N=%d K=%d rows=%lld", N, K, (long long)rows);
  const WgradItem it{(const bf16_t*)dy, N, ld_dy, (const bf16_t*)x, K, ld_x, dW};
  return wgrad_batch(&it, 1, rows, slab, slab_elems, (hipStream_t)stream);
}

// The stream of the current device's weight-gradient branches.  Re-settable (round 5: train.py's step watchdog moves the
// branch to another stream when the step is seen to have lost its overlap): every convdr_encoder_backward ends with the
// caller's stream waiting for the side stream (WgradFork::join), so BETWEEN two backward calls enqueued on one stream nothing
// is pending that a later launch on the new stream could overtake.  The stream must belong to the current device.
extern "C" int convdr_train_set_side_stream(convdr_stream_t stream) {
  CONVDR_REQUIRE(stream != nullptr, "convdr_train_set_side_stream: null stream");
  int sdev = -1, cdev = -2;
  CONVDR_CHECK_HIP(hipGetDevice(&cdev));
  if (hipStreamGetDevice((hipStream_t)stream, &sdev) != hipSuccess) { (void)hipGetLastError(); sdev = cdev; }
  CONVDR_REQUIRE(sdev == cdev, "convdr_train_set_side_stream: the stream belongs to device %d, the current device is %d", sdev, cdev);
  WgradFork& wf = WgradFork::get();
  g_ext_side[cur_device_slot()] = (hipStream_t)stream;
  if (wf.ok) wf.side = (hipStream_t)stream;
  return 0;
}

extern "C" int convdr_backward_wait_layer(int layer, convdr_stream_t stream) {
  WgradFork& wf = WgradFork::get();
  CONVDR_REQUIRE(wf.ok && layer >= -1 && layer < wf.layers_recorded,
                 "convdr_backward_wait_layer: layer %d of a backward with %d layers (none enqueued yet?)", layer,
                 wf.ok ? wf.layers_recorded : 0);
  if (layer == -1) {   // the embedding tables + embedding LayerNorm: written by the main chain's last kernels, before it joins the branch
    CONVDR_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, wf.emb_main, 0));
    return 0;
  }
  CONVDR_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, wf.layer_main[layer], 0));
  CONVDR_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, wf.layer_side[layer], 0));
  return 0;
}

// fp32 [n, k] row-major -> bf16 [k, n]
extern "C" int convdr_transpose_f32_bf16(const float* x, int n, int k, void* y, convdr_stream_t stream) {
  hipLaunchKernelGGL(k_transpose_f32_bf16, dim3((k + 63) / 64, (n + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, n, k,
                     (bf16_t*)y);
  CONVDR_CHECK_LAUNCH("k_transpose_f32_bf16");
  return 0;
}

extern "C" int convdr_mse_fwd_bwd(const float* s, const float* t, int64_t n, float grad_scale, float* loss, float* ds,
                                  convdr_stream_t stream) {
  CONVDR_REQUIRE(n > 0, "convdr_mse_fwd_bwd: n = %lld", (long long)n);
  hipLaunchKernelGGL(k_mse_fwd_bwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, s, t, n, grad_scale, loss, ds);
  CONVDR_CHECK_LAUNCH("k_mse_fwd_bwd");
  return 0;
}

extern "C" int convdr_rank_ce_fwd_bwd(const float* embs, const float* docs, int B, int K, int E, float grad_scale,
                                      float* loss_per_query, float* d_embs, int accumulate, convdr_stream_t stream) {
  CONVDR_REQUIRE(B > 0 && K > 0 && K <= 64 && E > 0, "convdr_rank_ce_fwd_bwd: bad sizes B=%d K=%d E=%d", B, K, E);
  hipLaunchKernelGGL(k_rank_ce_fwd_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, embs, docs, B, K, E, grad_scale,
                     loss_per_query, d_embs, accumulate);
  CONVDR_CHECK_LAUNCH("k_rank_ce_fwd_bwd");
  return 0;
}

extern "C" int convdr_pair_nll_fwd_bwd(const float* q, const float* a, const float* b, const float* bias_a, const float* bias_b,
                                       int B, int C, int E, float grad_scale, float* loss_per_query, float* d_q, float* d_a,
                                       float* d_b, convdr_stream_t stream) {
  CONVDR_REQUIRE(B > 0 && C >= 1 && C <= 32 && E > 0, "convdr_pair_nll_fwd_bwd: bad sizes B=%d C=%d E=%d", B, C, E);
  hipLaunchKernelGGL(k_pair_nll_fwd_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, q, a, b, bias_a, bias_b, B, C, E, grad_scale,
                     loss_per_query, d_q, d_a, d_b);
  CONVDR_CHECK_LAUNCH("k_pair_nll_fwd_bwd");
  return 0;
}

extern "C" int convdr_inbatch_ce_fwd_bwd(const float* embs, const float* docs, int B, int N, int E, const int32_t* pos,
                                         float grad_scale, float* loss_per_query, float* d_embs, int accumulate,
                                         convdr_stream_t stream) {
  CONVDR_REQUIRE(B > 0 && N > 0 && N <= 16384 && E > 0, "convdr_inbatch_ce_fwd_bwd: bad sizes B=%d N=%d E=%d", B, N, E);
  static DeviceOnce attr_done;
  if (attr_done.first())
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_inbatch_ce_fwd_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(k_inbatch_ce_fwd_bwd, dim3(B), dim3(256), (size_t)N * 4, (hipStream_t)stream, embs, docs, B, N, E, pos,
                     grad_scale, loss_per_query, d_embs, accumulate);
  CONVDR_CHECK_LAUNCH("k_inbatch_ce_fwd_bwd");
  return 0;
}

extern "C" int convdr_grad_norm_clip(float* grads, int64_t n, float max_norm, float pre_scale, float* scratch /* >= 1024 floats */,
                                     float* norm_and_coef /* [2] */, int apply, convdr_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)(ceil_div64(n, 256) < 1024 ? ceil_div64(n, 256) : 1024);
  hipLaunchKernelGGL(k_sumsq_partial, dim3(blocks), dim3(256), 0, st, grads, n, scratch);
  hipLaunchKernelGGL(k_norm_finish, dim3(1), dim3(64), 0, st, scratch, blocks, max_norm, pre_scale, norm_and_coef);
  if (apply) hipLaunchKernelGGL(k_scale_inplace, dim3(blocks), dim3(256), 0, st, grads, n, norm_and_coef + 1);
  CONVDR_CHECK_LAUNCH("convdr_grad_norm_clip");
  return 0;
}

extern "C" int convdr_grad_sumsq(const float* x, int64_t n, float* partials, int nblocks, convdr_stream_t stream) {
  CONVDR_REQUIRE(n >= 0 && nblocks >= 1 && nblocks <= 1024, "convdr_grad_sumsq: bad n / nblocks (%lld, %d)", (long long)n, nblocks);
  if (dbg_skip() & 128) CONVDR_CHECK_HIP(hipMemsetAsync(partials, 0, sizeof(float) * (size_t)nblocks, (hipStream_t)stream));   // (timing bound only)
  else
    for (int rep = 0; rep < ((dbg_double() & 256) ? 2 : 1); ++rep)
      hipLaunchKernelGGL(k_sumsq_partial, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, x, n, partials);
  CONVDR_CHECK_LAUNCH("k_sumsq_partial");
  return 0;
}

extern "C" int convdr_grad_norm_finish(const float* partials, int count, float max_norm, float pre_scale, float* norm_and_coef,
                                       convdr_stream_t stream) {
  CONVDR_REQUIRE(count >= 1, "convdr_grad_norm_finish: no partial sums");
  hipLaunchKernelGGL(k_norm_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, count, max_norm, pre_scale, norm_and_coef);
  CONVDR_CHECK_LAUNCH("k_norm_finish");
  return 0;
}

extern "C" int convdr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                                 double beta2, double eps, double weight_decay, int step, int correct_bias,
                                 const float* grad_scale, convdr_stream_t stream) {
  return convdr_adamw_step_packed(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, correct_bias, grad_scale, nullptr, 0,
                                  stream);
}

extern "C" int convdr_adamw_step_packed(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                                        double beta2, double eps, double weight_decay, int step, int correct_bias,
                                        const float* grad_scale, void* bf16_copy, int64_t bf16_first,
                                        convdr_stream_t stream) {
  CONVDR_REQUIRE(n >= 0 && step >= 1, "convdr_adamw_step: bad n/step");
  CONVDR_REQUIRE(bf16_copy == nullptr || (bf16_first >= 0 && bf16_first % 4 == 0 && ((uintptr_t)bf16_copy & 7) == 0),
                 "convdr_adamw_step_packed: the bf16 copy must start at a multiple of 4 elements (%lld) and be 8-byte aligned",
                 (long long)bf16_first);
  if (n == 0) return 0;
  double step_size = lr;
  if (correct_bias) step_size = lr * sqrt(1.0 - pow(beta2, (double)step)) / (1.0 - pow(beta1, (double)step));
  // One 256-thread block per CU, 16 bytes per lane and array: seven streams (4 read, 3 written) are fastest with few
  // requesters -- measured on the 110 M-parameter arena: 4096 scalar blocks 690-710 us, 1024 scalar 565-640, 256 x float4
  // 605-613, 384 x float4 685 (uneven over the CUs), 128 x float4 951
  const int maxb = device_cu_count();
  const int blocks = (int)(ceil_div64(n, 1024) < maxb ? ceil_div64(n, 1024) : maxb);
  hipLaunchKernelGGL(k_adamw_hf, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)beta1,
                     (float)beta2, (float)eps, (float)weight_decay, (float)step_size, grad_scale, (bf16_t*)bf16_copy, bf16_first);
  CONVDR_CHECK_LAUNCH("k_adamw_hf");
  return 0;
}

// x[i] *= scale[0] (device scalar)
extern "C" int convdr_scale_f32(float* x, int64_t n, const float* scale, convdr_stream_t stream) {
  if (n <= 0) return 0;
  const int blocks = (int)(ceil_div64(n, 256) < 2048 ? ceil_div64(n, 256) : 2048);
  hipLaunchKernelGGL(k_scale_inplace, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scale);
  CONVDR_CHECK_LAUNCH("k_scale_inplace");
  return 0;
}

static int pack_transposed(const void* base, bool src_bf16, int count, const int64_t* src_off, const int32_t* n, const int32_t* k,
                           const int64_t* dst_off, void* out, hipStream_t stream) {
  for (int i0 = 0; i0 < count; i0 += TR_MAX_JOBS) {
    TransposeJobs a{};
    int tiles = 0;
    for (int i = i0; i < count && i < i0 + TR_MAX_JOBS; ++i) {
      TransposeJob& q = a.j[a.count++];
      q.in = src_bf16 ? (const void*)((const bf16_t*)base + src_off[i]) : (const void*)((const float*)base + src_off[i]);
      q.out = (bf16_t*)out + dst_off[i];
      q.n = n[i]; q.k = k[i];
      q.tiles_k = (k[i] + 63) / 64;
      tiles += q.tiles_k * ((n[i] + 63) / 64);
      q.tile_end = tiles;
    }
    if (tiles == 0) continue;
    if (dbg_skip() & 256) continue;   // (timing bound only: the data-gradient GEMMs then run on stale transposed weights)
    for (int rep = 0; rep < ((dbg_skip() & 512) ? 2 : 1); ++rep) {   // (512, timing only: the launch twice -- what one more of it costs the step)
      if (src_bf16) hipLaunchKernelGGL(k_transpose_bf16_batch<true>, dim3((unsigned)tiles), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(k_transpose_bf16_batch<false>, dim3((unsigned)tiles), dim3(256), 0, stream, a);
    }
  }
  CONVDR_CHECK_LAUNCH("k_transpose_bf16_batch");
  return 0;
}

// Batched weight packing for the data-gradient GEMMs: for i < count, fp32 [n[i], k[i]] at base + src_off[i] (elements)
// -> bf16 [k[i], n[i]] at out + dst_off[i] (elements).  Host arrays; one launch per 64 matrices, no Python round trips.
extern "C" int convdr_pack_transposed(const float* base, int count, const int64_t* src_off, const int32_t* n, const int32_t* k,
                                      const int64_t* dst_off, void* out, convdr_stream_t stream) {
  return pack_transposed(base, false, count, src_off, n, k, dst_off, out, (hipStream_t)stream);
}

// The same from a bf16 source (the bf16 copy of the weights that convdr_adamw_step_packed keeps current): bits are moved, not rounded
extern "C" int convdr_pack_transposed_bf16(const void* base, int count, const int64_t* src_off, const int32_t* n, const int32_t* k,
                                           const int64_t* dst_off, void* out, convdr_stream_t stream) {
  return pack_transposed(base, true, count, src_off, n, k, dst_off, out, (hipStream_t)stream);
}
