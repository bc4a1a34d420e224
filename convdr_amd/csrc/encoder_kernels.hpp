// Device kernels of the BERT/RoBERTa dual-encoder forward on gfx950 (shared by encoder.hip and train.hip).
//
// Replaces the third-party arithmetic behind  self.roberta(input_ids, attention_mask)
// (/root/reference/model/models.py:141-142) and BertModel.forward (:208-209): HF transformers==2.3.0
// embeddings + 12 x [self-attention, output dense + residual + LayerNorm, FFN + residual + LayerNorm].
//
// Data layout in HBM ("packed rows"): only tokens with attention_mask == 1 are computed.  Sequence b owns
// rows [cu[b], cu[b] + len[b]) of every [rows, *] activation matrix; cu[b] is a multiple of 8 so that 16-byte
// accesses along the token axis stay aligned.  CLS-only pooling (models.py:43, use_mean = False for every
// registered config) makes this identical to the reference's padded computation (SURVEY.md §7 hard part 7).
//   X    [rows, H]  bf16   layer input / LayerNorm output (MFMA operand and residual)
//   Y    [rows, H]  f32    pre-LayerNorm sums (GEMM epilogue output, LayerNorm input)
//   Q, K [rows, H]  bf16   row-major, head h = columns [64 h, 64 h + 64)
//   Vt   [H, ldt]   bf16   V transposed (feature-major, token-contiguous): the P.V contraction runs over keys,
//                          and MFMA wants the contraction index contiguous per lane, so the QKV GEMM epilogue
//                          writes V already transposed instead of transposing it inside the attention kernel
//   ctx  [rows, H]  bf16   attention output
//   Hm   [rows, I]  bf16   gelu(FFN1)
#pragma once
#include "../../include/convdr_hip.h"
#include "dropout.hpp"
#include "gemm_nt.hpp"

namespace convdr {

// ---------------------------------------------------------------------------------------------
// pack: one wave per sequence.  Compacts the mask == 1 tokens of ids[b, :] to rows cu[b]..; position ids:
//   RoBERTa: cumsum(ids != pad_idx) * (ids != pad_idx) + pad_idx over the FULL row (masked positions count,
//            the reference pads with id 0 which is not RoBERTa's pad id 1 -- utils/util.py:146-185);
//   BERT:    the column index.
// Alignment rows [cu[b] + len, cu[b+1]) get token id -1 (embedding kernel writes zeros).
// ---------------------------------------------------------------------------------------------
// ids: int64 [B, L] (the drivers' .long() tensors) or int32 [B, L] (token-cache records, ids32 != 0);
// mask == nullptr means "prefix mask": position l is kept iff l < lens[b] (right padding).
static __global__ void __launch_bounds__(256) k_seq_pack(const void* __restrict__ ids_v, int ids32,
                                                  const int64_t* __restrict__ mask, const int32_t* __restrict__ lens,
                                                  int B, int L, const int32_t* __restrict__ cu, int kind, int pad_idx,
                                                  int max_pos, int vocab, int32_t* __restrict__ tok_id,
                                                  int32_t* __restrict__ tok_pos, int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int base = cu[b], end = cu[b + 1];
  const int64_t* ids = (const int64_t*)ids_v;
  const int32_t* ids_i = (const int32_t*)ids_v;
  const int len_b = lens[b];
  int kept = 0, nonpad = 0;
  int flags = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    const bool valid = l < L;
    int64_t id = valid ? (ids32 ? (int64_t)ids_i[(int64_t)b * L + l] : ids[(int64_t)b * L + l]) : (int64_t)pad_idx;
    const bool m = valid && (mask ? mask[(int64_t)b * L + l] != 0 : l < len_b);
    if (l == 0 && !m) flags |= CONVDR_ENC_STATUS_BAD_MASK;   // the CLS position must be unmasked
    const bool np = valid && id != pad_idx;
    const unsigned long long bm = __ballot(m), bnp = __ballot(np);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (m) {
      // The reference's embedding lookup raises IndexError for an id outside the table (models.py:141-142 -> nn.Embedding).
      // Here the id is clamped (no out-of-bounds read or atomic in the backward) and the batch is flagged; the host
      // raises at its next look at the status word.
      if (id < 0 || id >= (int64_t)vocab) { flags |= CONVDR_ENC_STATUS_BAD_TOKEN; id = 0; }
      const int row = base + kept + __popcll(bm & lt);
      if (row < end) {
        int p = kind == 0 ? (np ? nonpad + __popcll(bnp & (lt | (1ull << lane))) + pad_idx : pad_idx) : l;
        p = p < max_pos ? p : max_pos - 1;
        tok_id[row] = (int)id;
        tok_pos[row] = p;
      }
    }
    kept += __popcll(bm);
    nonpad += __popcll(bnp);
  }
  if (kept != len_b) flags |= CONVDR_ENC_STATUS_BAD_LENS;    // seq_lens[b] != mask[b].sum() (or > L)
  if (__ballot(flags != 0) != 0ull) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) flags |= __shfl_xor(flags, o, 64);
    if (lane == 0) atomicOr(status, flags);
  }
  for (int r = base + kept + lane; r < end; r += 64) {
    tok_id[r] = -1;
    tok_pos[r] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm helpers: one wave per row, H <= 1024, H % 4 == 0; lane owns elements 256 j + 4 lane + c.
// Biased variance, eps inside the sqrt (torch.nn.LayerNorm).
// ---------------------------------------------------------------------------------------------
struct LnRow {
  float4 v[4];
};

__device__ __forceinline__ void ln_normalize(LnRow& x, int H, int lane, float eps, const float* __restrict__ g,
                                             const float* __restrict__ b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (256 * j + 4 * lane < H) s += x.v[j].x + x.v[j].y + x.v[j].z + x.v[j].w;
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (256 * j + 4 * lane < H) {
      const float a = x.v[j].x - mean, c = x.v[j].y - mean, d = x.v[j].z - mean, e = x.v[j].w - mean;
      q += a * a + c * c + d * d + e * e;
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = 256 * j + 4 * lane;
    if (e0 < H) {
      const float4 gg = *(const float4*)(g + e0), bb = *(const float4*)(b + e0);
      x.v[j].x = (x.v[j].x - mean) * rstd * gg.x + bb.x;
      x.v[j].y = (x.v[j].y - mean) * rstd * gg.y + bb.y;
      x.v[j].z = (x.v[j].z - mean) * rstd * gg.z + bb.z;
      x.v[j].w = (x.v[j].w - mean) * rstd * gg.w + bb.w;
    }
  }
}

__device__ __forceinline__ void ln_store(const LnRow& x, int H, int lane, bf16_t* __restrict__ xb,
                                         float* __restrict__ xf) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = 256 * j + 4 * lane;
    if (e0 < H) {
      if (xb) {
        uint2 o;
        o.x = pack_bf16x2(x.v[j].x, x.v[j].y);
        o.y = pack_bf16x2(x.v[j].z, x.v[j].w);
        *(uint2*)(xb + e0) = o;
      }
      if (xf) *(float4*)(xf + e0) = x.v[j];
    }
  }
}

// embeddings: LayerNorm(word[id] + pos[p] + type[0]) -> bf16 X
static __global__ void __launch_bounds__(256) k_embed_ln(const int32_t* __restrict__ tok_id,
                                                  const int32_t* __restrict__ tok_pos, int64_t rows, int H,
                                                  const float* __restrict__ word, const float* __restrict__ pos,
                                                  const float* __restrict__ type0, const float* __restrict__ g,
                                                  const float* __restrict__ b, float eps, bf16_t* __restrict__ X,
                                                  const DropSite drop) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int id = tok_id[row];
  LnRow x;
  if (id < 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) x.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    ln_store(x, H, lane, X + row * H, nullptr);
    return;
  }
  const float* w = word + (int64_t)id * H;
  const float* p = pos + (int64_t)tok_pos[row] * H;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = 256 * j + 4 * lane;
    if (e0 < H) {
      const float4 a = *(const float4*)(w + e0), c = *(const float4*)(p + e0), t = *(const float4*)(type0 + e0);
      x.v[j] = make_float4(a.x + c.x + t.x, a.y + c.y + t.y, a.z + c.z + t.z, a.w + c.w + t.w);
    }
  }
  ln_normalize(x, H, lane, eps, g, b);
  if (drop.thresh) {   // embedding dropout (training)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        float m0, m1, m2, m3;
        drop_hidden4(drop, row, e0, H, m0, m1, m2, m3);
        x.v[j].x *= m0; x.v[j].y *= m1; x.v[j].z *= m2; x.v[j].w *= m3;
      }
    }
  }
  ln_store(x, H, lane, X + row * H, nullptr);
}

// rows of fp32 Y -> LayerNorm -> bf16 Xb (and/or fp32 Xf).  `gather` (optional) maps output row -> input row.
static __global__ void __launch_bounds__(256) k_layernorm(const float* __restrict__ Y, int64_t rows, int H,
                                                   const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                   bf16_t* __restrict__ Xb, float* __restrict__ Xf) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  LnRow x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = 256 * j + 4 * lane;
    if (e0 < H) x.v[j] = *(const float4*)(Y + row * H + e0);
  }
  ln_normalize(x, H, lane, eps, g, b);
  ln_store(x, H, lane, Xb ? Xb + row * H : nullptr, Xf ? Xf + row * H : nullptr);
}

// The same for H == 256 J with the bf16 output only (the encoder layers' case), as straight-line code: the kernel above keeps
// each 16-byte group behind an `e0 < H` guard, and hipcc gives every guarded load of gamma / beta -- which it cannot hoist
// over the guard -- its own block and its own s_waitcnt behind the two row reductions: three dependent L2 round trips per
// row.  Here the row, gamma and beta are requested together.  Same formulas (see k_layernorm_bwd_rows, train_kernels.hpp).
template <int J>
static __global__ void __launch_bounds__(256) k_layernorm_rows(const float* __restrict__ Y, int64_t rows,
                                                               const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                               bf16_t* __restrict__ Xb) {
  constexpr int H = 256 * J;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 x[J], gg[J], bb[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int e0 = 256 * j + 4 * lane;
    x[j] = *(const float4*)(Y + row * H + e0);
    gg[j] = *(const float4*)(g + e0);
    bb[j] = *(const float4*)(b + e0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) s += x[j].x + x[j].y + x[j].z + x[j].w;
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const float a = x[j].x - mean, c = x[j].y - mean, d = x[j].z - mean, e = x[j].w - mean;
    q += a * a + c * c + d * d + e * e;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int e0 = 256 * j + 4 * lane;
    uint2 o;
    o.x = pack_bf16x2((x[j].x - mean) * rstd * gg[j].x + bb[j].x, (x[j].y - mean) * rstd * gg[j].y + bb[j].y);
    o.y = pack_bf16x2((x[j].z - mean) * rstd * gg[j].z + bb[j].z, (x[j].w - mean) * rstd * gg[j].w + bb[j].w);
    *(uint2*)(Xb + row * H + e0) = o;
  }
}

// Split-contraction projection + residual + LayerNorm for FEW rows (H == 256 J): a K = 3072 projection of a few thousand rows is
// 126 workgroups of 48 K steps -- a latency chain on a quarter of the chip (the frozen teacher's 64 x ~40-token targets, a query
// batch of the evaluation loop).  The launcher cuts the contraction into `nsplit` slices (EPI_SLAB_F32: slab[s][row][H], no
// bias), and this kernel finishes them: y = bias + sum_s slab[s] (fixed order) + residual, X = LayerNorm(y) -> bf16.
template <int J>
static __global__ void __launch_bounds__(256) k_slab_finish_ln(const float* __restrict__ slab, int nsplit, int64_t rows,
                                                               const float* __restrict__ bias, const bf16_t* __restrict__ R,
                                                               const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                               bf16_t* __restrict__ Xb) {
  constexpr int H = 256 * J;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 x[J], gg[J], bb[J], part[4][J];
  uint2 r[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int e0 = 256 * j + 4 * lane;
    x[j] = *(const float4*)(bias + e0);
    gg[j] = *(const float4*)(g + e0);
    bb[j] = *(const float4*)(b + e0);
    r[j] = *(const uint2*)(R + row * H + e0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      part[s][j] = *(const float4*)(slab + ((int64_t)(s < nsplit ? s : 0) * rows + row) * H + e0);   // (clamped: no branch around a load)
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < nsplit) { x[j].x += part[s][j].x; x[j].y += part[s][j].y; x[j].z += part[s][j].z; x[j].w += part[s][j].w; }
    x[j].x += __uint_as_float(r[j].x << 16); x[j].y += __uint_as_float(r[j].x & 0xffff0000u);
    x[j].z += __uint_as_float(r[j].y << 16); x[j].w += __uint_as_float(r[j].y & 0xffff0000u);
    sum += x[j].x + x[j].y + x[j].z + x[j].w;
  }
  const float mean = wave_sum(sum) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const float a = x[j].x - mean, c = x[j].y - mean, d = x[j].z - mean, e = x[j].w - mean;
    q += a * a + c * c + d * d + e * e;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int e0 = 256 * j + 4 * lane;
    uint2 o;
    o.x = pack_bf16x2((x[j].x - mean) * rstd * gg[j].x + bb[j].x, (x[j].y - mean) * rstd * gg[j].y + bb[j].y);
    o.y = pack_bf16x2((x[j].z - mean) * rstd * gg[j].z + bb[j].z, (x[j].w - mean) * rstd * gg[j].w + bb[j].w);
    *(uint2*)(Xb + row * H + e0) = o;
  }
}

// fp32 rows -> LayerNorm -> bf16 rows: the straight-line kernel where it applies (convdr_set_option "ln_rows" 0: never)
inline int64_t g_ln_rows = 1;
static inline void launch_layernorm_bf16(const float* Y, int64_t rows, int H, const float* g, const float* b, float eps, bf16_t* Xb,
                                         hipStream_t st) {
  if (g_ln_rows && H == 768)
    hipLaunchKernelGGL(k_layernorm_rows<3>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, Y, rows, g, b, eps, Xb);
  else
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, Y, rows, H, g, b, eps, Xb, (float*)nullptr);
}

// out[b, :] = in[cu[b], :]  (CLS rows), bf16 and/or fp32
static __global__ void __launch_bounds__(256) k_gather_cls(const int32_t* __restrict__ cu, int B, int H,
                                                    const bf16_t* __restrict__ Xb, const float* __restrict__ Xf,
                                                    bf16_t* __restrict__ Ob, float* __restrict__ Of) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t row = cu[b];
  for (int e0 = 4 * lane; e0 < H; e0 += 256) {
    if (Ob) *(uint2*)(Ob + (int64_t)b * H + e0) = *(const uint2*)(Xb + row * H + e0);
    if (Of) *(float4*)(Of + (int64_t)b * H + e0) = *(const float4*)(Xf + row * H + e0);
  }
}

// EmbeddingMixin.masked_mean (models.py:32-35): out[b, :] = sum over the sequence's tokens of X[row, :] / len[b].
// One workgroup per sequence; thread t owns columns 4 t .. 4 t + 3 (H <= 1024); rows are read whole (coalesced).
static __global__ void __launch_bounds__(256) k_masked_mean(const bf16_t* __restrict__ X, const int32_t* __restrict__ cu,
                                                     const int32_t* __restrict__ lens, int H, bf16_t* __restrict__ Ob,
                                                     float* __restrict__ Of) {
  const int b = blockIdx.x, c = 4 * threadIdx.x;
  if (c >= H) return;
  const int64_t base = cu[b];
  const int len = lens[b];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = 0; t < len; ++t) {
    const uint2 v = *(const uint2*)(X + (base + t) * H + c);
    s.x += __uint_as_float(v.x << 16); s.y += __uint_as_float(v.x & 0xffff0000u);
    s.z += __uint_as_float(v.y << 16); s.w += __uint_as_float(v.y & 0xffff0000u);
  }
  const float inv = 1.f / (float)len;
  s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
  if (Of) *(float4*)(Of + (int64_t)b * H + c) = s;
  if (Ob) {
    uint2 o;
    o.x = pack_bf16x2(s.x, s.y);
    o.y = pack_bf16x2(s.z, s.w);
    *(uint2*)(Ob + (int64_t)b * H + c) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// GEMM with fused epilogues.  The engine computes a 128 x 128 tile  acc[r][l] = sum_k Rm[r0+r][k] Lm[l0+l][k]
// where the "R" operand's index lands on accumulator REGISTERS (4 consecutive r per register quad) and the
// "L" operand's index on LANES.  Every output below is stored at  out[l * ld + r]  (r contiguous), i.e. 8-byte
// (bf16) / 16-byte (fp32) pieces per lane, so:
//    token-major outputs  C[token][feature]:  R = weight rows (features), L = activation rows (tokens)
//    V^T                  Vt[feature][token]: R = tokens,                 L = features   (roles swapped)
// ---------------------------------------------------------------------------------------------
enum { EPI_BF16 = 0, EPI_GELU_BF16 = 1, EPI_RESID_F32 = 2, EPI_QKV = 3, EPI_F32 = 4,
       EPI_GELU_SAVE = 5,   // training FFN1: Cb = gelu(y) and Cb2 = y (pre-activation, for the backward)
       /* 6 was the gelu'-multiplying dgrad epilogue of round 1; that pass is k_dgelu_colsum now */
       EPI_SLAB_F32 = 7,    // wgrad partial: Cf[split][rows][N] = acc (no bias)
       // FFN1 of the inference forward when FFN2 is the row-complete projection + LayerNorm kernel: Cb = gelu(y) in the
       // BLOCKED layout [rows / 32][N / 8][32 tokens][8 features] (hm_blocked_offset).  In the accumulator layout lanes
       // (token, hi = 0 / 1) hold the two halves of a feature octet; one v_permlane32_swap per dword hands each lane a
       // whole octet (16 bytes), and in this layout the 32 lanes of a half-wave then write 512 CONTIGUOUS bytes: the
       // tile leaves straight from the registers in whole lines -- no LDS park, no barrier, no re-read -- and the
       // consumer's LDS-DMA (16 tokens x 4 octets per instruction) reads whole lines too.
       EPI_GELU_BLK = 8,
       // Training FFN1 of the full layers (round 5): Cb = gelu(y) row-major (FFN2's operand) and Cb2 = gelu'(y) in the BLOCKED
       // layout -- the derivative is evaluated HERE, where Q(|y|) of the forward's own fit is already in a register (one
       // quartic more, no second transcendental), and leaves register-direct in whole lines like EPI_GELU_BLK's tile.  The
       // backward's FFN2 data-gradient GEMM (EPI_MUL_GP) multiplies its accumulators by it in its epilogue: the separate
       // gelu' pass over [rows, I] (k_dgelu_colsum: 41 us per layer on the activation-gradient chain) is gone.
       EPI_GELU_GP = 9,
       EPI_MUL_GP = 10 };   // Cb = (acc) * Gp: Gp = the blocked gelu' image EPI_GELU_GP wrote
// element offset of (token t, feature f) in the blocked layout of a [rows, N] matrix
__host__ __device__ inline int64_t hm_blocked_offset(int64_t t, int f, int N) {
  return ((t >> 5) * (N >> 3) + (f >> 3)) * 256 + (t & 31) * 8 + (f & 7);
}

struct GemmArgs {
  const bf16_t* W;    // [N, K] weights (features)
  const bf16_t* X;    // [rows, K] activations (tokens)
  int64_t rows;       // tokens
  int N, K;
  int64_t ldw, ldx;   // leading dimensions of W / X in elements (0 -> K)
  int k_split_len;    // EPI_SLAB_F32: contraction slice per blockIdx.y (0 -> whole K)
  const float* bias;  // [N] or nullptr (zeros)
  bf16_t* Cb;         // EPI_BF16 / EPI_GELU_BF16: [rows, N]
  bf16_t* Cb2;        // EPI_GELU_SAVE: pre-activation [rows, N]
  float* Cf;          // EPI_RESID_F32 / EPI_F32:  [rows, N]
  const bf16_t* Gp;   // EPI_MUL_GP: gelu'(pre-activation), blocked layout [rows / 32][N / 8][32][8] (written through Cb2 by EPI_GELU_GP)
  const bf16_t* R;    // EPI_RESID_F32: residual [rows, N] bf16
  const float* Rf;    // EPI_RESID_F32: fp32 residual instead of R when non-null (gradient residual stream)
  bf16_t *Qo, *Ko, *Vt;  // EPI_QKV (N = 3H): Q [rows, H], K [rows, H], Vt [H, ldt]
  int third0;            // EPI_QKV: 1 = W / bias start at the K third and N = 2H (last layer: Q is needed for the CLS rows only)
  int qk_blocked;        // EPI_QKV: Q and K go to the blocked layout [rows / 32][H / 8][32][8] (k_attention_fwd<.., QK_BLK>)
  int H;
  int64_t ldt;
  int tilesN, tilesT;
  int tile_hint;      // launcher: 256 = take 256 x 256 tiles when the shape allows, whatever the cost model says (0 = cost model)
  int nt_out;         // bf16 tile outputs leave non-temporally (set by the launcher for outputs of 128 MB and more: what the next
                      // kernel streams from HBM anyway stays out of L2; a training-size output its consumer finds in L2 does not)
  int dbg_same_tile;  // experiment: every workgroup reads tile (0, 0) (all-L2-hit upper bound); results are garbage
  int dbg_skip_epi;   // experiment: 1 no epilogue, 3 no global stores of bf16 tiles, 4 every tile stores to tile 0 (garbage results)
  int dbg_prelanded;  // experiment: a tile's first K chunks are not waited for -- the prologue as if perfectly hidden (garbage results)
  unsigned long long* trace;   // experiment: [workgroup][64 tiles][16 phases] s_memtime stamps of wave 0 (or null)
  // measurement (bench.py): workgroup 0 stamps {shader cycles (s_memtime), 100 MHz real time (s_memrealtime)} when it starts
  // and when it has finished its last tile: [4] -> the shader clock this launch actually ran at (the part is power-managed:
  // 2.0-2.1 GHz under these kernels, not the 2.4 GHz behind the datasheet peak).  Null: off.
  unsigned long long* clock_probe;
  DropSite drop;      // EPI_RESID_F32 (training): dropout on the dense output (bias included) before the residual add; thresh 0 = off
};


// HF "gelu" is x * 0.5 * (1 + erf(x / sqrt 2)) = x Phi(x).  With t = |x| and the normal tail Q(t) = 1 - Phi(t):
//     x Phi(x) = max(x, 0) - t Q(t),
// and Q(t) = 0.5 exp2(-(c1 t + c2 t^2 + c3 t^3 + c4 t^4)) fitted (minimax on the product t Q, t <= 9; beyond that
// t Q < 1e-17 and t is clamped because the quartic turns around).  Max |error| against the exact erf form is 8.8e-6 over
// all x in fp32 (tests/test_gelu_fit_cpu.py) -- three orders below the bf16 rounding of the stored activation -- for
// 7 VALU + ONE transcendental per element (the logistic form this replaces, 1 / (1 + exp(-x (a + b u + c u^2))), needed
// v_exp_f32 and v_rcp_f32 and was 2.7e-5 off; A&S 7.1.26 needs 15 + 2).  Measured: 3x closer to the erf form, 0.2 % off
// the FFN1 kernel -- the epilogue's time is in parking, barriers and stores, not in this arithmetic.
// The 0.5 rides in the exponent (-1).
__device__ __forceinline__ float gelu_tail(float x) {
  const float t = fminf(fabsf(x), 9.f);
  float p = fmaf(t, 0.0041585f, -0.04571999f);
  p = fmaf(p, t, -0.46495319f);
  p = fmaf(p, t, -1.14955714f);
  const float q = __builtin_amdgcn_exp2f(fmaf(p, t, -1.f));   // Q(t)
  return fmaf(-t, q, fmaxf(x, 0.f));
}
// The same function on two elements, shaped for the VALU-bound GELU epilogue (13 issue slots per element as hipcc compiles
// the scalar form: fminf / fmaxf on raw MFMA results each cost a canonicalising v_max on top, and the exp is a
// quarter-rate instruction): min / max as single instructions (inline asm: |x| is a source modifier), the five FMAs as
// v_pk_fma_f32 on the pair -> 9 slots per element.  Bit-identical to gelu_tail for finite inputs.
__device__ __forceinline__ void gelu_tail2(float& x0, float& x1) {
  f32x2_t t, r;
  float a, b;
  const float nine = 9.f;
  asm("v_min_f32 %0, |%1|, %2" : "=v"(a) : "v"(x0), "s"(nine));
  asm("v_min_f32 %0, |%1|, %2" : "=v"(b) : "v"(x1), "s"(nine));
  t.x = a; t.y = b;
  asm("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(x0));
  asm("v_max_f32 %0, 0, %1" : "=v"(b) : "v"(x1));
  r.x = a; r.y = b;
  const f32x2_t c3 = {0.0041585f, 0.0041585f}, c2 = {-0.04571999f, -0.04571999f}, c1 = {-0.46495319f, -0.46495319f},
                c0 = {-1.14955714f, -1.14955714f}, m1 = {-1.f, -1.f};
  f32x2_t p = __builtin_elementwise_fma(t, c3, c2);
  p = __builtin_elementwise_fma(p, t, c1);
  p = __builtin_elementwise_fma(p, t, c0);
  p = __builtin_elementwise_fma(p, t, m1);
  f32x2_t q;
  q.x = __builtin_amdgcn_exp2f(p.x);
  q.y = __builtin_amdgcn_exp2f(p.y);
  r = __builtin_elementwise_fma(-t, q, r);
  x0 = r.x; x1 = r.y;
}

// d/dx [x * Phi(x)] = Phi(x) + x * phi(x), from the same tail fit: Phi(x) = 1 - Q(|x|) (x >= 0) or Q(|x|), and
// x phi(x) = x / sqrt(2 pi) * exp2(-x^2 log2(e) / 2).  Max |error| 3.9e-5 over all x (tests/test_gelu_fit_cpu.py), two
// orders below the bf16 rounding of the gradient it multiplies; 11 VALU + 2 transcendentals (the A&S erf form this
// replaces: ~20 + 2).
__device__ __forceinline__ float gelu_grad(float x) {
  const float t = fminf(fabsf(x), 9.f);
  float p = fmaf(t, 0.0041585f, -0.04571999f);
  p = fmaf(p, t, -0.46495319f);
  p = fmaf(p, t, -1.14955714f);
  const float q = __builtin_amdgcn_exp2f(fmaf(p, t, -1.f));   // Q(t)
  const float cdf = x >= 0.f ? 1.f - q : q;
  const float e = __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

// gelu(x) AND gelu'(x) of two elements from ONE evaluation of the tail fit (EPI_GELU_GP).  With q = Q(t), t = |x|:
//     gelu'(x) = Phi(x) + x phi(x) = 1 + r (x >= 0),  -r (x < 0),   r = t phi(t) - Q(t) = q m(t),
// m(t) = t phi(t) / Q(t) - 1 (the inverse Mills ratio times t, minus one: smooth, ~ t^2 for large t) fitted by a quartic
// against the FITTED q (minimax on the product q m: max |error| of gelu' 3.6e-5 over all x, tests/test_gelu_fit_cpu.py --
// the level of gelu_grad's two-transcendental form, two orders below the bf16 rounding of the stored derivative).
// Branch-free: gelu'(x) = 0.5 + copysign(0.5 + r, x).  x0, x1 are replaced by gelu(x) (bit-identical to gelu_tail2).
__device__ __forceinline__ void gelu_tail2_gp(float& x0, float& x1, float& g0, float& g1) {
  f32x2_t t, r;
  float a, b;
  const float nine = 9.f;
  asm("v_min_f32 %0, |%1|, %2" : "=v"(a) : "v"(x0), "s"(nine));
  asm("v_min_f32 %0, |%1|, %2" : "=v"(b) : "v"(x1), "s"(nine));
  t.x = a; t.y = b;
  asm("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(x0));
  asm("v_max_f32 %0, 0, %1" : "=v"(b) : "v"(x1));
  r.x = a; r.y = b;
  const f32x2_t c3 = {0.0041585f, 0.0041585f}, c2 = {-0.04571999f, -0.04571999f}, c1 = {-0.46495319f, -0.46495319f},
                c0 = {-1.14955714f, -1.14955714f}, m1 = {-1.f, -1.f};
  f32x2_t p = __builtin_elementwise_fma(t, c3, c2);
  p = __builtin_elementwise_fma(p, t, c1);
  p = __builtin_elementwise_fma(p, t, c0);
  p = __builtin_elementwise_fma(p, t, m1);
  f32x2_t q;
  q.x = __builtin_amdgcn_exp2f(p.x);
  q.y = __builtin_amdgcn_exp2f(p.y);
  const f32x2_t d4 = {-0.0117551f, -0.0117551f}, d3 = {0.09555683f, 0.09555683f}, d2 = {0.64461331f, 0.64461331f},
                d1 = {0.79644568f, 0.79644568f}, d0 = {-0.99992798f, -0.99992798f}, half = {0.5f, 0.5f};
  f32x2_t m = __builtin_elementwise_fma(t, d4, d3);
  m = __builtin_elementwise_fma(m, t, d2);
  m = __builtin_elementwise_fma(m, t, d1);
  m = __builtin_elementwise_fma(m, t, d0);
  const f32x2_t hr = __builtin_elementwise_fma(q, m, half);   // 0.5 + r
  g0 = 0.5f + __builtin_copysignf(hr.x, x0);
  g1 = 0.5f + __builtin_copysignf(hr.y, x1);
  r = __builtin_elementwise_fma(-t, q, r);
  x0 = r.x; x1 = r.y;
}

// ---- bf16 output tiles leave through LDS ---------------------------------------------------------------------
// In the accumulator layout a lane owns 4 consecutive R indices of ONE L row, so a direct store instruction touches
// 32 different output rows with 16 bytes each: 8192 partial-line write transactions per 256 x 256 tile.  Instead the
// converted tile is parked in dead operand slots as C[L row][R index] bf16 and written out 16 bytes per lane, every
// wave instruction covering whole 128-byte lines (QKV: 13.9 -> 11.5 ms per 12 layers, round 1).  The image holds
// TL / PASSES rows, so the tile leaves in PASSES passes of NT / PASSES MFMA column blocks.  Round 1 parked a pass
// cooperatively (a workgroup barrier, then every thread stored rows other waves had written); since round 2 every
// wave parks and stores ITS OWN 32 NTP x 32 MT part (put_w / store_w / store_w_part) with no barrier inside the
// passes -- the token-major outputs and the transposed V^T tiles of the QKV projection alike.
template <class T>
struct CTile {
  static constexpr int PASSES = (T::TL * T::TR * 2) / T::STAGE_BYTES;    // Tile256: 2, Tile128: 1
  static constexpr int NTP = T::NT / PASSES;                             // MFMA column blocks per pass
  static constexpr int ROWS = T::TL / PASSES;
  static_assert(PASSES >= 1 && PASSES * NTP == T::NT, "a pass is a whole number of MFMA column blocks");
  // The parked pass lives in two half regions (rows [0, HALF) and [HALF, ROWS)): one contiguous stage in the
  // two-stage kernels (hi = lo + HALF_BYTES), two separate dead operand slots in the R3 form.
  static constexpr int HALF = ROWS / 2, HALF_BYTES = HALF * T::TR * 2;
  struct Base { uint32_t lo, hi; };
  __device__ static __forceinline__ Base contiguous(const char* sC) {
    static_assert(2 * HALF_BYTES <= T::STAGE_BYTES, "a parked pass must fit the idle stage");
    return Base{lds_off(sC), lds_off(sC) + HALF_BYTES};
  }
  // Wave-local form: a wave parks ITS OWN part of the pass -- 32 NTP token rows x the 32 MT features it computed, 8 KB,
  // rows of MT * 64 bytes, 16-byte chunk index XOR row -- and writes it out itself, 64 / CW whole rows per instruction.
  // No workgroup barrier between a pass's arithmetic and its stores, so the two waves of a SIMD drift apart and one
  // computes while the other sits in its (~175-cycle-per-instruction) store issue; with the cooperative form both
  // waves reached the stores together and the VALU idled through them (1.4 + 1.1 k cycles of a 49 k-cycle FFN1 tile).
  static constexpr int CW = T::MT * 4;            // 16-byte chunks per wave row
  static constexpr int RW = NTP * 32;             // rows per wave and pass
  static constexpr int W_BYTES = RW * CW * 16;    // 8 KB
  static constexpr int W_STORES = RW * CW / 64;
  static_assert(W_BYTES * T::WAVES <= 2 * HALF_BYTES && (64 % CW) == 0, "wave-local park must fit the pass image");
  __device__ static __forceinline__ uint32_t wave_base(Base sC, int wave) {
    constexpr int PER_HALF = HALF_BYTES / W_BYTES;
    return (wave >= PER_HALF ? sC.hi + (wave - PER_HALF) * W_BYTES : sC.lo + wave * W_BYTES);
  }
  __device__ static __forceinline__ void put_w(uint32_t wb, const WavePos<T>& w, int mt, int ntl, int g, uint2 o) {
    const int row = ntl * 32 + w.li;
    lds_write_b64_hidden(wb + row * (CW * 16) + (((mt * 4 + g) ^ (row & (CW - 1))) << 4) + w.hi * 8, (u32x2_t){o.x, o.y});
  }
  // chunks [C0, C0 + CN) of every row only (CN * 16 bytes per row, 64 / CN rows per instruction)
  template <int C0, int CN>
  __device__ static __forceinline__ void store_w_part(bool nt, uint32_t wb, int lane, bf16_t* dst, int64_t ld, int64_t row_limit,
                                                      int64_t col_limit) {
    constexpr int RPI = 64 / CN, N_ST = RW / RPI;
    static_assert(64 % CN == 0 && N_ST % 4 == 0, "store_w_part geometry");
    const int r0 = lane / CN, c = C0 + (lane - r0 * CN);
    const int64_t nv = col_limit - c * 8;
#pragma unroll
    for (int i0 = 0; i0 < N_ST; i0 += 4) {
      u32x4_t v[4];
      auto at = [&](int i) {
        const int row = i * RPI + r0;
        return wb + row * (CW * 16) + ((c ^ (row & (CW - 1))) << 4);
      };
      lds_read4_b128_hidden(at(i0), at(i0 + 1), at(i0 + 2), at(i0 + 3), v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (i0 + j) * RPI + r0;
        bf16_t* pr = dst + (int64_t)row * ld + c * 8;
        if (row < row_limit && nv > 0) {
          if (nv >= 8) {   // (nt: wave-uniform, GemmArgs::nt_out)
            if (nt) store16<true>(pr, v[j]);
            else store16<false>(pr, v[j]);
          }
          else *(uint2*)pr = make_uint2(v[j].x, v[j].y);
        }
      }
    }
  }
  // dst -> element (row 0, column 0) of the WAVE's part of the pass; rows < row_limit and columns < col_limit are written
  __device__ static __forceinline__ void store_w(bool nt, uint32_t wb, int lane, bf16_t* dst, int64_t ld, int64_t row_limit,
                                                 int64_t col_limit) {
    constexpr int RPI = 64 / CW;
    const int r0 = lane / CW, c = lane - r0 * CW;
    const int64_t nv = col_limit - c * 8;
    static_assert(W_STORES % 4 == 0, "lds_read4_b128_hidden reads four chunks");
#pragma unroll
    for (int i0 = 0; i0 < W_STORES; i0 += 4) {
      u32x4_t v[4];
      auto at = [&](int i) {
        const int row = i * RPI + r0;
        return wb + row * (CW * 16) + ((c ^ (row & (CW - 1))) << 4);
      };
      lds_read4_b128_hidden(at(i0), at(i0 + 1), at(i0 + 2), at(i0 + 3), v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (i0 + j) * RPI + r0;
        bf16_t* pr = dst + (int64_t)row * ld + c * 8;
        if (row < row_limit && nv > 0) {
          if (nv >= 8) {   // (nt: wave-uniform, GemmArgs::nt_out)
            if (nt) store16<true>(pr, v[j]);
            else store16<false>(pr, v[j]);
          }
          else *(uint2*)pr = make_uint2(v[j].x, v[j].y);
        }
      }
    }
  }
};

// One workgroup walks a strided sequence of output tiles (persistent when the grid is smaller than the tile count:
// gemm_launch.hpp launches one workgroup per CU-slot).  While the epilogue of tile i runs, the first K chunk of tile
// i + 1 is already streaming into the idle operand stage, so neither the workgroup dispatch gap nor the first
// HBM/L2 round trip of a tile (~2-3 us of a ~20-30 us tile) is exposed.
template <int EPI, class T>
static __global__ void __launch_bounds__(T::THREADS, 2) k_gemm(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CT = CTile<T>;
  const WavePos<T> w;
  const int64_t ldw = a.ldw ? a.ldw : a.K, ldx = a.ldx ? a.ldx : a.K;
  int kbeg = 0, klen = a.K;
  if constexpr (EPI == EPI_SLAB_F32) {
    if (a.k_split_len) { kbeg = blockIdx.y * a.k_split_len; klen = a.k_split_len; }
  }
  // tile walk: XCD x (= blockIdx.x % 8) owns a contiguous chunk of the logical tile order -- consecutive logical
  // tiles sweep the feature tiles of one token tile, so the activation tile stays in that XCD's L2 -- and its
  // workgroups take the chunk's tiles round-robin (panel orders measured no better on any encoder shape)
  const uint32_t ntiles = (uint32_t)a.tilesN * a.tilesT;
  const uint32_t xcd = blockIdx.x & 7u, q8 = ntiles >> 3, r8 = ntiles & 7u;
  const uint32_t chunk_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const uint32_t chunk_len = q8 + (xcd < r8 ? 1u : 0u);
  const uint32_t stride = (gridDim.x + 7u) >> 3;
  uint32_t idx = blockIdx.x >> 3;
  if (idx >= chunk_len) return;
  if (a.clock_probe && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    a.clock_probe[0] = __builtin_amdgcn_s_memtime();
    a.clock_probe[1] = __builtin_amdgcn_s_memrealtime();
  }

  struct Coord { int64_t t0; int n0; int swap; };   // swap: V third of the fused QKV projection (tokens on registers)
  auto decode = [&](uint32_t i) {
    const uint32_t logical = chunk_base + i;
    int tt = logical / a.tilesN, tn = logical - tt * a.tilesN;
#ifdef CONVDR_GEMM_PANEL   // A/B builds only (make VARIANT=panelN EXTRA=-DCONVDR_GEMM_PANEL=N; profiles/r05_gemm_raster_ab.txt):
    // weight-set-aware raster -- an XCD sweeps its token tiles once per PANEL of CONVDR_GEMM_PANEL feature tiles, so that the
    // weights its 32 workgroups have live at any time are PANEL tiles (6 x 393 KB = 2.4 MB of FFN1's 4.7 MB) instead of all
    // of them, at the price of re-reading the XCD's activation tiles once per panel.  Shapes whose tile counts do not divide
    // keep the default order.
    if (a.tilesN % CONVDR_GEMM_PANEL == 0 && a.tilesN > CONVDR_GEMM_PANEL && (a.tilesT & 7) == 0) {
      const uint32_t nT = a.tilesT >> 3, per_panel = CONVDR_GEMM_PANEL * nT;      // token tiles of this XCD's chunk
      const uint32_t panel = i / per_panel, r = i - panel * per_panel;
      tt = xcd * nT + r / CONVDR_GEMM_PANEL;
      tn = panel * CONVDR_GEMM_PANEL + r % CONVDR_GEMM_PANEL;
    }
#endif
    if (a.dbg_same_tile) { tt = 0; tn = 0; }
    Coord c;
    c.t0 = (int64_t)tt * T::TL;      // the launcher lays tiles out as [tilesT][tilesN] with TR == TL
    c.n0 = tn * T::TR;
    c.swap = false;
    if constexpr (EPI == EPI_QKV) c.swap = c.n0 + a.third0 * a.H >= 2 * a.H;   // H % TR == 0 (checked by the launcher)
    return c;
  };
  auto tile_src = [&](const Coord& c) {
    return c.swap ? TileSrc<T>(a.X, ldx, a.rows, a.W, ldw, a.N, c.t0, c.n0, w)
                  : TileSrc<T>(a.W + kbeg, ldw, a.N, a.X + kbeg, ldx, a.rows, c.n0, c.t0, w);
  };

  // R3 (256 x 256 tiles): the 3 R-slot / 2 L-slot K step of gemm_nt.hpp.  All 160 KB are operand slots, so between two
  // tiles every byte has a second job: when a tile's main loop returns the slot state {rs, ls} of the next tile, R[rs]
  // and L[ls] take the next tile's chunks 0 at once, the first KB of R[rs + 1] holds the next tile's bias slice (its
  // chunk 1 is issued in step 0 of the next main loop, after every wave has folded the bias into its accumulators),
  // and the two slots the last step read -- R[rs + 2] and L[ls ^ 1], dead after the epilogue's first barrier -- are the
  // two halves of the parked output tile.
  constexpr bool R3 = T::WAVES == 8;   // (the two-stage K step serves the 4-wave 128 x 128 tiles)
  R3Slots st{0, 0};
  auto sbias_of = [&](R3Slots s) { return (float*)(smem + (s.rs == 2 ? 0 : s.rs + 1) * T::R_BYTES); };
  auto tile_src_all = [&](const Coord& cc) {
    return cc.swap ? TileSrcAll<T>(a.X, ldx, a.rows, a.W, ldw, a.N, cc.t0, cc.n0, w)
                   : TileSrcAll<T>(a.W + kbeg, ldw, a.N, a.X + kbeg, ldx, a.rows, cc.n0, cc.t0, w);
  };
  float* sbias = R3 ? sbias_of(st) : (float*)(smem + T::SMEM_BYTES);
  int trace_tile = 0;
#ifdef CONVDR_ENABLE_TRACE   // make TRACE=1: phase stamps for tools/gemm_trace.py
#define CONVDR_TRACE(ph)                                                                                       \
  if (a.trace && threadIdx.x == 0 && trace_tile < 64)                                                          \
    a.trace[((size_t)blockIdx.x * 64 + trace_tile) * 16 + (ph)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_TRACE(ph)
#endif
  // Bias handling is shaped by one compiler fact: while an LDS-DMA is in flight hipcc puts s_waitcnt vmcnt(0) in
  // front of the next LDS read or global-load use it can see, which would drain the next tile's prefetch at the top
  // of the epilogue.  So the epilogue of a bf16 tile touches LDS only through inline asm (CTile), and the bias never
  // appears there: a tile's bias slice is fetched during the PREVIOUS tile's main loop, parked in LDS right after it
  // (before the prefetch is issued), and folded into the accumulator initialisation at the tile's start -- where a
  // wait for chunk 0 is due anyway.  (V-third tiles have the feature on the lane: their bias sits in registers.)
  auto bias_slice = [&](const Coord& cc) {   // this thread's element of a tile's bias slice
    const int f = cc.n0 + (int)threadIdx.x;
    return (!cc.swap && a.bias && threadIdx.x < T::TR && f < a.N) ? a.bias[f] : 0.f;
  };
  Coord c = decode(idx);
  int buf = 0;
  bool landed = false;   // chunk 0 of the current tile has been waited for
  if (threadIdx.x < T::TR) sbias[threadIdx.x] = bias_slice(c);
  TileSrc<T> src = tile_src(c);
  TileSrcAll<T> src3 = tile_src_all(c);
  if constexpr (R3) gemm_r3_prologue<T>(src3, a.K, smem, w, st, false);
  else gemm_issue_stage<T>(src, 0, smem + buf * T::STAGE_BYTES, w);
  __syncthreads();
  for (;;) {
    const uint32_t next = idx + stride;
    const bool has_next = next < chunk_len;
    const Coord cn = has_next ? decode(next) : c;
    GemmAcc<T> acc;
    float bias_lane[T::NT];   // V third only
    if (c.swap) {
      acc.zero();
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt) {
        const int f = c.n0 + w.l_index(nt);  // feature on the lane
        bias_lane[nt] = a.bias[f < a.N ? f : a.N - 1];
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt) bias_lane[nt] = 0.f;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt) {
        u32x4_t bq[4];
        const uint32_t sb = lds_off(sbias);
        lds_read4_b128_hidden(sb + w.r_base(mt, 0) * 4, sb + w.r_base(mt, 1) * 4, sb + w.r_base(mt, 2) * 4,
                              sb + w.r_base(mt, 3) * 4, bq[0], bq[1], bq[2], bq[3]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int nt = 0; nt < T::NT; ++nt) {
            acc.c[mt][nt][4 * g + 0] = __uint_as_float(bq[g].x); acc.c[mt][nt][4 * g + 1] = __uint_as_float(bq[g].y);
            acc.c[mt][nt][4 * g + 2] = __uint_as_float(bq[g].z); acc.c[mt][nt][4 * g + 3] = __uint_as_float(bq[g].w);
          }
      }
    }
    float bias_next = has_next ? bias_slice(cn) : 0.f;
    // EPI_RESID_F32 on the 128^2 tiles (training forward at a few thousand rows: one wave of tiles, so a tile's latency
    // IS the kernel's): the bf16 residual fragments are fetched here, under the main loop, instead of in the epilogue
    // where each of the two column blocks exposed a full memory round trip (32 VGPRs; the 256^2 form has none to spare)
    constexpr bool RES_PRE = EPI == EPI_RESID_F32 && T::WAVES == 4;
    uint2 res_pre[RES_PRE ? T::NT : 1][T::MT][4];
    if constexpr (RES_PRE) {
      if (a.Rf == nullptr) {
#pragma unroll
        for (int nt = 0; nt < T::NT; ++nt) {
          const int64_t t = c.t0 + w.l_index(nt);
          const int64_t tc = t < a.rows ? t : a.rows - 1;
#pragma unroll
          for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              int f = c.n0 + w.r_base(mt, g);
              f = f < a.N ? f : a.N - 4;
              res_pre[nt][mt][g] = *(const uint2*)(a.R + tc * a.N + f);
            }
        }
      }
    }
    CONVDR_TRACE(0)
    int idle;   // the stage the last K step did not read
    if constexpr (R3) {
      st = gemm_nt_mainloop_r3<T>(src3, c.swap ? a.K : klen, smem, acc, w, st, true, true, landed || a.dbg_prelanded,
                                  (a.trace && trace_tile == 8) ? a.trace + ((size_t)blockIdx.x * 64 + 56) * 16 : nullptr);
      idle = 0;
      sbias = sbias_of(st);   // free: neither read by the last step nor a prologue target
    } else {
    idle = gemm_nt_mainloop<T>(src, c.swap ? a.K : klen, smem, acc, w, buf, true, landed,
                               (a.trace && trace_tile == 8) ? a.trace + ((size_t)blockIdx.x * 64 + 56) * 16 : nullptr);
    }
    landed = false;
    // EPI_MUL_GP: the tile's gelu' image (128 KB from HBM, written by the forward milliseconds ago) is fetched HERE, ahead of
    // the next tile's prologue in the vector-memory queue and with the fragment registers of the main loop free to hold it
    // (64 VGPRs for a 256 x 256 tile): its round trip passes under the prologue issue and the epilogue's first barrier.
    // Fetched inside the epilogue, one 32-token block at a time, the same loads left the matrix pipe idle for two HBM round
    // trips per tile: the dgrad GEMM gained exactly the 46 us per layer the separate gelu' pass had cost (step unchanged).
    constexpr bool GP_PRE = EPI == EPI_MUL_GP;
    u32x4_t gp_pre[GP_PRE ? T::NT : 1][GP_PRE ? T::MT : 1][2];
    if constexpr (GP_PRE) {
      int tid_p = threadIdx.x;
      asm volatile("" : "+v"(tid_p));
      const WavePos<T> wp(tid_p);
      const int64_t rows32p = (a.rows + 31) & ~(int64_t)31;
      // (every load is UNCONDITIONAL -- blocks past the end of the matrix read the image's first block, their products are
      //  never stored: behind a lane condition hipcc put each load in its own branch with s_waitcnt vmcnt(0) at the join,
      //  sixteen serialised HBM round trips per tile)
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt) {
        const int64_t tb = c.t0 + (wp.wl * T::NT + nt) * 32;
        const int64_t tbc = tb < rows32p ? tb : 0;
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt) {
          const int f0 = c.n0 + wp.wr * T::MT * 32 + mt * 32;
          const int f0c = f0 < a.N ? f0 : 0;
          const bf16_t* blk = a.Gp + ((tbc >> 5) * (a.N >> 3) + (f0c >> 3)) * 256 + wp.li * 8;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            gp_pre[nt][mt][j] = __builtin_nontemporal_load((const u32x4_t*)(blk + (int64_t)(2 * j + wp.hi) * 256));
        }
      }
    }
    // hipcc does not see the main loop's inline-asm waits: make it retire the bias loads HERE (a no-op wait, nothing
    // is in flight), not at their first use further down
    asm volatile("" : "+v"(bias_next));
    if constexpr (RES_PRE) {
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(res_pre[nt][mt][g].x), "+v"(res_pre[nt][mt][g].y));
    }
#pragma unroll
    for (int i = 0; i < T::NT; ++i) asm volatile("" : "+v"(bias_lane[i]));
    if (threadIdx.x < T::TR) sbias[threadIdx.x] = bias_next;   // (every wave has read this tile's slice before its
    CONVDR_TRACE(1)                                             //  first main-loop barrier; nothing is in flight here)
    if (has_next) {
      if constexpr (R3) {
        src3 = tile_src_all(cn);
        gemm_r3_prologue<T>(src3, a.K, smem, w, st, false);
      } else {
        src = tile_src(cn);
        gemm_issue_stage<T>(src, 0, smem + idle * T::STAGE_BYTES, w);
      }
    }
    // epilogue scratch: the stage (R3: the two slots) the last K step read
    // (tiles with a 128-row L operand: an L slot is smaller than half a parked pass, and the 32 KB of the 160 that the
    //  operand slots leave over take its place)
    typename CT::Base sC;
    if constexpr (R3) {
      static_assert(T::R_BYTES >= CT::HALF_BYTES && 3 * T::R_BYTES + 2 * T::L_BYTES + (T::L_BYTES >= CT::HALF_BYTES ? 0 : CT::HALF_BYTES) <= 160 * 1024,
                    "parked pass: one dead R slot + one dead L slot (or the spare LDS behind the slots)");
      sC = typename CT::Base{lds_off(smem + (st.rs == 0 ? 2 : st.rs - 1) * T::R_BYTES),
                             T::L_BYTES >= CT::HALF_BYTES ? lds_off(smem + 3 * T::R_BYTES + (st.ls ^ 1) * T::L_BYTES)
                                                          : lds_off(smem + 3 * T::R_BYTES + 2 * T::L_BYTES)};
    } else {
      sC = CT::contiguous(smem + (idle ^ 1) * T::STAGE_BYTES);
    }
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));   // opaque: keeps the epilogue's lane-dependent addresses out of the main loop's registers
    const WavePos<T> we(tid_e);
    const int64_t t0 = c.t0;
    const int n0 = c.n0;

    if (a.dbg_skip_epi == 1) {
      float s = 0.f;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < T::NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += acc.c[mt][nt][r];
      if (s == 12345.678f) a.Cb[0] = f32_to_bf16(s);
    } else if (c.swap) {
      // ---- V third -> Vt[feature][token]: features on lanes, tokens on registers; wave-local park like the
      // token-major outputs (a wave's part: 32 NTP features x its 32 MT tokens) ----
      lds_barrier();   // every wave is done with the last K step's stage
#pragma unroll
      for (int pass = 0; pass < CT::PASSES; ++pass) {
        const uint32_t wb = CT::wave_base(sC, we.wave);
#pragma unroll
        for (int ntl = 0; ntl < CT::NTP; ++ntl) {
          const int nt = pass * CT::NTP + ntl;
          const float bv = bias_lane[nt];
#pragma unroll
          for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x16& v = acc.c[mt][nt];
              uint2 o;
              o.x = pack_bf16x2(v[4 * g + 0] + bv, v[4 * g + 1] + bv);
              o.y = pack_bf16x2(v[4 * g + 2] + bv, v[4 * g + 3] + bv);
              CT::put_w(wb, we, mt, ntl, g, o);
            }
        }
        if (pass == CT::PASSES - 1 && has_next) {   // the prefetch has had the whole epilogue to land: retire it
          lds_dma_wait_all();                        // BEFORE the last stores enter the (in-order) queue
          landed = true;
        }
        const int row0 = (we.wl * T::NT + pass * CT::NTP) * 32, col0 = we.wr * T::MT * 32;   // feature / token offsets
        CT::store_w(a.nt_out != 0, wb, we.lane, a.Vt + (int64_t)(n0 + (a.third0 - 2) * a.H + row0) * a.ldt + t0 + col0, a.ldt,
                    a.N - n0 - row0, a.rows - t0 - col0);
      }
    } else if (EPI == EPI_GELU_BLK || (EPI == EPI_QKV && a.qk_blocked)) {
      // ---- blocked bf16 output straight from the registers (see EPI_GELU_BLK; EPI_QKV: the Q and K thirds) ----
      lds_barrier();     // (the next tile's sbias is visible; nothing of this epilogue touches LDS)
      CONVDR_TRACE(2)
      const int64_t rows32 = (a.rows + 31) & ~(int64_t)31;
      bf16_t* dstm = a.Cb;        // destination matrix, its width in octets, the tile's first feature inside it
      int oct_w = a.N >> 3, f0 = n0;
      if constexpr (EPI == EPI_QKV) {
        const int na = n0 + a.third0 * a.H;
        dstm = na < a.H ? a.Qo : a.Ko;
        f0 = na < a.H ? na : na - a.H;
        oct_w = a.H >> 3;
      }
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt) {
        const int64_t tb = t0 + (we.wl * T::NT + nt) * 32;   // first token of this 32-token block (wave-uniform)
        if (nt == T::NT - 1 && has_next) {   // the prefetch is retired BEFORE the last stores enter the (in-order) queue
          lds_dma_wait_all();
          landed = true;
        }
        bf16_t* blk = dstm + ((tb >> 5) * oct_w + ((f0 + we.wr * T::MT * 32) >> 3)) * 256 + we.li * 8;
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            uint2 o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int g = 2 * j + e;
              const f32x16& v = acc.c[mt][nt];
              float y0 = v[4 * g + 0], y1 = v[4 * g + 1], y2 = v[4 * g + 2], y3 = v[4 * g + 3];   // bias included
              if constexpr (EPI == EPI_GELU_BLK) { gelu_tail2(y0, y1); gelu_tail2(y2, y3); }
              o[e].x = pack_bf16x2(y0, y1);
              o[e].y = pack_bf16x2(y2, y3);
            }
            // lanes (li, 0) / (li, 1) hold features +0..3 / +4..7 of octets 2j (o[0]) and 2j + 1 (o[1]) of token li:
            // after the swaps lane (li, 0) owns octet 2j, lane (li, 1) octet 2j + 1
            const auto sx = __builtin_amdgcn_permlane32_swap(o[0].x, o[1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(o[0].y, o[1].y, false, false);
            u32x4_t q;
            q.x = sx[0]; q.y = sy[0]; q.z = sx[1]; q.w = sy[1];
            if (tb < rows32 && n0 + we.wr * T::MT * 32 + mt * 32 < a.N)
              store16<NT_GEMM_BLK>(blk + (int64_t)(mt * 4 + 2 * j + we.hi) * 256, q);
          }
      }
      CONVDR_TRACE(6)
    } else {
      // ---- epilogue: the tile's bias slice is parked in LDS behind the stages (no vmcnt round trip per register
      // quad), all residual loads of a 32-token column block are issued up front ----
      lds_barrier();     // every wave is done with the last K step's stage (and the next tile's sbias is visible)
      CONVDR_TRACE(2)
      const bool full_n = n0 + T::TR <= a.N;  // workgroup-uniform: no per-quad feature bound checks on the fast path
      constexpr bool BF16_OUT = EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV || EPI == EPI_GELU_SAVE ||
                                EPI == EPI_GELU_GP || EPI == EPI_MUL_GP;
      constexpr bool GP = EPI == EPI_GELU_GP || EPI == EPI_MUL_GP;   // a blocked gelu' image is written / read beside the tile
      const int64_t rows32 = (a.rows + 31) & ~(int64_t)31;
      constexpr int NOUT = EPI == EPI_GELU_SAVE ? 2 : 1;   // second output: the pre-activation
#pragma unroll
      for (int out = 0; out < NOUT; ++out)
#pragma unroll
        for (int pass = 0; pass < CT::PASSES; ++pass) {
          const uint32_t wb = CT::wave_base(sC, we.wave);   // (wave-local park: a wave re-reads only what it wrote)
          // wide wave rows: the two 128-byte halves of a pass row are stored as soon as each is parked (FFN1 -0.8 %)
          constexpr bool HALVES = BF16_OUT && T::MT >= 4 && CT::NTP == 1;
          bf16_t* wdst = nullptr;
          int64_t wld = 0, wrows = 0, wcols = 0;
          if constexpr (HALVES) {
            bf16_t* dst;
            int64_t cols;
            if constexpr (EPI == EPI_QKV) {
              const int na = n0 + a.third0 * a.H;
              dst = na < a.H ? a.Qo + t0 * a.H + na : a.Ko + t0 * a.H + (na - a.H);
              wld = a.H; cols = T::TR;
            } else {
              dst = (out ? a.Cb2 : a.Cb) + t0 * a.N + n0;
              wld = a.N; cols = a.N - n0;
            }
            const int row0 = (we.wl * T::NT + pass * CT::NTP) * 32, col0 = we.wr * T::MT * 32;
            wdst = dst + (int64_t)row0 * wld + col0; wrows = a.rows - t0 - row0; wcols = cols - col0;
          }
#pragma unroll
          for (int ntl = 0; ntl < CT::NTP; ++ntl) {
            const int nt = pass * CT::NTP + ntl;
            const int64_t t = t0 + we.l_index(nt);  // token on the lane
            const bool t_ok = t < a.rows;
            const int64_t tc = t_ok ? t : a.rows - 1;
            uint2 res[T::MT][4];
            // blocked gelu' image: octet (mt * 4 + 2 j + hi) of this lane's token (see EPI_GELU_BLK for the layout)
            u32x4_t gpq[GP ? T::MT : 1][2];
            bf16_t* gp_blk = nullptr;
            bool gp_ok = false;
            if constexpr (GP) {
              const int64_t tb = t0 + (we.wl * T::NT + nt) * 32;   // first token of this 32-token block (wave-uniform)
              bf16_t* img = EPI == EPI_GELU_GP ? a.Cb2 : const_cast<bf16_t*>(a.Gp);
              gp_blk = img + ((tb >> 5) * (a.N >> 3) + ((n0 + we.wr * T::MT * 32) >> 3)) * 256 + we.li * 8;
              gp_ok = tb < rows32;
              if constexpr (EPI == EPI_MUL_GP) {   // (fetched right after the main loop: gp_pre)
#pragma unroll
                for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
                  for (int j = 0; j < 2; ++j) gpq[mt][j] = gp_pre[nt][mt][j];
              }
            }
            if constexpr (EPI == EPI_RESID_F32) {
              if (a.Rf == nullptr) {
#pragma unroll
                for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
                  for (int g = 0; g < 4; ++g) {
                    if constexpr (RES_PRE) {
                      res[mt][g] = res_pre[nt][mt][g];
                    } else {
                      int f = n0 + we.r_base(mt, g);
                      f = (full_n || f < a.N) ? f : a.N - 4;
                      res[mt][g] = *(const uint2*)(a.R + tc * a.N + f);
                    }
                  }
              }
            }
            uint2 gpq_o0 = make_uint2(0u, 0u), gpq_o1 = make_uint2(0u, 0u);
#pragma unroll
            for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int fl = we.r_base(mt, g);      // 4 consecutive features (N % 4 == 0)
                const int f = n0 + fl;
                const f32x16& v = acc.c[mt][nt];
                float y0 = v[4 * g + 0], y1 = v[4 * g + 1], y2 = v[4 * g + 2], y3 = v[4 * g + 3];   // bias included
                if constexpr (EPI == EPI_GELU_BF16 || EPI == EPI_GELU_SAVE) {
                  if (out == 0) {
                    gelu_tail2(y0, y1); gelu_tail2(y2, y3);
                  }
                }
                if constexpr (EPI == EPI_GELU_GP) {
                  float d0, d1, d2, d3;
                  gelu_tail2_gp(y0, y1, d0, d1); gelu_tail2_gp(y2, y3, d2, d3);
                  // quads g = 2 j (features +0..3 / +4..7 of octet 2 j for hi = 0 / 1) and g = 2 j + 1 of one token: after the
                  // swaps lane (li, 0) owns octet 2 j, lane (li, 1) octet 2 j + 1 -- 512 contiguous bytes per half-wave
                  uint2& dq = (g & 1) ? gpq_o1 : gpq_o0;
                  dq.x = pack_bf16x2(d0, d1);
                  dq.y = pack_bf16x2(d2, d3);
                  if (g & 1) {
                    const auto sx = __builtin_amdgcn_permlane32_swap(gpq_o0.x, gpq_o1.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(gpq_o0.y, gpq_o1.y, false, false);
                    u32x4_t q4;
                    q4.x = sx[0]; q4.y = sy[0]; q4.z = sx[1]; q4.w = sy[1];
                    if (gp_ok && (full_n || f < a.N))
                      store16<true>(gp_blk + (int64_t)(mt * 4 + (g & 2) + we.hi) * 256, q4);
                  }
                }
                if constexpr (EPI == EPI_MUL_GP) {
                  // the inverse of the swaps above hands each lane its two quads of the octet pair back
                  if (!(g & 1)) {
                    const u32x4_t q4 = gpq[mt][g >> 1];
                    const auto sx = __builtin_amdgcn_permlane32_swap(q4.x, q4.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(q4.y, q4.w, false, false);
                    gpq_o0 = make_uint2(sx[0], sy[0]);
                    gpq_o1 = make_uint2(sx[1], sy[1]);
                  }
                  const uint2 dq = (g & 1) ? gpq_o1 : gpq_o0;
                  y0 *= __uint_as_float(dq.x << 16); y1 *= __uint_as_float(dq.x & 0xffff0000u);
                  y2 *= __uint_as_float(dq.y << 16); y3 *= __uint_as_float(dq.y & 0xffff0000u);
                }
                if constexpr (EPI == EPI_RESID_F32) {
                  if (a.drop.thresh) {   // workgroup-uniform: hidden dropout of the training forward (BertSelfOutput / BertOutput)
                    float m0, m1, m2, m3;
                    drop_hidden4(a.drop, t, f, a.N, m0, m1, m2, m3);
                    y0 *= m0; y1 *= m1; y2 *= m2; y3 *= m3;
                  }
                  if (a.Rf) {
                    const int fc = (full_n || f < a.N) ? f : a.N - 4;
                    const float4 r = *(const float4*)(a.Rf + tc * a.N + fc);
                    y0 += r.x; y1 += r.y; y2 += r.z; y3 += r.w;
                  } else {
                    const uint2 r = res[mt][g];
                    y0 += __uint_as_float(r.x << 16); y1 += __uint_as_float(r.x & 0xffff0000u);
                    y2 += __uint_as_float(r.y << 16); y3 += __uint_as_float(r.y & 0xffff0000u);
                  }
                }
                if constexpr (BF16_OUT) {
                  uint2 o;
                  o.x = pack_bf16x2(y0, y1);
                  o.y = pack_bf16x2(y2, y3);
                  CT::put_w(wb, we, mt, ntl, g, o);
                  if constexpr (HALVES) {
                    if (mt == T::MT / 2 - 1 && g == 3) {   // the first half of the row is parked: send it off
                      if (out == NOUT - 1 && pass == CT::PASSES - 1 && has_next) { lds_dma_wait_all(); landed = true; }
                      CT::template store_w_part<0, CT::CW / 2>(a.nt_out != 0, wb, we.lane, wdst, wld, wrows, wcols);
                    }
                  }
                } else if (t_ok && (full_n || f < a.N)) {
                  if constexpr (EPI == EPI_SLAB_F32) {
                    *(float4*)(a.Cf + ((int64_t)blockIdx.y * a.rows + t) * a.N + f) = make_float4(y0, y1, y2, y3);
                  } else {
                    *(float4*)(a.Cf + t * a.N + f) = make_float4(y0, y1, y2, y3);
                  }
                }
              }
          }
          CONVDR_TRACE(3 + 4 * pass)
          if constexpr (BF16_OUT) {
            CONVDR_TRACE(4 + 4 * pass)
            if (out == NOUT - 1 && pass == CT::PASSES - 1 && has_next) {   // see the V third above
              lds_dma_wait_all();
              landed = true;
            }
            CONVDR_TRACE(5 + 4 * pass)
            if (a.dbg_skip_epi != 3) {
              bf16_t* dst;
              int64_t ldo, cols;
              if constexpr (EPI == EPI_QKV) {   // the tile lies inside the Q or the K third (H % TR == 0)
                const int na = n0 + a.third0 * a.H;
                dst = na < a.H ? a.Qo + t0 * a.H + na : a.Ko + t0 * a.H + (na - a.H);
                ldo = a.H; cols = T::TR;
              } else {
                dst = a.dbg_skip_epi == 4 ? a.Cb : (out ? a.Cb2 : a.Cb) + t0 * a.N + n0;
                ldo = a.N; cols = a.N - n0;
              }
              if constexpr (HALVES) {
                CT::template store_w_part<CT::CW / 2, CT::CW / 2>(a.nt_out != 0, wb, we.lane, wdst, wld, wrows, wcols);
              } else {
                const int row0 = (we.wl * T::NT + pass * CT::NTP) * 32, col0 = we.wr * T::MT * 32;
                CT::store_w(a.nt_out != 0, wb, we.lane, dst + (int64_t)row0 * ldo + col0, ldo, a.rows - t0 - row0, cols - col0);
              }
            }
            CONVDR_TRACE(6 + 4 * pass)
          }
        }
    }
    ++trace_tile;
    if (!has_next) break;
    idx = next;
    c = cn;
    buf = idle;
  }
  if (a.clock_probe && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    a.clock_probe[2] = __builtin_amdgcn_s_memtime();
    a.clock_probe[3] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------------------------
// Self-attention, head_dim 64, varlen (one sequence per blockIdx.z, no padding keys exist).
// Workgroup = 4 waves = 128 queries of one (sequence, head); K and V^T tiles of 64 keys staged in LDS by
// LDS-DMA and shared by the 4 waves; each wave owns 32 queries.  Everything per query is LANE-LOCAL:
//   S^T = K Q^T   (A = K rows, B = Q)  -> lane = query, registers = keys
//   O^T = V^T P^T (A = V^T rows, B = P straight from the S^T registers) -> lane = query, registers = head dims
// The K rows feeding MFMA row i are permuted (bits 2 and 3 of i swapped) so that the 8 S^T registers
// 8s..8s+7 of a lane hold exactly the keys 16 s + 8 (lane >> 5) + 0..7 that MFMA expects as the lane's
// B-operand k-slots in the P.V step: no cross-lane shuffles, no LDS round trip for P.
// Online softmax over key tiles; lanes q and q+32 hold the two halves of a query's keys / head dims and
// exchange only the running max and the final row sum.  Writes LSE (natural log) when lse != nullptr.
// ---------------------------------------------------------------------------------------------
constexpr int ATT_KV_AUX = 2;   // cache policy of the K / V^T tile DMA of k_attention_fwd (nt: read by this workgroup only; 3.61 -> 3.29 ms per 12 layers)
struct AttnArgs {
  const bf16_t *Q, *K, *Vt;
  int64_t ldt;
  const int32_t *cu, *lens;
  int H;
  int64_t ldq;      // row stride of Q and K in elements (H, or 3H when they live in a fused [rows, 3H] buffer)
  bf16_t* ctx;
  float* lse;       // [heads, ldt] or nullptr
  float scale;      // 1 / sqrt(head_dim)
  unsigned long long* trace;   // TRACE builds: [workgroup][8] s_memtime stamps of thread 0 (or null)
};
#ifdef CONVDR_ENABLE_TRACE
#define CONVDR_ATT_TRACE(i)                                                                                  \
  if (a.trace && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x == 0 && blockIdx.z < 2048)                \
    a.trace[blockIdx.z * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_ATT_TRACE(i)
#endif

constexpr int ATT_TILE_PAIR = 2 * 64 * 128;   // K tile + V^T tile, 8 KB each
constexpr int ATT_SMEM_BYTES = 2 * ATT_TILE_PAIR;  // double buffered

// A wave's 32 x 64 output tile (MFMA layout: lane (li, hi) holds 8-byte pieces of row li) -> global rows in whole
// 128-byte lines.  Stored straight from the registers a wave instruction touches 32 rows with 16 bytes each (256
// partial-line operations per wave against 32 for its K / V^T tiles); here the wave parks the tile in 4 KB of dead LDS
// (16-byte chunk index XOR row & 7) and each store instruction writes 8 whole rows.  `so`: the wave's 4 KB (every wave
// of the workgroup must be done with whatever lived there); dst: element (row 0, column 0) of the tile; rows < nvalid
// are written.  Attention forward 4.08 -> 3.86 ms per 12 layers.
__device__ __forceinline__ void attn_park_store(char* so, const f32x16 (&o)[2], float scale, int lane, bf16_t* dst,
                                                int64_t pitch, int nvalid) {
  const int li = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 ov;
      ov.x = pack_bf16x2(o[dt][4 * g + 0] * scale, o[dt][4 * g + 1] * scale);
      ov.y = pack_bf16x2(o[dt][4 * g + 2] * scale, o[dt][4 * g + 3] * scale);
      *(uint2*)(so + li * 128 + (((dt * 4 + g) ^ (li & 7)) << 4) + hi * 8) = ov;
    }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave's LDS operations execute in order; this pins the compiler's)
  const int c8 = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 3) + 8 * i;
    const uint4 v = *(const uint4*)(so + row * 128 + ((c8 ^ (row & 7)) << 4));
    if (row < nvalid) *(uint4*)(dst + row * pitch + c8 * 8) = v;
  }
  __builtin_amdgcn_wave_barrier();   // (the region may be parked into again)
}

// CLS_Q (last layer of an inference pass): only the CLS row of every sequence is needed downstream.  Q is then a
// [B, H] matrix of CLS queries (row b), the workgroup still streams the sequence's K / V^T tiles, wave 0 alone does the
// arithmetic (all of its 32 query columns carry the same query) and one lane pair stores ctx[b] ([B, H]).
// CTX_BLK: the output goes to the blocked layout [rows / 32][H / 8][32 tokens][8 dims] (hm_blocked_offset) that the
// row-complete output projection stages with whole-line LDS-DMA: lanes (query, hi = 0 / 1) hold the two halves of a dim
// octet, one v_permlane32_swap per dword gives each lane 16 bytes, and runs of 8 tokens are whole 128-byte lines
// (sequences start at multiples of 8 rows) -- no LDS park, no barrier before the stores.
// QK_BLK: Q and K arrive in the same blocked layout (written by the QKV projection's blocked epilogue): a K tile's
// LDS-DMA pieces are runs of 8 tokens x 16 bytes = whole lines as before, and the Q fragment loads -- four 16-byte
// loads per lane that touched 32 rows x 32 bytes per instruction in the row-major form (128 of a wave's 224 line
// operations) -- read 512 contiguous bytes per half-wave.  (CLS_Q keeps its [B, H] row-major query matrix.)
template <bool CLS_Q, bool CTX_BLK = false, bool QK_BLK = false>
static __global__ void __launch_bounds__(256, 4) k_attention_fwd(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.z, h = blockIdx.y;
  const int len = a.lens[b];
  const int q0 = blockIdx.x * 128;
  if (q0 >= len) return;
  CONVDR_ATT_TRACE(0)
  const int64_t base = a.cu[b];
  const int plen = a.cu[b + 1] - (int)base;  // len rounded up to the row alignment (fetched with the other scalars: a
                                             // scalar load issued in the epilogue costs a full ~4 k-cycle round trip)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int q = CLS_Q ? 0 : q0 + wave * 32 + li;
  const int qc = q < len ? q : len - 1;
  const int H = a.H;
  CONVDR_ATT_TRACE(1)

  bf16x8 qf[4];
  if constexpr (QK_BLK && !CLS_Q) {
    const int64_t t = base + qc;
    const bf16_t* qp = a.Q + ((t >> 5) * (a.H >> 3) + h * 8 + hi) * 256 + (t & 31) * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // dims 16 s + 8 hi .. + 7 = octet 2 s + hi
      qf[s] = __builtin_nontemporal_load((const bf16x8*)(qp + 2 * s * 256));   // (read by this workgroup only)
    }
  } else {
    const bf16_t* qp = CLS_Q ? a.Q + (int64_t)b * a.ldq + h * 64 + 8 * hi : a.Q + (base + qc) * a.ldq + h * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
  }
  const float c = a.scale * 1.44269504088896341f;  // exp(x * scale) = exp2(x * c)
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

  const int sw = (lane >> 1) & 7;
  const int krow = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);  // bits 2 <-> 3
  const int ksw = (krow >> 1) & 7;

  // K / V^T tiles are double buffered: the LDS-DMA of tile t+1 flies under the MFMAs of tile t
  auto stage_tile = [&](int kv0, int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // 64 rows x 128 B per tile = 2 LDS-DMA rounds of 256 lanes x 16 B
      const int r0 = (i * 4 + wave) * 8;
      const int row = r0 + (lane >> 3);
      const int gch = (lane & 7) ^ ((row >> 1) & 7);
      if constexpr (QK_BLK) {
        const int64_t t = base + kv0 + row;
        glds16_aux<ATT_KV_AUX>((const char*)(a.K + ((t >> 5) * (a.H >> 3) + h * 8 + gch) * 256 + (t & 31) * 8), smem + buf * ATT_TILE_PAIR + r0 * 128);
      } else {
        glds16_aux<ATT_KV_AUX>((const char*)(a.K + (base + kv0 + row) * a.ldq + h * 64) + gch * 16, smem + buf * ATT_TILE_PAIR + r0 * 128);
      }
      glds16_aux<ATT_KV_AUX>((const char*)(a.Vt + (int64_t)(h * 64 + row) * a.ldt + base + kv0) + gch * 16,
             smem + buf * ATT_TILE_PAIR + 64 * 128 + r0 * 128);
    }
  };
  // Both buffers are filled up front: the kernel is bound by memory latency x bytes in flight (a workgroup's whole
  // working set is 64 KB and its arithmetic ~1 us), so the second tile's round trip must not start after the first's
  // has completed -- for the 128-token passages of the corpus that is the entire K/V of the head.
  stage_tile(0, 0);
  if (len > 64) stage_tile(64, 1);
  for (int kv0 = 0, it = 0; kv0 < len; kv0 += 64, ++it) {
    const int buf = it & 1;
    lds_dma_wait_all();  // explicit: hipcc's automatic vmcnt wait for LDS-DMA is not reliable (gemm_nt.hpp)
    __syncthreads();     // tile `it` landed for everyone; everyone finished reading tile it-1 (the other buffer)
    if (it >= 1 && kv0 + 64 < len) stage_tile(kv0 + 64, buf ^ 1);
    if (it == 0) { CONVDR_ATT_TRACE(2) }
    if (it == 1) { CONVDR_ATT_TRACE(3) }
    if (CLS_Q && wave != 0) continue;   // (has staged its share and passed the barrier)
    const char* sK = smem + buf * ATT_TILE_PAIR;
    const char* sV = sK + 64 * 128;

    // ---- S^T = K Q^T for the 64 keys of this tile ----
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
      const char* kp = sK + (kt * 32 + krow) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *(const bf16x8*)(kp + (((2 * s + hi) ^ ksw) * 16));
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
      }
    }
    // register r of tile kt <-> key kv0 + 32 kt + 16 (r >> 3) + 8 hi + (r & 7)
    float mx = -INFINITY;
    if (kv0 + 64 > len) {  // ragged last tile only (workgroup-uniform): mask the keys past the sequence
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= len) st[kt][r] = -INFINITY;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);               // finite: every tile has >= 1 valid key
    const float mnc = mn * c;
    const float alpha = __builtin_amdgcn_exp2f(m * c - mnc);   // m = -inf on the first tile -> 0
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], c, -mnc));   // one FMA + one v_exp_f32 per score
        st[kt][r] = p;
        ps += p;
      }
    l = l * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {  // 16 keys per step
      const int kt = s4 >> 1, r0 = (s4 & 1) * 8;
      union { bf16x8 v; uint32_t u[4]; } pb;
#pragma unroll
      for (int j = 0; j < 4; ++j) pb.u[j] = pack_bf16x2(st[kt][r0 + 2 * j], st[kt][r0 + 2 * j + 1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 vf = *(const bf16x8*)(sV + (dt * 32 + li) * 128 + (((2 * s4 + hi) ^ sw) * 16));
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb.v, o[dt], 0, 0, 0);
      }
    }
  }

  CONVDR_ATT_TRACE(4)
  l += __shfl_xor(l, 32, 64);
  if constexpr (CLS_Q) {
    if (wave == 0 && li == 0) {
      const float inv = 1.f / l;
      bf16_t* dst = a.ctx + (int64_t)b * H + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 ov;
          ov.x = pack_bf16x2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
          ov.y = pack_bf16x2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
          *(uint2*)(dst + dt * 32 + 8 * g + 4 * hi) = ov;
        }
    }
    return;
  }
  if constexpr (CTX_BLK) {
    // (lane-dependent addresses from an opaque copy of the thread index: otherwise hipcc computes them ahead of the key
    //  loop and spills them across it -- 7 VGPR spills at this kernel's 128-register budget)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int q_e = q0 + (tid_e >> 6) * 32 + (tid_e & 31), hi_e = (tid_e >> 5) & 1;
    const float inv = q_e < len ? 1.f / l : 0.f;     // alignment rows [len, plen) get zeros
    const int64_t t = base + q_e;
    bf16_t* blk = a.ctx + ((t >> 5) * (H >> 3) + h * 8) * 256 + (t & 31) * 8;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t x0 = pack_bf16x2(o[dt][8 * j + 0] * inv, o[dt][8 * j + 1] * inv);
        const uint32_t y0 = pack_bf16x2(o[dt][8 * j + 2] * inv, o[dt][8 * j + 3] * inv);
        const uint32_t x1 = pack_bf16x2(o[dt][8 * j + 4] * inv, o[dt][8 * j + 5] * inv);
        const uint32_t y1 = pack_bf16x2(o[dt][8 * j + 6] * inv, o[dt][8 * j + 7] * inv);
        const auto sx = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(y0, y1, false, false);
        u32x4_t v;
        v.x = sx[0]; v.y = sy[0]; v.z = sx[1]; v.w = sy[1];
        if (q_e < plen) store16<NT_ATT>(blk + (dt * 4 + 2 * j + hi_e) * 256, v);
      }
    if (q_e < plen && a.lse && hi_e == 0) a.lse[(int64_t)h * a.ldt + base + q_e] = q_e < len ? m * a.scale + logf(l) : 0.f;
    CONVDR_ATT_TRACE(5)
    return;
  }
  __syncthreads();   // every wave is done with the last K / V^T tile: its buffers take the output tiles (attn_park_store)
  {
    // alignment rows [len, plen) get zeros: they feed later GEMMs / V^T columns and must stay finite
    const float inv = q < len ? 1.f / l : 0.f;
    const int r0 = q0 + wave * 32;
    attn_park_store(smem + wave * 4096, o, inv, lane, a.ctx + (base + r0) * H + h * 64, H, plen - r0);
    if (q < plen && a.lse && hi == 0) a.lse[(int64_t)h * a.ldt + base + q] = q < len ? m * a.scale + logf(l) : 0.f;
  }
  CONVDR_ATT_TRACE(5)
}

// fp32 rows -> bf16 rows (weight packing at load / after each optimizer step)
static __global__ void __launch_bounds__(256) k_cast_f32_bf16(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = *(const float4*)(x + 4 * i);
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    *(uint2*)(y + 4 * i) = o;
  }
}

}  // namespace convdr
