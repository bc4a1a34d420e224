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
                                                  int max_pos, int32_t* __restrict__ tok_id,
                                                  int32_t* __restrict__ tok_pos) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int base = cu[b], end = cu[b + 1];
  const int64_t* ids = (const int64_t*)ids_v;
  const int32_t* ids_i = (const int32_t*)ids_v;
  const int len_b = lens[b];
  int kept = 0, nonpad = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    const bool valid = l < L;
    const int64_t id = valid ? (ids32 ? (int64_t)ids_i[(int64_t)b * L + l] : ids[(int64_t)b * L + l]) : (int64_t)pad_idx;
    const bool m = valid && (mask ? mask[(int64_t)b * L + l] != 0 : l < len_b);
    const bool np = valid && id != pad_idx;
    const unsigned long long bm = __ballot(m), bnp = __ballot(np);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (m) {
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
                                                  const float* __restrict__ b, float eps, bf16_t* __restrict__ X) {
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
       EPI_DGELU_BF16 = 6,  // backward of FFN1's activation: Cb = acc * gelu'(R[t, f])
       EPI_SLAB_F32 = 7 };  // wgrad partial: Cf[split][rows][N] = acc (no bias)

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
  const bf16_t* R;    // EPI_RESID_F32: residual [rows, N] bf16 (EPI_DGELU_BF16: the pre-activation)
  const float* Rf;    // EPI_RESID_F32: fp32 residual instead of R when non-null (gradient residual stream)
  bf16_t *Qo, *Ko, *Vt;  // EPI_QKV (N = 3H): Q [rows, H], K [rows, H], Vt [H, ldt]
  int H;
  int64_t ldt;
  int tilesN, tilesT;
  int dbg_same_tile;  // experiment: every workgroup reads tile (0, 0) (all-L2-hit upper bound); results are garbage
};

// exact-erf GELU (HF "gelu": x * 0.5 * (1 + erf(x / sqrt(2)))).  erf by Abramowitz & Stegun 7.1.26
// (|abs error| <= 1.5e-7, far below the bf16 output rounding) instead of libm's erff: 1 rcp + 1 exp + 7 FMA.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));   // v_rcp_f32 (1 ulp), not the IEEE divide sequence
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.f - p * t * __expf(-z * z);   // erf(|x| / sqrt 2)
  return 0.5f * x + 0.5f * fabsf(x) * e;          // 0.5 x (1 + sign(x) e)
}

// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));   // v_rcp_f32 (1 ulp), not the IEEE divide sequence
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float ex = __expf(-z * z);
  const float e = 1.f - p * t * ex;                       // erf(|x| / sqrt 2)
  const float cdf = 0.5f + (x < 0.f ? -0.5f : 0.5f) * e;  // Phi(x)
  return cdf + x * 0.3989422804014327f * ex;              // exp(-x^2 / 2) == ex
}

template <int EPI, class T>
static __global__ void __launch_bounds__(T::THREADS, 2) k_gemm(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // consecutive logical tiles sweep the feature tiles of one token tile: the activation tile stays in L2
  const uint32_t logical = xcd_remap(blockIdx.x, gridDim.x);
  int tt = logical / a.tilesN, tn = logical - tt * a.tilesN;
  if (a.dbg_same_tile) { tt = 0; tn = 0; }
  const WavePos<T> w;
  GemmAcc<T> acc;
  acc.zero();
  bool tokens_on_regs = false;
  if constexpr (EPI == EPI_QKV) tokens_on_regs = tn * T::TR >= 2 * a.H;   // H % TR == 0 (checked by the launcher)
  const int64_t ldw = a.ldw ? a.ldw : a.K, ldx = a.ldx ? a.ldx : a.K;
  int kbeg = 0, klen = a.K;
  if constexpr (EPI == EPI_SLAB_F32) {
    if (a.k_split_len) { kbeg = blockIdx.y * a.k_split_len; klen = a.k_split_len; }
  }

  if (tokens_on_regs) {  // V third of the fused QKV projection -> Vt[feature][token]: roles swapped
    const int64_t t0 = (int64_t)tt * T::TL;      // the launcher lays tiles out as [tilesT][tilesN] with TR == TL
    const int n0 = tn * T::TR;
    gemm_nt_mainloop<T>(a.X, ldx, a.rows, a.W, ldw, a.N, a.K, t0, n0, smem, acc, w);
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int f = n0 + w.l_index(nt);  // feature on the lane
      if (f >= a.N) continue;
      const float bv = a.bias[f];
      bf16_t* dst = a.Vt + (int64_t)(f - 2 * a.H) * a.ldt;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t t = t0 + w.r_base(mt, g);  // 4 consecutive tokens (rows % 4 == 0)
          if (t < a.rows) {
            const f32x16& v = acc.c[mt][nt];
            uint2 o;
            o.x = pack_bf16x2(v[4 * g + 0] + bv, v[4 * g + 1] + bv);
            o.y = pack_bf16x2(v[4 * g + 2] + bv, v[4 * g + 3] + bv);
            *(uint2*)(dst + t) = o;
          }
        }
    }
    return;
  }

  const int64_t t0 = (int64_t)tt * T::TL;
  const int n0 = tn * T::TR;
  gemm_nt_mainloop<T>(a.W, ldw, a.N, a.X, ldx, a.rows, klen, n0, t0, smem, acc, w, kbeg);

  // ---- epilogue.  The operand buffers are dead: park the tile's bias slice in LDS (no vmcnt round trip per
  // register quad), issue all residual loads of a 32-token column block up front, then convert and store. ----
  __syncthreads();
  float* sbias = (float*)smem;
  for (int i = threadIdx.x; i < T::TR; i += T::THREADS) sbias[i] = (a.bias && n0 + i < a.N) ? a.bias[n0 + i] : 0.f;
  __syncthreads();
  const bool full_n = n0 + T::TR <= a.N;  // workgroup-uniform: no per-quad feature bound checks on the fast path
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const int64_t t = t0 + w.l_index(nt);  // token on the lane
    const bool t_ok = t < a.rows;
    const int64_t tc = t_ok ? t : a.rows - 1;
    uint2 res[T::MT][4];
    if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_DGELU_BF16) {
      if (EPI == EPI_DGELU_BF16 || a.Rf == nullptr) {
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            int f = n0 + w.r_base(mt, g);
            f = (full_n || f < a.N) ? f : a.N - 4;
            res[mt][g] = *(const uint2*)(a.R + tc * a.N + f);
          }
      }
    }
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fl = w.r_base(mt, g);      // 4 consecutive features (N % 4 == 0)
        const int f = n0 + fl;
        const f32x16& v = acc.c[mt][nt];
        const float4 bv = *(const float4*)(sbias + fl);
        float y0 = v[4 * g + 0] + bv.x, y1 = v[4 * g + 1] + bv.y, y2 = v[4 * g + 2] + bv.z, y3 = v[4 * g + 3] + bv.w;
        float p0 = y0, p1 = y1, p2 = y2, p3 = y3;  // pre-activation
        if constexpr (EPI == EPI_GELU_BF16 || EPI == EPI_GELU_SAVE) {
          y0 = gelu_erf(y0); y1 = gelu_erf(y1); y2 = gelu_erf(y2); y3 = gelu_erf(y3);
        }
        if constexpr (EPI == EPI_DGELU_BF16) {
          const uint2 r = res[mt][g];
          y0 *= gelu_grad(__uint_as_float(r.x << 16)); y1 *= gelu_grad(__uint_as_float(r.x & 0xffff0000u));
          y2 *= gelu_grad(__uint_as_float(r.y << 16)); y3 *= gelu_grad(__uint_as_float(r.y & 0xffff0000u));
        }
        if constexpr (EPI == EPI_RESID_F32) {
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
        if (t_ok && (full_n || f < a.N)) {
          if constexpr (EPI == EPI_SLAB_F32) {
            *(float4*)(a.Cf + ((int64_t)blockIdx.y * a.rows + t) * a.N + f) = make_float4(y0, y1, y2, y3);
          } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_F32) {
            *(float4*)(a.Cf + t * a.N + f) = make_float4(y0, y1, y2, y3);
          } else {
            uint2 o;
            o.x = pack_bf16x2(y0, y1);
            o.y = pack_bf16x2(y2, y3);
            if constexpr (EPI == EPI_QKV) {
              bf16_t* dst = f < a.H ? a.Qo + t * a.H + f : a.Ko + t * a.H + (f - a.H);
              *(uint2*)dst = o;
            } else {
              *(uint2*)(a.Cb + t * a.N + f) = o;
              if constexpr (EPI == EPI_GELU_SAVE) {
                uint2 o2;
                o2.x = pack_bf16x2(p0, p1);
                o2.y = pack_bf16x2(p2, p3);
                *(uint2*)(a.Cb2 + t * a.N + f) = o2;
              }
            }
          }
        }
      }
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
struct AttnArgs {
  const bf16_t *Q, *K, *Vt;
  int64_t ldt;
  const int32_t *cu, *lens;
  int H;
  int64_t ldq;      // row stride of Q and K in elements (H, or 3H when they live in a fused [rows, 3H] buffer)
  bf16_t* ctx;
  float* lse;       // [heads, ldt] or nullptr
  float scale;      // 1 / sqrt(head_dim)
};

constexpr int ATT_TILE_PAIR = 2 * 64 * 128;   // K tile + V^T tile, 8 KB each
constexpr int ATT_SMEM_BYTES = 2 * ATT_TILE_PAIR;  // double buffered

static __global__ void __launch_bounds__(256, 4) k_attention_fwd(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.z, h = blockIdx.y;
  const int len = a.lens[b];
  const int q0 = blockIdx.x * 128;
  if (q0 >= len) return;
  const int64_t base = a.cu[b];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int q = q0 + wave * 32 + li;
  const int qc = q < len ? q : len - 1;
  const int H = a.H;

  bf16x8 qf[4];
  {
    const bf16_t* qp = a.Q + (base + qc) * a.ldq + h * 64 + 8 * hi;
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
      glds16((const char*)(a.K + (base + kv0 + row) * a.ldq + h * 64) + gch * 16, smem + buf * ATT_TILE_PAIR + r0 * 128);
      glds16((const char*)(a.Vt + (int64_t)(h * 64 + row) * a.ldt + base + kv0) + gch * 16,
             smem + buf * ATT_TILE_PAIR + 64 * 128 + r0 * 128);
    }
  };
  stage_tile(0, 0);
  for (int kv0 = 0, it = 0; kv0 < len; kv0 += 64, ++it) {
    const int buf = it & 1;
    lds_dma_wait_all();  // explicit: hipcc's automatic vmcnt wait for LDS-DMA is not reliable (gemm_nt.hpp)
    __syncthreads();     // tile `it` landed for everyone; everyone finished reading tile it-1 (the other buffer)
    if (kv0 + 64 < len) stage_tile(kv0 + 64, buf ^ 1);
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

  l += __shfl_xor(l, 32, 64);
  const int plen = a.cu[b + 1] - (int)base;  // len rounded up to the row alignment
  if (q < plen) {
    // alignment rows [len, plen) get zeros: they feed later GEMMs / V^T columns and must stay finite
    const float inv = q < len ? 1.f / l : 0.f;
    bf16_t* dst = a.ctx + (base + q) * H + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 ov;
        ov.x = pack_bf16x2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        ov.y = pack_bf16x2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(dst + dt * 32 + 8 * g + 4 * hi) = ov;
      }
    if (a.lse && hi == 0) a.lse[(int64_t)h * a.ldt + base + q] = q < len ? m * a.scale + logf(l) : 0.f;
  }
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
