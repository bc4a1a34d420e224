// Row-complete GEMM with fused residual + LayerNorm epilogue (inference path, hidden size 768):
//     X_out[t, :] = LayerNorm( A[t, :] . W^T + bias + R[t, :] ) * gamma + beta        (bf16 out)
// replaces  k_gemm<EPI_RESID_F32> (fp32 pre-LN sums to HBM) + k_layernorm (read them back): one workgroup owns
// 128 tokens x ALL 768 output features, so the row statistics are available on chip and the 3 KB/token fp32
// round trip (27 % of the encoder's HBM traffic) disappears.
//   HF call sites: BertSelfOutput / BertOutput  (dense -> dropout -> LayerNorm(x + residual)), reached from
//   /root/reference/model/models.py:141-142.
// Geometry: 8 waves = 4 (features) x 2 (tokens); wave tile 192 features x 64 tokens = 6 x 2 MFMA 32x32x16 tiles
// (192 accumulator VGPRs).  Operands staged by LDS-DMA in 32-wide K slices, double buffered:
// W slice 768 x 64 B = 48 KB, A slice 128 x 64 B = 8 KB  -> 112 KB LDS, one workgroup per CU.
// 64-byte LDS rows: 16-byte chunk index XOR (row >> 2) & 3 keeps the 32-row ds_read_b128 fragment reads
// conflict-free (applied to the DMA source address, LDS-DMA writes lane-linear).
#pragma once
#include "gemm_nt.hpp"

namespace convdr {

using TileLN = TileCfg<4, 2, 6, 2>;   // TR = 768 features, TL = 128 tokens
constexpr int LN_SLICE = 32;
constexpr int LN_R_BYTES = TileLN::TR * 64, LN_L_BYTES = TileLN::TL * 64;
constexpr int LN_SMEM_BYTES = 2 * (LN_R_BYTES + LN_L_BYTES);   // 112 KB

struct GemmLnArgs {
  const bf16_t* W;      // [768, K]
  const bf16_t* A;      // [rows, K]
  int64_t rows;
  int K;
  const float* bias;    // [768]
  const bf16_t* R;      // residual [rows, 768]
  const float *gamma, *beta;
  float eps;
  bf16_t* X;            // out [rows, 768] (may alias R: a workgroup reads its residual rows before it writes them)
};

template <int ROWS>
__device__ __forceinline__ void ln_stage32(const bf16_t* __restrict__ G, int64_t ld, int64_t row0, int64_t nrows, int ks,
                                           char* lds_tile, int wave, int lane) {
  constexpr int ROUNDS = ROWS / (16 * 8);
  static_assert(ROUNDS * 128 == ROWS, "rows must be a multiple of 128");
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const int r0 = (i * 8 + wave) * 16;
    const int row = r0 + (lane >> 2);
    int64_t grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    const int gch = (lane & 3) ^ ((row >> 2) & 3);
    glds16((const char*)G + ((grow * ld + (int64_t)ks * LN_SLICE) << 1) + gch * 16, lds_tile + r0 * 64);
  }
}

__global__ void __launch_bounds__(512, 2) k_gemm_resid_ln(const GemmLnArgs a) {
  using T = TileLN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WavePos<T> w;
  const int64_t t0 = (int64_t)blockIdx.x * T::TL;
  char* sR = smem;
  char* sL = smem + 2 * LN_R_BYTES;
  GemmAcc<T> acc;
  acc.zero();
  const int nk = a.K / LN_SLICE;
  const int sw = (w.li >> 2) & 3;
  const int offR = (w.wr * T::MT * 32 + w.li) * 64;
  const int offL = (w.wl * T::NT * 32 + w.li) * 64;

  ln_stage32<T::TR>(a.W, a.K, 0, T::TR, 0, sR, w.wave, w.lane);
  ln_stage32<T::TL>(a.A, a.K, t0, a.rows, 0, sL, w.wave, w.lane);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    lds_dma_wait_all();
    __syncthreads();
    if (kt + 1 < nk) {
      ln_stage32<T::TR>(a.W, a.K, 0, T::TR, kt + 1, sR + (buf ^ 1) * LN_R_BYTES, w.wave, w.lane);
      ln_stage32<T::TL>(a.A, a.K, t0, a.rows, kt + 1, sL + (buf ^ 1) * LN_L_BYTES, w.wave, w.lane);
    }
    const char* tR = sR + buf * LN_R_BYTES + offR;
    const char* tL = sL + buf * LN_L_BYTES + offL;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ch = ((2 * s + w.hi) ^ sw) * 16;
      bf16x8 fa[T::MT], fb[T::NT];
#pragma unroll
      for (int j = 0; j < T::NT; ++j) fb[j] = *(const bf16x8*)(tL + j * 32 * 64 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i) fa[i] = *(const bf16x8*)(tR + i * 32 * 64 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
          acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc.c[i][j], 0, 0, 0);
    }
  }

  // ---------------- epilogue: + bias + residual, LayerNorm over the 768 features of each token ----------------
  __syncthreads();   // operand buffers are dead
  float* sBias = (float*)smem;           // [768]
  float* sGam = sBias + 768;
  float* sBet = sGam + 768;
  float* sRed = sBet + 768;              // [128 tokens][8 partial slots]
  float* sStat = sRed + 128 * 8;         // [128] mean, then rstd
  for (int i = threadIdx.x; i < 768; i += 512) { sBias[i] = a.bias[i]; sGam[i] = a.gamma[i]; sBet[i] = a.beta[i]; }
  __syncthreads();
  const int slot = w.wr * 2 + w.hi;
  int64_t tok[T::NT];
  bool ok[T::NT];
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    tok[nt] = t0 + w.l_index(nt);
    ok[nt] = tok[nt] < a.rows;
  }
  // y = acc + bias + residual (kept in the accumulator registers); partial row sums
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const int64_t tc = ok[nt] ? tok[nt] : a.rows - 1;
    float s = 0.f;
#pragma unroll
    for (int mh = 0; mh < T::MT; mh += 3) {
      uint2 res[3][4];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) res[m][g] = *(const uint2*)(a.R + tc * 768 + w.r_base(mh + m, g));
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x16& v = acc.c[mh + m][nt];
          const float4 bv = *(const float4*)(sBias + w.r_base(mh + m, g));
          const uint2 r = res[m][g];
          v[4 * g + 0] += bv.x + __uint_as_float(r.x << 16);
          v[4 * g + 1] += bv.y + __uint_as_float(r.x & 0xffff0000u);
          v[4 * g + 2] += bv.z + __uint_as_float(r.y << 16);
          v[4 * g + 3] += bv.w + __uint_as_float(r.y & 0xffff0000u);
          s += (v[4 * g + 0] + v[4 * g + 1]) + (v[4 * g + 2] + v[4 * g + 3]);
        }
    }
    sRed[(w.wl * 64 + nt * 32 + w.li) * 8 + slot] = s;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const float* p = sRed + threadIdx.x * 8;
    sStat[threadIdx.x] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * (1.f / 768.f);
  }
  __syncthreads();
  float mean[T::NT];
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    mean[nt] = sStat[w.wl * 64 + nt * 32 + w.li];
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc.c[mt][nt][r] - mean[nt];
        q += d * d;
      }
    sRed[(w.wl * 64 + nt * 32 + w.li) * 8 + slot] = q;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const float* p = sRed + threadIdx.x * 8;
    const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * (1.f / 768.f);
    sStat[128 + threadIdx.x] = rsqrtf(var + a.eps);
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    if (!ok[nt]) continue;
    const float rstd = sStat[128 + w.wl * 64 + nt * 32 + w.li];
    bf16_t* dst = a.X + tok[nt] * 768;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = w.r_base(mt, g);
        const float4 gg = *(const float4*)(sGam + f), bb = *(const float4*)(sBet + f);
        const f32x16& v = acc.c[mt][nt];
        uint2 o;
        o.x = pack_bf16x2_sw((v[4 * g + 0] - mean[nt]) * rstd * gg.x + bb.x, (v[4 * g + 1] - mean[nt]) * rstd * gg.y + bb.y);
        o.y = pack_bf16x2_sw((v[4 * g + 2] - mean[nt]) * rstd * gg.z + bb.z, (v[4 * g + 3] - mean[nt]) * rstd * gg.w + bb.w);
        *(uint2*)(dst + f) = o;
      }
  }
}

}  // namespace convdr
