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
  unsigned long long* trace;   // experiment: [workgroup][16] s_memtime stamps of thread 0 (or null)
};
#ifdef CONVDR_ENABLE_TRACE   // make TRACE=1: phase stamps for tools/gemm_trace_ln.py
#define CONVDR_LN_TRACE(ph) \
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + (ph)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_LN_TRACE(ph)
#endif

// 32-wide K slices: 64-byte LDS rows, 16 rows per wave per round, rounds 128 rows apart (so the swizzle term
// (row >> 2) & 3 does not depend on the round).  Same buffer-descriptor addressing as gemm_nt.hpp's gemm_stage.
__device__ __forceinline__ StageSrc ln_stage_src(const bf16_t* __restrict__ G, int64_t ld, int64_t row0, int64_t nrows,
                                                  int wave, int lane) {
  StageSrc s;
  int64_t bytes = (nrows - row0) * ld * 2;
  bytes = bytes < 0 ? 0 : (bytes > 0xffffffffll ? 0xffffffffll : bytes);
  const uint64_t base = (uint64_t)(G + row0 * ld);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
  const int row = wave * 16 + (lane >> 2);
  const int gch = (lane & 3) ^ ((row >> 2) & 3);
  s.voff = (uint32_t)(row * ld * 2) + gch * 16;
  s.round_pitch = __builtin_amdgcn_readfirstlane((uint32_t)(128 * ld * 2));
  return s;
}

template <int ROWS>
__device__ __forceinline__ void ln_stage32(const StageSrc& s, int ks, char* lds_tile, int wave) {
  constexpr int ROUNDS = ROWS / (16 * 8);
  static_assert(ROUNDS * 128 == ROWS, "rows must be a multiple of 128");
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lptr_t)(lds_tile + (i * 8 + wave) * 16 * 64), 16, s.voff,
                                             i * s.round_pitch + ks * (LN_SLICE * 2), 0, 0);
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
  CONVDR_LN_TRACE(0)
  const int nk = a.K / LN_SLICE;
  const int sw = (w.li >> 2) & 3;
  const int offR = (w.wr * T::MT * 32 + w.li) * 64;
  const int offL = (w.wl * T::NT * 32 + w.li) * 64;

  const StageSrc srcW = ln_stage_src(a.W, a.K, 0, T::TR, w.wave, w.lane);
  const StageSrc srcA = ln_stage_src(a.A, a.K, t0, a.rows, w.wave, w.lane);
  ln_stage32<T::TR>(srcW, 0, sR, w.wave);
  ln_stage32<T::TL>(srcA, 0, sL, w.wave);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    lds_dma_wait_all();
    __syncthreads();
    if (kt + 1 < nk) {
      ln_stage32<T::TR>(srcW, kt + 1, sR + (buf ^ 1) * LN_R_BYTES, w.wave);
      ln_stage32<T::TL>(srcA, kt + 1, sL + (buf ^ 1) * LN_L_BYTES, w.wave);
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
  CONVDR_LN_TRACE(1)
  __syncthreads();   // operand buffers are dead
  float* sBias = (float*)smem;           // [768]
  float* sGam = sBias + 768;
  float* sBet = sGam + 768;
  float* sRed = sBet + 768;              // [128 tokens][8 partial slots]
  float* sStat = sRed + 128 * 8;         // [128] mean, then rstd
  for (int i = threadIdx.x; i < 768; i += 512) { sBias[i] = a.bias[i]; sGam[i] = a.gamma[i]; sBet[i] = a.beta[i]; }
  __syncthreads();
  CONVDR_LN_TRACE(2)
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
      // Residual quads as 16-byte loads: the lower half-wave fetches features [16 p, 16 p + 8) of the row -- its own
      // quad g = 2 p and the upper half-wave's -- the upper half-wave [16 p + 8, 16 p + 16) -- the lower's quad
      // g = 2 p + 1 and its own; one v_permlane32_swap per dword hands the foreign halves over.  Half the load
      // instructions of the 8-byte form and 32 contiguous bytes per row per instruction.
      uint4 res[3][2];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          res[m][p] = *(const uint4*)(a.R + tc * 768 + (w.wr * T::MT + mh + m) * 32 + 16 * p + 8 * w.hi);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          uint4 r4 = res[m][p];
          auto sx = __builtin_amdgcn_permlane32_swap(r4.x, r4.z, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(r4.y, r4.w, false, false);
          const uint2 rq[2] = {make_uint2(sx[0], sy[0]), make_uint2(sx[1], sy[1])};   // quads g = 2 p, 2 p + 1
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int g = 2 * p + q;
            f32x16& v = acc.c[mh + m][nt];
            const float4 bv = *(const float4*)(sBias + w.r_base(mh + m, g));
            const uint2 r = rq[q];
            v[4 * g + 0] += bv.x + __uint_as_float(r.x << 16);
            v[4 * g + 1] += bv.y + __uint_as_float(r.x & 0xffff0000u);
            v[4 * g + 2] += bv.z + __uint_as_float(r.y << 16);
            v[4 * g + 3] += bv.w + __uint_as_float(r.y & 0xffff0000u);
            s += (v[4 * g + 0] + v[4 * g + 1]) + (v[4 * g + 2] + v[4 * g + 3]);
          }
        }
    }
    sRed[(w.wl * 64 + nt * 32 + w.li) * 8 + slot] = s;
  }
  CONVDR_LN_TRACE(3)
  __syncthreads();
  CONVDR_LN_TRACE(4)
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
  CONVDR_LN_TRACE(5)
  __syncthreads();
  if (threadIdx.x < 128) {
    const float* p = sRed + threadIdx.x * 8;
    const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * (1.f / 768.f);
    sStat[128 + threadIdx.x] = rsqrtf(var + a.eps);
  }
  __syncthreads();
  CONVDR_LN_TRACE(6)
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    if (!ok[nt]) continue;
    const float rstd = sStat[128 + w.wl * 64 + nt * 32 + w.li];
    bf16_t* dst = a.X + tok[nt] * 768;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        uint2 o[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int g = 2 * p + q;
          const int f = w.r_base(mt, g);
          const float4 gg = *(const float4*)(sGam + f), bb = *(const float4*)(sBet + f);
          const f32x16& v = acc.c[mt][nt];
          o[q].x = pack_bf16x2_sw((v[4 * g + 0] - mean[nt]) * rstd * gg.x + bb.x, (v[4 * g + 1] - mean[nt]) * rstd * gg.y + bb.y);
          o[q].y = pack_bf16x2_sw((v[4 * g + 2] - mean[nt]) * rstd * gg.z + bb.z, (v[4 * g + 3] - mean[nt]) * rstd * gg.w + bb.w);
        }
        // the same half-wave exchange as the residual loads, the other way round: 16 contiguous bytes per lane
        auto sx = __builtin_amdgcn_permlane32_swap(o[0].x, o[1].x, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(o[0].y, o[1].y, false, false);
        *(uint4*)(dst + (w.wr * T::MT + mt) * 32 + 16 * p + 8 * w.hi) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
  }
  CONVDR_LN_TRACE(7)
}

}  // namespace convdr
