// bf16 MFMA "NT" tile engine for gfx950:  acc[128 x 128] = A[m0.., :] * B[n0.., :]^T
// with A [M, K] and B [N, K] both row-major bf16 (K contiguous), fp32 accumulate.
//
// This one main loop serves every dense contraction on the ConvDR hot path:
//   * similarity scan     S = P_block * Q^T          (ip_topk.hip; replaces the SGEMM inside
//                                                     faiss.IndexFlatIP.search, run_convdr_inference.py:182)
//   * encoder projections Y = X * W^T (+ epilogue)    (encoder.hip; nn.Linear weights are [out, in])
//
// Geometry (CDNA4): 256 threads = 4 waves (2 x 2), wave tile 64 x 64 = 2 x 2 MFMA 32x32x16 tiles,
// BK = 64.  Operand tiles are staged HBM/L2 -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4),
// double buffered (2 x 32 KB), one barrier per K step: the loads for step t+1 fly under the 16 MFMAs
// of step t.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 so that
// the ds_read_b128 fragment reads (32 rows x one chunk) are bank-conflict free.  LDS-DMA writes
// lane-linear, so the swizzle is applied to the per-lane GLOBAL source address (it stays inside the
// row's 128-byte line, coalescing is unchanged).
#pragma once
#include "common.hpp"

namespace convdr {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64, GEMM_THREADS = 256;
constexpr int GEMM_TILE_BYTES = 128 * GEMM_BK * 2;        // 16 KB per operand tile
constexpr int GEMM_SMEM_BYTES = 4 * GEMM_TILE_BYTES;      // A[2] + B[2] = 64 KB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Stage rows [row0, row0+128) x k-chunk kt of G ([nrows, ld] bf16) into a 16 KB LDS tile.
// Rows past nrows-1 are clamped (their products are never stored).
__device__ __forceinline__ void gemm_stage(const bf16_t* __restrict__ G, int64_t ld, int64_t row0,
                                           int64_t nrows, int kt, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r0 = (i * 4 + wave) * 8;
    const int row = r0 + (lane >> 3);
    int64_t grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    const int gch = (lane & 7) ^ ((row >> 1) & 7);
    const char* src = (const char*)G + ((grow * ld + (int64_t)kt * GEMM_BK) << 1) + gch * 16;
    glds16(src, lds_tile + r0 * 128);
  }
}

struct GemmAcc {
  f32x16 c[2][2];  // [mt][nt]; element r of lane l: row = mt*32 + (r&3) + 8*(r>>2) + 4*(l>>5), col = nt*32 + (l&31)
};

// acc += A[m0:m0+128, :] * B[n0:n0+128, :]^T   (K must be a multiple of 64)
__device__ __forceinline__ void gemm_nt_mainloop(const bf16_t* __restrict__ A, int64_t lda, int64_t M,
                                                 const bf16_t* __restrict__ B, int64_t ldb, int64_t N,
                                                 int K, int64_t m0, int64_t n0, char* smem, GemmAcc& acc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  char* sA = smem;
  char* sB = smem + 2 * GEMM_TILE_BYTES;
  const int nk = K / GEMM_BK;

  const int sw = (lane >> 1) & 7, hi = lane >> 5;
  const int offA = (wm * 64 + (lane & 31)) * 128;
  const int offB = (wn * 64 + (lane & 31)) * 128;

  gemm_stage(A, lda, m0, M, 0, sA, wave, lane);
  gemm_stage(B, ldb, n0, N, 0, sB, wave, lane);

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();  // tile kt landed for every wave (compiler drains vmcnt before the barrier);
                      // every wave is done reading buffer buf^1 (step kt-1)
    if (kt + 1 < nk) {
      gemm_stage(A, lda, m0, M, kt + 1, sA + (buf ^ 1) * GEMM_TILE_BYTES, wave, lane);
      gemm_stage(B, ldb, n0, N, kt + 1, sB + (buf ^ 1) * GEMM_TILE_BYTES, wave, lane);
    }
    const char* tA = sA + buf * GEMM_TILE_BYTES + offA;
    const char* tB = sB + buf * GEMM_TILE_BYTES + offB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = ((2 * s + hi) ^ sw) * 16;
      bf16x8 a0 = *(const bf16x8*)(tA + ch);
      bf16x8 a1 = *(const bf16x8*)(tA + 32 * 128 + ch);
      bf16x8 b0 = *(const bf16x8*)(tB + ch);
      bf16x8 b1 = *(const bf16x8*)(tB + 32 * 128 + ch);
      acc.c[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc.c[0][0], 0, 0, 0);
      acc.c[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc.c[0][1], 0, 0, 0);
      acc.c[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc.c[1][0], 0, 0, 0);
      acc.c[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc.c[1][1], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void gemm_acc_zero(GemmAcc& acc) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;
}

// row (within the 128-row tile) of accumulator element r in MFMA tile mt for this lane
__device__ __forceinline__ int gemm_acc_row(int wm, int mt, int r, int lane) {
  return wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int gemm_acc_col(int wn, int nt, int lane) { return wn * 64 + nt * 32 + (lane & 31); }

}  // namespace convdr
