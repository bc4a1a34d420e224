// bf16 MFMA "NT" tile engine for gfx950:  acc[TR x TL] = R[r0.., :] * L[l0.., :]^T
// with both operands row-major bf16 [rows, K] (K contiguous), fp32 accumulate.
//
// This one main loop serves every dense contraction on the ConvDR hot path:
//   * similarity scan     S = P_block * Q^T          (ip_topk.hip; replaces the SGEMM inside
//                                                     faiss.IndexFlatIP.search, run_convdr_inference.py:182)
//   * encoder projections Y = X * W^T (+ epilogue)    (encoder*.hip; nn.Linear weights are [out, in])
//
// The "R" operand's row index lands on accumulator REGISTERS (4 consecutive rows per register quad), the "L"
// operand's row index on LANES (v_mfma_f32_32x32x16_bf16: D[i][j], j = lane & 31, i = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)).
//
// Geometry is a template: WR x WL waves, each owning MT x NT MFMA 32x32 tiles, BK = 64.
//   Tile128: 2 x 2 waves, 2 x 2 tiles -> 128 x 128, 256 threads, 64 KB LDS, 2 workgroups per CU
//   Tile256: 2 x 4 waves, 4 x 2 tiles -> 256 x 256, 512 threads, 128 KB LDS, 1 workgroup per CU; halves the
//            L2->LDS bytes per FLOP and makes one K step long enough (2048 MFMA cycles per SIMD) that the
//            one-step-ahead LDS-DMA prefetch covers HBM latency without a second resident workgroup
// Operand tiles are staged HBM/L2 -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered, one
// barrier per K step: the loads for step t+1 fly under the MFMAs of step t.  LDS rows are 128 B; the 16-byte
// chunk index is XOR-swizzled with (row >> 1) & 7 so that the ds_read_b128 fragment reads (32 rows x one chunk)
// are bank-conflict free.  LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane GLOBAL source
// address (it stays inside the row's 128-byte line, coalescing is unchanged).
#pragma once
#include "common.hpp"

namespace convdr {

constexpr int GEMM_BK = 64;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// all of this wave's outstanding LDS-DMA (and other vector-memory) operations have completed
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int WR_, int WL_, int MT_, int NT_>
struct TileCfg {
  static constexpr int WR = WR_, WL = WL_, MT = MT_, NT = NT_;
  static constexpr int TR = WR * MT * 32, TL = WL * NT * 32;   // tile extent on the R / L operand
  static constexpr int WAVES = WR * WL, THREADS = WAVES * 64;
  static constexpr int R_BYTES = TR * 128, L_BYTES = TL * 128;  // one K step of each operand
  static constexpr int SMEM_BYTES = 2 * (R_BYTES + L_BYTES);
  static constexpr int MIN_WAVES_PER_SIMD = THREADS >= 512 ? 2 : 2;
};
using Tile128 = TileCfg<2, 2, 2, 2>;
using Tile256 = TileCfg<2, 4, 4, 2>;

template <class T>
struct GemmAcc {
  f32x16 c[T::MT][T::NT];
  // element r of c[mt][nt] for lane l:  R index = wr*MT*32 + mt*32 + (r&3) + 8*(r>>2) + 4*(l>>5)
  //                                      L index = wl*NT*32 + nt*32 + (l&31)
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
      for (int j = 0; j < T::NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
  }
};

template <class T>
struct WavePos {
  int lane, wave, wr, wl, hi, li;
  __device__ __forceinline__ WavePos() {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    wr = wave / T::WL;
    wl = wave - wr * T::WL;
    hi = lane >> 5;
    li = lane & 31;
  }
  // first of the 4 consecutive R indices held in registers 4g..4g+3 of tile mt
  __device__ __forceinline__ int r_base(int mt, int g) const { return (wr * T::MT + mt) * 32 + 8 * g + 4 * hi; }
  __device__ __forceinline__ int r_index(int mt, int r) const { return (wr * T::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }
  __device__ __forceinline__ int l_index(int nt) const { return (wl * T::NT + nt) * 32 + li; }
};

// Stage rows [row0, row0 + ROWS) x k-chunk kt of G ([nrows, ld] bf16) into an LDS tile of ROWS x 128 B.
// Rows past nrows-1 are clamped (their products are never stored).
template <int ROWS, int WAVES>
__device__ __forceinline__ void gemm_stage(const bf16_t* __restrict__ G, int64_t ld, int64_t row0, int64_t nrows,
                                           int kt, char* lds_tile, int wave, int lane) {
  constexpr int ROUNDS = ROWS / (8 * WAVES);
  static_assert(ROUNDS * 8 * WAVES == ROWS, "tile rows must be a multiple of 8 * waves");
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const int r0 = (i * WAVES + wave) * 8;
    const int row = r0 + (lane >> 3);
    int64_t grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    const int gch = (lane & 7) ^ ((row >> 1) & 7);
    const char* src = (const char*)G + ((grow * ld + (int64_t)kt * GEMM_BK) << 1) + gch * 16;
    glds16(src, lds_tile + r0 * 128);
  }
}

// acc += R[r0:r0+TR, :] * L[l0:l0+TL, :]^T   (K must be a multiple of 64)
template <class T>
__device__ __forceinline__ void gemm_nt_mainloop(const bf16_t* __restrict__ R, int64_t ldr, int64_t nR,
                                                 const bf16_t* __restrict__ L, int64_t ldl, int64_t nL, int K,
                                                 int64_t r0, int64_t l0, char* smem, GemmAcc<T>& acc,
                                                 const WavePos<T>& w, int k_begin = 0) {
  // contraction range [k_begin, k_begin + K): split-K callers pass a slice
  R += k_begin;
  L += k_begin;
  char* sR = smem;
  char* sL = smem + 2 * T::R_BYTES;
  const int nk = K / GEMM_BK;
  const int sw = (w.lane >> 1) & 7;
  const int offR = (w.wr * T::MT * 32 + w.li) * 128;
  const int offL = (w.wl * T::NT * 32 + w.li) * 128;

  gemm_stage<T::TR, T::WAVES>(R, ldr, r0, nR, 0, sR, w.wave, w.lane);
  gemm_stage<T::TL, T::WAVES>(L, ldl, l0, nL, 0, sL, w.wave, w.lane);

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // Tile kt must have LANDED in LDS for every wave before anyone reads it.  LDS-DMA completion is tracked by
    // vmcnt; hipcc's automatic wait before the barrier is NOT reliable for global_load_lds (observed missing in
    // one of two inlined copies of this loop -> rare stale operand rows), so drain explicitly.
    lds_dma_wait_all();
    __syncthreads();  // ... and every wave is done reading buffer buf^1 (step kt-1)
    if (kt + 1 < nk) {
      gemm_stage<T::TR, T::WAVES>(R, ldr, r0, nR, kt + 1, sR + (buf ^ 1) * T::R_BYTES, w.wave, w.lane);
      gemm_stage<T::TL, T::WAVES>(L, ldl, l0, nL, kt + 1, sL + (buf ^ 1) * T::L_BYTES, w.wave, w.lane);
    }
    const char* tR = sR + buf * T::R_BYTES + offR;
    const char* tL = sL + buf * T::L_BYTES + offL;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = ((2 * s + w.hi) ^ sw) * 16;
      bf16x8 a[T::MT], b[T::NT];
#pragma unroll
      for (int j = 0; j < T::NT; ++j) b[j] = *(const bf16x8*)(tL + j * 32 * 128 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i) a[i] = *(const bf16x8*)(tR + i * 32 * 128 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
          acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc.c[i][j], 0, 0, 0);
    }
  }
}

}  // namespace convdr
