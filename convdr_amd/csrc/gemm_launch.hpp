// Host-side launcher of k_gemm (shared by the inference and training orchestration).
#pragma once
#include <stdlib.h>

#include "encoder_kernels.hpp"

namespace convdr {

template <int EPI, class T>
inline int launch_gemm_t(GemmArgs a, hipStream_t st, const char* prof_name) {
  static_assert(T::TR == T::TL, "square tiles: the QKV kernel swaps operand roles per tile");
  static bool attr_done = false;
  if (!attr_done) {
    CONVDR_CHECK_HIP(
        hipFuncSetAttribute((const void*)k_gemm<EPI, T>, hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM_BYTES));
    attr_done = true;
  }
  static const int dbg = getenv("CONVDR_DBG_SAME_TILE") ? atoi(getenv("CONVDR_DBG_SAME_TILE")) : 0;
  a.dbg_same_tile = dbg;
  a.tilesN = (a.N + T::TR - 1) / T::TR;
  a.tilesT = (int)ceil_div64(a.rows, T::TL);
  if (a.tilesT == 0) return 0;
  unsigned splits = 1;
  if (EPI == EPI_SLAB_F32 && a.k_split_len) {
    CONVDR_REQUIRE(a.k_split_len % GEMM_BK == 0 && a.K % a.k_split_len == 0, "gemm: bad split-K slice %d of %d",
                   a.k_split_len, a.K);
    splits = a.K / a.k_split_len;
  }
  ProfScope prof(prof_name, st);
  hipLaunchKernelGGL((k_gemm<EPI, T>), dim3((unsigned)a.tilesN * a.tilesT, splits), dim3(T::THREADS), T::SMEM_BYTES, st,
                     a);
  CONVDR_CHECK_LAUNCH("k_gemm");
  return 0;
}

// 256 x 256 tiles when the problem fills them (N % 256 == 0, for QKV also H % 256 == 0, and enough token rows
// to occupy the 256 CUs), else 128 x 128.
template <int EPI>
inline int launch_gemm(GemmArgs a, hipStream_t st, const char* prof_name) {
  CONVDR_REQUIRE(a.K % GEMM_BK == 0 && a.N % 4 == 0, "gemm: need K %% 64 == 0 and N %% 4 == 0 (K=%d N=%d)", a.K, a.N);
  if (EPI == EPI_QKV) CONVDR_REQUIRE(a.H % 128 == 0, "gemm: fused QKV needs hidden %% 128 == 0 (%d)", a.H);
  const bool fits = a.N % 256 == 0 && (EPI != EPI_QKV || a.H % 256 == 0);
  int64_t tiles256 = (int64_t)(a.N / 256) * ceil_div64(a.rows, 256);
  if (EPI == EPI_SLAB_F32 && a.k_split_len) tiles256 *= a.K / a.k_split_len;
  if (fits && tiles256 >= 192) return launch_gemm_t<EPI, Tile256>(a, st, prof_name);
  return launch_gemm_t<EPI, Tile128>(a, st, prof_name);
}

}  // namespace convdr
