// Host-side launcher of k_gemm (shared by the inference and training orchestration).
#pragma once
#include <stdlib.h>

#include "encoder_kernels.hpp"

namespace convdr {

// experiment hook: when set (convdr_set_option "gemm_trace" = device pointer), FFN1 launches record phase stamps
inline void* g_gemm_trace = nullptr;
inline void* g_gemm_trace_ln = nullptr;
inline void* g_attn_trace = nullptr;
inline void* g_clock_probe = nullptr;   // convdr_set_option "clock_probe": [slots][4] uint64, one slot per FFN1 launch (round robin)
inline int64_t g_clock_probe_slots = 1, g_clock_probe_next = 0;

template <int EPI, class T>
inline int launch_gemm_t(GemmArgs a, hipStream_t st, const char* prof_name) {
  static_assert(T::TR == T::TL || EPI != EPI_QKV, "square tiles: the QKV kernel swaps operand roles per tile");
  static DeviceOnce attr_done;   // function attributes are per device
  if (attr_done.first()) {
    CONVDR_CHECK_HIP(
        hipFuncSetAttribute((const void*)k_gemm<EPI, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            T::WAVES == 8 ? 160 * 1024 : T::SMEM_BYTES + T::TR * 4));
  }
#ifdef CONVDR_ENABLE_TRACE   // timing experiments that produce garbage results: only in the `make TRACE=1` library
  static const int dbg = getenv("CONVDR_DBG_SAME_TILE") ? atoi(getenv("CONVDR_DBG_SAME_TILE")) : 0;
  a.dbg_same_tile = dbg;
  static const int dbg_epi = getenv("CONVDR_DBG_SKIP_EPI") ? atoi(getenv("CONVDR_DBG_SKIP_EPI")) : 0;
  a.dbg_skip_epi = dbg_epi;
  static const int dbg_pre = getenv("CONVDR_DBG_PRELANDED") ? atoi(getenv("CONVDR_DBG_PRELANDED")) : 0;
  a.dbg_prelanded = dbg_pre;
#endif
  static const int trace_epi = getenv("CONVDR_TRACE_EPI") ? atoi(getenv("CONVDR_TRACE_EPI")) : (int)EPI_GELU_BF16;
  a.trace = (EPI == trace_epi) ? (unsigned long long*)g_gemm_trace : nullptr;
  a.clock_probe = nullptr;
  if (g_clock_probe && strcmp(prof_name, "gemm_ffn1") == 0) {
    a.clock_probe = (unsigned long long*)g_clock_probe + 4 * (g_clock_probe_next % g_clock_probe_slots);
    ++g_clock_probe_next;
  }
  a.nt_out = NT_CTILE && (int64_t)a.rows * a.N * 2 >= ((int64_t)128 << 20);
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
  // persistent walk: at most one workgroup per resident slot (Tile256: 1 per CU, Tile128: 2 per CU)
  static const int dbg_np = getenv("CONVDR_DBG_NONPERSISTENT") ? atoi(getenv("CONVDR_DBG_NONPERSISTENT")) : 0;
  const int64_t slots = (int64_t)device_cu_count() * (T::SMEM_BYTES > 80 * 1024 ? 1 : 2);
  int64_t grid = (int64_t)a.tilesN * a.tilesT;
  if (!dbg_np && splits == 1 && grid > slots) grid = slots;
  const size_t lds = T::WAVES == 8 ? 160 * 1024 : T::SMEM_BYTES + T::TR * 4;
  hipLaunchKernelGGL((k_gemm<EPI, T>), dim3((unsigned)grid, splits), dim3(T::THREADS), lds, st, a);
#ifdef CONVDR_ENABLE_TRACE
  // CONVDR_DBG_DOUBLE (tools/dbg/double_probe.sh): a class of idempotent launches goes out TWICE -- the step's difference is that
  // class's marginal cost with the product's own operands (1 = the training / inference forward's GEMMs, 2 = data-gradient GEMMs)
  static const int dbl = getenv("CONVDR_DBG_DOUBLE") ? atoi(getenv("CONVDR_DBG_DOUBLE")) : 0;
  if (dbl & (strcmp(prof_name, "gemm_dgrad") == 0 ? 2 : strncmp(prof_name, "gemm_", 5) == 0 && strcmp(prof_name, "gemm_cls") != 0 ? 1 : 0))
    hipLaunchKernelGGL((k_gemm<EPI, T>), dim3((unsigned)grid, splits), dim3(T::THREADS), lds, st, a);
#endif
  CONVDR_CHECK_LAUNCH("k_gemm");
  return 0;
}

// Tile choice.  256 x 256 when the problem fills the chip with them (N % 256 == 0, for QKV also H % 256 == 0, >= 192
// tiles); else, for N % 256 == 0, whichever of 256 x 256 / 256 x 128 (TileWide) / 128 x 128 has the cheapest last round:
// cost = rounds over the resident slots x (K steps x step time + epilogue), step times as measured on MI355X at the
// configs[2] training size (profiles/r04_tile_policy.txt).  g_gemm_tile_policy (convdr_set_option "gemm_tile_policy"):
// 0 = this model, 1 / 2 / 3 = force 256 x 256 / 256 x 128 / 128 x 128 where the shape allows (A/B runs).
inline int64_t g_gemm_tile_policy = 0;
// convdr_set_option "attn_bwd_fused": 1 (default) = sequences of at most 256 tokens take k_attention_bwd_fused (dQ, dK, dV in one
// workgroup per (sequence, head)); 0 = always the dQ kernel + the dK / dV kernel (tests and A/B runs exercise both)
inline int64_t g_attn_bwd_fused = 1;
// convdr_set_option "embed_bwd_deterministic": 1 = the word / position embedding gradients are summed in token-row order by
// table-row owners (k_embed_scatter_det) instead of with fp32 atomics: bitwise run-to-run reproducible like every other gradient
inline int64_t g_embed_bwd_det = getenv("CONVDR_EMBED_BWD_DETERMINISTIC") ? atoi(getenv("CONVDR_EMBED_BWD_DETERMINISTIC")) : 0;
// convdr_set_option "gelu_gp": 1 (default) = the training forward's FFN1 writes gelu'(pre-activation) (EPI_GELU_GP) and the backward's
// FFN2 data-gradient GEMM multiplies by it in its epilogue (EPI_MUL_GP); 0 = the round-2..4 form (pre-activation saved, separate
// k_dgelu_colsum pass).  Read by the forward AND its backward: change it only between steps.
inline int64_t g_gelu_gp = 1;
// convdr_set_option "ln_bwd_rows": LayerNorm backward of the encoder layers (H = 768): 0 = the general kernel, 1 = the straight-line
// form (k_layernorm_bwd_rows) without, 2 (default) = with the register prefetch of the next row.  Same formulas; rounding-level differences.
inline int64_t g_ln_bwd_rows = getenv("CONVDR_LN_BWD_ROWS") ? atoi(getenv("CONVDR_LN_BWD_ROWS")) : 2;
struct TileCost { double step_us, epi_us; int per_cu; };
inline double gemm_tile_cost(int64_t tiles, int nk, const TileCost& c) {
  const int64_t slots = (int64_t)device_cu_count() * c.per_cu;
  return (double)ceil_div64(tiles, slots) * (nk * c.step_us + c.epi_us);
}
template <int EPI>
inline int launch_gemm(GemmArgs a, hipStream_t st, const char* prof_name) {
  CONVDR_REQUIRE(a.K % GEMM_BK == 0 && a.N % 4 == 0, "gemm: need K %% 64 == 0 and N %% 4 == 0 (K=%d N=%d)", a.K, a.N);
  if (EPI == EPI_QKV) CONVDR_REQUIRE(a.H % 128 == 0 && a.ldt % 8 == 0, "gemm: fused QKV needs hidden %% 128 == 0 (%d)", a.H);
  if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_GELU_SAVE || EPI == EPI_GELU_BLK || EPI == EPI_GELU_GP || EPI == EPI_MUL_GP)
    CONVDR_REQUIRE(a.N % 8 == 0, "gemm: bf16 outputs are stored 16 bytes at a time, need N %% 8 == 0 (N=%d)", a.N);
  const bool fits = a.N % 256 == 0 && (EPI != EPI_QKV || a.H % 256 == 0);
  int splits = 1;
  if (EPI == EPI_SLAB_F32 && a.k_split_len) splits = a.K / a.k_split_len;
  const int64_t tiles256 = (int64_t)(a.N / 256) * ceil_div64(a.rows, 256) * splits;
  static const int force128 = getenv("CONVDR_DBG_TILE128") ? atoi(getenv("CONVDR_DBG_TILE128")) : 0;
  static const int min256 = getenv("CONVDR_TILE256_MIN_TILES") ? atoi(getenv("CONVDR_TILE256_MIN_TILES")) : 192;   // A/B knob
  constexpr bool WIDE_OK = EPI == EPI_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_GELU_SAVE || EPI == EPI_GELU_BF16 || EPI == EPI_F32 ||
                           EPI == EPI_GELU_GP || EPI == EPI_MUL_GP;
  int choice = (fits && tiles256 >= min256) ? 256 : 128;
  if (fits && !force128 && g_gemm_tile_policy == 1) choice = 256;
  else if (fits && !force128 && g_gemm_tile_policy == 3) choice = 128;
  else if (fits && WIDE_OK && !force128) {
    if (g_gemm_tile_policy == 2) choice = 192;
    else {
      const int nk = (EPI == EPI_SLAB_F32 && a.k_split_len ? a.k_split_len : a.K) / GEMM_BK;
      const int64_t tilesW = (int64_t)(a.N / 256) * ceil_div64(a.rows, 128) * splits;
      const int64_t tiles128 = (int64_t)(a.N / 128) * ceil_div64(a.rows, 128) * splits;
      static const TileCost c256{1.42, 4.0, 1}, cW{1.32, 2.5, 1}, c128{1.30, 2.0, 2};
      const double t256 = gemm_tile_cost(tiles256, nk, c256), tW = gemm_tile_cost(tilesW, nk, cW), t128 = gemm_tile_cost(tiles128, nk, c128);
      choice = (t256 <= tW && t256 <= t128) ? 256 : (tW <= t128 ? 192 : 128);
    }
  }
  if (fits && a.tile_hint == 256) choice = 256;
  if (force128) choice = 128;
  if (choice == 256) return launch_gemm_t<EPI, Tile256>(a, st, prof_name);
  if constexpr (WIDE_OK) {
    if (choice == 192) return launch_gemm_t<EPI, TileWide>(a, st, prof_name);
  }
  return launch_gemm_t<EPI, Tile128>(a, st, prof_name);
}

}  // namespace convdr
