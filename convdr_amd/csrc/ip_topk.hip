// Exact brute-force inner-product top-k over a resident passage-embedding block (gfx950).
//
// Replaces faiss.IndexFlatIP.add/.search as driven by the reference at
//   /root/reference/drivers/run_convdr_inference.py:180-182  (gpu_index.add(block); D, I = gpu_index.search(Q, topN))
//
// Design (see DESIGN.md §"IP scan"): never materialise [nq, n].
//   prepare   fp32 block -> bf16 scan copy + max row norm                      (HBM-bound, once per .add)
//   sample    bf16 MFMA GEMM over ~1/32 of the block, epilogue keeps per-(query, 32 passages) top-2,
//             per-query radix select in LDS -> tau[q] ~ score of rank `rank_target` in the whole block
//   scan      persistent bf16 MFMA GEMM  P_bf16[n,d] * Q_bf16[nq,d]^T  (gemm_nt.hpp); epilogue: each lane (= query)
//             counts the accumulators >= tau[q], reserves that many slots of the query's candidate list with one
//             atomic and writes (index, score) of its hits
//   cut       per query: k-th largest scan score by radix select, band {S~ >= S~(k) - 2 eps} moved to the front,
//             certificate OK iff the list is complete down to the cut (cut >= tau, no overflow),
//             eps = 0.0079 * |q| * max|p - centre|   (bf16 rounding bound)
//   rescore   one wave per band candidate: fp64 dot of the fp32 originals, canonical order (oracle/search.py)
//   select    per query: the k best exact scores (radix select + bitonic sort of the survivors by
//             (exact score desc, index asc))
#include "gemm_nt.hpp"

#include <float.h>
#include <math.h>

#include "../../include/convdr_hip.h"

namespace convdr {

constexpr int IP_MODE_FULL = 0, IP_MODE_TOP2 = 1, IP_MODE_EMIT = 2;
constexpr int IP_FULL_MAX_N = 32768;      // <= this many passages: "sample" = all scores, exact rank select
constexpr int IP_SAMPLE_MIN = 32768;      // sampled passages (>= 1/32 of the block)
constexpr int IP_SAMPLE_MAX = 262144;
// Error-band coefficients: |scan score - exact score| <= coef * |q| * max|p - centre| for every passage of the block.
// u = 2^-8 bounds the relative error of a bf16 rounding, the accumulation term charges 2^-23 per accumulated product
// (gamma_K of an fp32 sum whose per-step rounding is at worst truncation: the MFMA's internal rounding is not
// documented; it sums 8-16 products per step, so this over-counts) relative to sum |q_i p_i| <= |q| |p|.  Computed from d
// (round 1 used constants that only covered d <= 1024 with the 2^-24 model).
//   bf16 scan:        (1 + u)^2 - 1 = 2u + u^2          + d * 2^-23
//   split-bf16 scan:  hi*hi + hi*lo + lo*hi leaves out lo*lo and the two second-rounding remainders,
//                     each <= u^2 (1 + u) |q| |p|         + 3 d * 2^-23 (one accumulator chain over the three passes)
//   fp16 scan:        same forms with u = 2^-11 (8x tighter: 1.07e-3 at d = 768 in one pass, 2.8e-4 split) PLUS an absolute
//                     term: below 2^-14 a half has no relative precision.  Both operands are scaled by powers of two
//                     (exact; per block / per query) so that their norms sit near 2^12, and an element is charged
//                     eta = 2^-14 of absolute error -- which also covers hardware that flushes subnormal MFMA inputs:
//                     |fl(q) fl(p) - q p| summed over d  <=  rel |q||p| + eta (1 + u) sqrt(d) (|q| + |p|) + d eta^2
//                     (ip_eps_abs; ~1e-6 of |q||p| at these scales).  The scan then works entirely in scaled units:
//                     thresholds, candidate scores and eps; the results come from the fp64 re-scoring of the originals.
constexpr int IP_KIND_BF16 = 0, IP_KIND_F16 = 1;
static inline float ip_eps_coef(int d, bool x3, int kind = IP_KIND_BF16) {
  const double u = kind == IP_KIND_F16 ? 1.0 / 2048.0 : 1.0 / 256.0, acc = (double)d / 8388608.0;
  const double c = x3 ? 3.0 * u * u * (1.0 + 2.0 * u) + 3.0 * acc : 2.0 * u + u * u + acc;
  return (float)(c * 1.0001);
}
constexpr double IP_F16_ETA = 1.0 / 16384.0;
static inline float ip_eps_abs(int d, bool x3, int kind) {   // multiplies (|q| + max|p|), scaled units
  if (kind != IP_KIND_F16) return 0.f;
  // split scan: hi and lo are both halfs; the remainder x - hi - lo carries at most eta as well
  return (float)((x3 ? 2.0 : 1.0) * IP_F16_ETA * (1.0 + 1.0 / 2048.0) * sqrt((double)d) * 1.0001);
}
constexpr float IP_F16_TARGET_EXP = 12.f;    // scales put the (first) max norm into [2^12, 2^13)
constexpr float IP_F16_NORM_LIMIT = 60000.f; // scaled norms above this may hold elements that round to inf: CONVDR_IP_RANGE

// ------------------------------------------------------------------------------------------
// rows fp32 -> bf16 of (x - centre) (+ per-row L2 norm of the centred row, + global max norm).
// Optional second output: the bf16 of the rounding remainder (x - centre) - hi, for the split-bf16 scan.
// One wave per row, float4 loads.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f16x2_t r = {(_Float16)lo, (_Float16)hi};   // v_cvt_pkrtz is NOT used: round to nearest even
  return *(const uint32_t*)&r;
}
template <int KIND>
__device__ __forceinline__ uint32_t pack_half2(float lo, float hi) {
  return KIND == IP_KIND_F16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi);
}
template <int KIND>
__device__ __forceinline__ float half_lo_to_f32(uint32_t packed) {
  if constexpr (KIND == IP_KIND_F16) { const uint16_t h = (uint16_t)packed; return (float)*(const _Float16*)&h; }
  return bf16_to_f32((bf16_t)packed);
}
template <int KIND>
__device__ __forceinline__ float half_hi_to_f32(uint32_t packed) {
  if constexpr (KIND == IP_KIND_F16) { const uint16_t h = (uint16_t)(packed >> 16); return (float)*(const _Float16*)&h; }
  return bf16_to_f32((bf16_t)(packed >> 16));
}

// PER_ROW_SCALE (queries of the fp16 scan): every row is multiplied by its own power of two 2^(12 - e), |row| = m 2^e
// with m in [0.5, 1), so its scaled norm lies in [2^11, 2^12); row_norm receives the SCALED norm.  Otherwise `scale`
// (a power of two; 1 for bf16) multiplies every row and row_norm / max_norm are UNSCALED.
template <int KIND, bool PER_ROW_SCALE>
__global__ void __launch_bounds__(256) k_rows_to_half(const float* __restrict__ X, int64_t n, int d,
                                                      const float* __restrict__ centre, float scale, bf16_t* __restrict__ Y,
                                                      bf16_t* __restrict__ Ylo, float* __restrict__ row_norm,
                                                      float* __restrict__ max_norm, int64_t n_pad = 0,
                                                      uint32_t* __restrict__ zero_u32 = nullptr, int64_t zero_count = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wmax = 0.f;
  // query preparation of a search: the padding rows [n, n_pad) of Y / Ylo and the candidate counters are zeroed here
  // instead of by three memsets in front of this kernel (a search of <= 128 queries is a chain of short launches)
  for (int64_t row = n + (int64_t)blockIdx.x * 4 + wave; row < n_pad; row += (int64_t)gridDim.x * 4)
    for (int e = lane * 4; e < d; e += 256) {
      *(uint2*)(Y + row * d + e) = make_uint2(0u, 0u);
      if (Ylo) *(uint2*)(Ylo + row * d + e) = make_uint2(0u, 0u);
    }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < zero_count; i += (int64_t)gridDim.x * 256) zero_u32[i] = 0u;
  // Norms are summed in fp64: eps is proportional to them, and in fp32 the squares of elements below ~1e-19 underflow
  // (a block of tiny-magnitude embeddings would get norm 0 = "no rounding error").  The pass is HBM-bound either way.
  auto sq4 = [](const float4& v) {
    return (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
  };
  auto wave_sum_f64 = [](double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
    const float* x = X + row * d;
    double ss = 0.0;
    float sc = scale;
    if constexpr (PER_ROW_SCALE) {
      for (int e = lane * 4; e < d; e += 256) ss += sq4(*(const float4*)(x + e));
      ss = wave_sum_f64(ss);
      int ex = 0;
      const float nm0 = (float)sqrt(ss);
      (void)frexpf(nm0, &ex);
      int sh = (int)IP_F16_TARGET_EXP - ex;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      sc = (nm0 > 0.f && nm0 < INFINITY) ? ldexpf(1.f, sh) : 1.f;
      ss = 0.0;
    }
    for (int e = lane * 4; e < d; e += 256) {
      float4 v = *(const float4*)(x + e);
      if (centre) {
        const float4 c = *(const float4*)(centre + e);
        v.x -= c.x; v.y -= c.y; v.z -= c.z; v.w -= c.w;
      }
      if constexpr (PER_ROW_SCALE) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
      ss += sq4(v);
      if (Y) {
        if constexpr (!PER_ROW_SCALE && KIND == IP_KIND_F16) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        uint2 o;
        o.x = pack_half2<KIND>(v.x, v.y);
        o.y = pack_half2<KIND>(v.z, v.w);
        *(uint2*)(Y + row * d + e) = o;
        if (Ylo) {
          uint2 l;
          l.x = pack_half2<KIND>(v.x - half_lo_to_f32<KIND>(o.x), v.y - half_hi_to_f32<KIND>(o.x));
          l.y = pack_half2<KIND>(v.z - half_lo_to_f32<KIND>(o.y), v.w - half_hi_to_f32<KIND>(o.y));
          *(uint2*)(Ylo + row * d + e) = l;
        }
      }
    }
    ss = wave_sum_f64(ss);
    const float nm = (float)(sqrt(ss) * (1.0 + 1e-6));   // (rounded up: a bound)
    if (row_norm && lane == 0) row_norm[row] = nm;
    wmax = fmaxf(wmax, nm);
  }
  if (max_norm) {
    __shared__ float sm[4];
    if (lane == 0) sm[wave] = wmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
      atomicMax((int*)max_norm, __float_as_int(m));  // non-negative floats order like ints
    }
  }
}

// column sums of fp32 rows: part[chunk][d]; grid (ceil(d / 256), chunks)
__global__ void __launch_bounds__(256) k_colsum_f32(const float* __restrict__ X, int64_t n, int d, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (n + gridDim.y - 1) / gridDim.y;
  const int64_t t0 = per * blockIdx.y, t1 = t0 + per < n ? t0 + per : n;
  if (c >= d) return;
  float s = 0.f;
  for (int64_t t = t0; t < t1; ++t) s += X[t * d + c];
  part[(int64_t)blockIdx.y * d + c] = s;
}
__global__ void __launch_bounds__(256) k_colmean_finish(const float* __restrict__ part, int chunks, int d, int64_t n,
                                                        float* __restrict__ mean) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  double s = 0.0;
  for (int p = 0; p < chunks; ++p) s += part[(int64_t)p * d + c];
  mean[c] = (float)(s / (double)n);
}

__global__ void k_fill_f32(float* p, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// The scan GEMM.  A = passages (rows -> accumulator rows), B = queries (-> accumulator columns,
// one query per lane), so the per-query threshold lives in one VGPR per 32-column MFMA tile.
// Grid: 1-D, nPt * nQt blocks, XCD-remapped so the nQt query tiles of one passage tile run
// back-to-back on one XCD (the passage tile is fetched from HBM once, then hits that XCD's L2).
// ------------------------------------------------------------------------------------------
// One hit counter per 128-byte line.  The returning atomics that reserve candidate slots execute at the memory side and
// serialise per LINE: with the counters packed 32 to a line, 1,000 queries shared 32 lines and the ~1.6 M reservations of
// a 1 M-passage scan were 30 % of its time; one line per query spreads them over every channel.
constexpr int IP_COUNT_STRIDE = 32;

struct ScanArgs {
  const bf16_t* P;   // [n, d]
  const bf16_t* Qb;  // [nq_pad, d] (rows >= nq are zero)
  const bf16_t* Plo; // split-bf16 scan: remainders (nullptr = plain bf16 scan)
  const bf16_t* Qlo;
  int64_t n;
  int nq, nq_pad, d;
  int nPt, nQt;      // tiles actually visited / query tiles
  int pt_stride;     // visited passage tile t -> tile t * pt_stride
  const float* tau;  // EMIT: [nq_pad]
  uint32_t* counts;  // EMIT: [nq_pad * IP_COUNT_STRIDE], counter of query q at q * IP_COUNT_STRIDE
  uint32_t* cand_id; // EMIT: [nq, cap]
  float* cand_s;     // EMIT: [nq, cap]
  int cap;
  float* T;          // FULL: [nPt*128, nq_pad]; TOP2: [nPt*8, nq_pad]
  int dbg_prelanded; // timing experiment (TRACE library only): a tile's first K chunks are not waited for (garbage results)
};

template <int MODE, class T>
__device__ __forceinline__ void scan_epilogue(const ScanArgs& a, GemmAcc<T>& acc, const WavePos<T>& w, int ts, int64_t m0,
                                              int64_t n0, const float (&tau_lane)[T::NT]) {
  if constexpr (MODE == IP_MODE_EMIT) {
    // A lane owns one query per 32-column tile and MT*16 of the tile's passages.  It counts its hits, reserves that
    // many slots of the query's candidate list with ONE atomic (a device-scope returning atomic is a ~1-2 us round
    // trip to the memory side; one per hit -- ~100 per tile, serialised by the branches around them -- was 30 % of
    // the scan), then writes the hits into consecutive slots.  The list order differs from run to run either way;
    // the downstream kernels do not depend on it.
    const int rows_left = (int)(a.n - m0 < (int64_t)T::TR ? a.n - m0 : (int64_t)T::TR);
    if (rows_left < T::TR) {   // the last passage tile: rows past n scored 0 (zero-filled operand) and must never hit
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (w.r_index(mt, r) >= rows_left) acc.c[mt][nt][r] = -INFINITY;
    }
    uint32_t cnt[T::NT], slot[T::NT];
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      uint32_t c = 0;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) c += acc.c[mt][nt][r] >= tau_lane[nt] ? 1u : 0u;
      cnt[nt] = c;
    }
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int q = (int)n0 + w.l_index(nt);
      slot[nt] = cnt[nt] ? atomicAdd(&a.counts[(int64_t)q * IP_COUNT_STRIDE], cnt[nt]) : 0u;
    }
    // hipcc: retire the reservations HERE, once.  Left to the compiler, every hit below starts with s_waitcnt vmcnt(0)
    // (the slot might still be pending on the path that skipped the previous hit), which on gfx9 also waits for the
    // previous hit's STORES: a dozen serialised write round trips per wave per tile.
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) asm volatile("" : "+v"(slot[nt]));
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      if (cnt[nt] == 0) continue;
      const int q = (int)n0 + w.l_index(nt);
      const float tau = tau_lane[nt];
      uint32_t* ids = a.cand_id + (int64_t)q * a.cap;
      float* sc = a.cand_s + (int64_t)q * a.cap;
      uint32_t sl = slot[nt];
      const uint32_t row0 = (uint32_t)m0;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt) {
        const f32x16 v = acc.c[mt][nt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (v[r] >= tau) {
            if (sl < (uint32_t)a.cap) {
              ids[sl] = row0 + (uint32_t)w.r_index(mt, r);
              sc[sl] = v[r];
            }
            ++sl;
          }
        }
      }
    }
  } else if constexpr (MODE == IP_MODE_FULL) {
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int q = (int)n0 + w.l_index(nt);
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + w.r_index(mt, r);
          a.T[row * a.nq_pad + q] = row < a.n ? acc.c[mt][nt][r] : -INFINITY;
        }
    }
  } else {  // TOP2: best two of this lane's MT*16 scores (one query, MT*16 of the tile's passages)
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int q = (int)n0 + w.l_index(nt);
      float b0 = -INFINITY, b1 = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + w.r_index(mt, r);
          const float s = row < a.n ? acc.c[mt][nt][r] : -INFINITY;
          const float lo = fminf(b0, s);
          b0 = fmaxf(b0, s);
          b1 = fmaxf(b1, lo);
        }
      const int64_t slot = (((int64_t)ts * T::WR + w.wr) * 2 + w.hi) * 2;
      a.T[slot * a.nq_pad + q] = b0;
      a.T[(slot + 1) * a.nq_pad + q] = b1;
    }
  }
}

// Persistent: one workgroup per resident slot walks a strided sequence of tiles (XCD x owns a contiguous chunk of the
// tile order, query tile fastest); the first K chunk of the next tile streams into the idle operand stage while the
// epilogue of this one runs, so neither the workgroup dispatch gap nor the cold HBM round trip for a fresh passage
// tile is exposed (they were ~40 % of a 12-step tile).
template <int MODE, class T, bool X3, bool F16>
__global__ void __launch_bounds__(T::THREADS, 2) k_ip_scan(const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t ntiles = (uint32_t)a.nPt * (uint32_t)a.nQt;
  const uint32_t xcd = blockIdx.x & 7u, q8 = ntiles >> 3, r8 = ntiles & 7u;
  const uint32_t chunk_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const uint32_t chunk_len = q8 + (xcd < r8 ? 1u : 0u);
  const uint32_t stride = (gridDim.x + 7u) >> 3;
  uint32_t idx = blockIdx.x >> 3;
  if (idx >= chunk_len) return;
  const WavePos<T> w;
  auto coords = [&](uint32_t i, int& ts, int64_t& m0, int64_t& n0) {
    const uint32_t logical = chunk_base + i;
    ts = (int)(logical / a.nQt);
    const int qt = (int)(logical - (uint32_t)ts * a.nQt);
    m0 = (int64_t)ts * a.pt_stride * T::TR;
    n0 = (int64_t)qt * T::TL;
  };
  int ts;
  int64_t m0, n0;
  coords(idx, ts, m0, n0);
  TileSrc<T> src(a.P, a.d, a.n, a.Qb, a.d, a.nq_pad, m0, n0, w);
  int buf = 0;
  gemm_issue_stage<T>(src, 0, smem + buf * T::STAGE_BYTES, w);
  // A tile's thresholds are loaded one tile ahead (loop-carried, so the load cannot be sunk to its use after the main
  // loop, where its full latency would be exposed) and are retired by the main loop's own waits; an ordinary load used
  // while the next tile's LDS-DMA is in flight would drain that too.
  float tau_next[T::NT];
  auto load_tau = [&](int64_t n0_) {
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int q = (int)n0_ + w.l_index(nt);
      tau_next[nt] = (MODE == IP_MODE_EMIT && q < a.nq) ? a.tau[q] : INFINITY;
    }
  };
  load_tau(n0);
  for (;;) {
    GemmAcc<T> acc;
    acc.zero();
    float tau_lane[T::NT];
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) tau_lane[nt] = tau_next[nt];
    int idle = gemm_nt_mainloop<T, F16>(src, a.d, smem, acc, w, buf, true);
    if constexpr (X3) {  // S~ = Ph Qh + Ph Ql + Pl Qh: fp32-class scores from three bf16 passes into the same accumulators
      __syncthreads();
      const TileSrc<T> s2(a.P, a.d, a.n, a.Qlo, a.d, a.nq_pad, m0, n0, w);
      idle = gemm_nt_mainloop<T, F16>(s2, a.d, smem, acc, w, idle);
      __syncthreads();
      const TileSrc<T> s3(a.Plo, a.d, a.n, a.Qb, a.d, a.nq_pad, m0, n0, w);
      idle = gemm_nt_mainloop<T, F16>(s3, a.d, smem, acc, w, idle);
    }
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) asm volatile("" : "+v"(tau_lane[nt]));   // hipcc: the loads are retired HERE
    const uint32_t next = idx + stride;
    const bool has_next = next < chunk_len;
    const int ts_cur = ts;
    const int64_t m0_cur = m0, n0_cur = n0;
    if (has_next) {
      coords(next, ts, m0, n0);
      load_tau(n0);
      src = TileSrc<T>(a.P, a.d, a.n, a.Qb, a.d, a.nq_pad, m0, n0, w);
      gemm_issue_stage<T>(src, 0, smem + idle * T::STAGE_BYTES, w);
    }
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));   // opaque: the epilogue's lane-dependent addresses stay out of the main loop's registers
    const WavePos<T> we(tid_e);
    scan_epilogue<MODE, T>(a, acc, we, ts_cur, m0_cur, n0_cur, tau_lane);
    if (!has_next) break;
    idx = next;
    buf = idle;
  }
}

// Emission with the slot reservation off the critical path (k_ip_scan_r3).  scan_epilogue<EMIT> counts a lane's hits,
// reserves their list slots with a returning atomic and has to WAIT for it (~1.5 us of a ~20 us tile, plus the counting
// sweep) before it can write them.  Here one sweep both counts and keeps a lane's first two hits in registers (hits
// are ~0.2 per lane and tile; a lane with more than two -- one in several thousand -- takes the old reserve-and-wait
// route for the rest), the reservation is issued and NOT waited for, and the two hits are written after the NEXT
// tile's main loop, by which time it has long returned.
template <class T>
struct EmitPending {
  uint32_t cnt[T::NT], slot[T::NT], id0[T::NT], id1[T::NT];
  float s0[T::NT], s1[T::NT];
  int64_t n0;
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) { cnt[nt] = 0; slot[nt] = 0; id0[nt] = id1[nt] = 0; s0[nt] = s1[nt] = 0.f; }
    n0 = 0;
  }
};

template <class T>
__device__ __forceinline__ void scan_emit_capture(const ScanArgs& a, GemmAcc<T>& acc, const WavePos<T>& w, int64_t m0, int64_t n0,
                                                  const float (&tau_lane)[T::NT], EmitPending<T>& pd) {
  const int rows_left = (int)(a.n - m0 < (int64_t)T::TR ? a.n - m0 : (int64_t)T::TR);
  if (rows_left < T::TR) {   // the last passage tile: rows past n scored 0 (zero-filled operand) and must never hit
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (w.r_index(mt, r) >= rows_left) acc.c[mt][nt][r] = -INFINITY;
  }
  const uint32_t row0 = (uint32_t)m0;
  pd.n0 = n0;
  // per accumulator register: one compare and one wave-uniform branch when no lane hits (9 registers in 10); the
  // capture itself is predicated, not branched (nested divergent branches cost hipcc an exec-mask stack in SGPR spills)
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const float tau = tau_lane[nt];
    uint32_t c = 0, i0 = 0, i1 = 0;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) {
      const f32x16 v = acc.c[mt][nt];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool hit = v[r] >= tau;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit) != 0ull, 0)) {
          const uint32_t id = row0 + (uint32_t)w.r_index(mt, r);
          const bool f0 = hit && c == 0, f1 = hit && c == 1;
          s0 = f0 ? v[r] : s0; i0 = f0 ? id : i0;
          s1 = f1 ? v[r] : s1; i1 = f1 ? id : i1;
          c += hit ? 1u : 0u;
        }
      }
    }
    pd.cnt[nt] = c; pd.id0[nt] = i0; pd.id1[nt] = i1; pd.s0[nt] = s0; pd.s1[nt] = s1;
  }
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const int q = (int)n0 + w.l_index(nt);
    pd.slot[nt] = pd.cnt[nt] ? atomicAdd(&a.counts[(int64_t)q * IP_COUNT_STRIDE], pd.cnt[nt]) : 0u;
  }
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(pd.cnt[nt] > 2) == 0ull, 1)) continue;   // no lane of the wave overflowed
    asm volatile("" : "+v"(pd.slot[nt]));   // hipcc: wait for the reservation once, not in front of every store
    const int q = (int)n0 + w.l_index(nt);
    float tau = tau_lane[nt];
    asm volatile("" : "+v"(tau));   // opaque copy: or hipcc keeps all 128 compare masks of the sweep above alive (in spilled SGPRs) for this one
    uint32_t* ids = a.cand_id + (int64_t)q * a.cap;
    float* sc = a.cand_s + (int64_t)q * a.cap;
    uint32_t k = 0;
    const uint32_t base = pd.slot[nt];
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) {
      const f32x16 v = acc.c[mt][nt];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool hit = v[r] >= tau;
        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
          if (hit && k >= 2 && base + k < (uint32_t)a.cap) {
            ids[base + k] = row0 + (uint32_t)w.r_index(mt, r);
            sc[base + k] = v[r];
          }
          k += hit ? 1u : 0u;
        }
      }
    }
  }
}

template <class T>
__device__ __forceinline__ void scan_emit_flush(const ScanArgs& a, const WavePos<T>& w, EmitPending<T>& pd) {
  // hipcc: retire the reservations HERE, once (nothing else is in flight at the two call sites)
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) asm volatile("" : "+v"(pd.slot[nt]));
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    if (pd.cnt[nt] == 0) continue;
    const int q = (int)pd.n0 + w.l_index(nt);
    uint32_t* ids = a.cand_id + (int64_t)q * a.cap;
    float* sc = a.cand_s + (int64_t)q * a.cap;
    const uint32_t sl = pd.slot[nt];
    if (sl < (uint32_t)a.cap) { ids[sl] = pd.id0[nt]; sc[sl] = pd.s0[nt]; }
    if (pd.cnt[nt] > 1 && sl + 1 < (uint32_t)a.cap) { ids[sl + 1] = pd.id1[nt]; sc[sl + 1] = pd.s1[nt]; }
    pd.cnt[nt] = 0;
  }
}

// The emitting scan of 256 x 256 tiles on the 3 R-slot / 2 L-slot main loop (gemm_nt_mainloop_r3); same persistent walk,
// next-tile prologue under the epilogue and one-tile-ahead thresholds as k_ip_scan.  (The split-bf16 scan and the
// sampling modes stay on the two-stage loop.)
template <class T, bool F16>
__global__ void __launch_bounds__(T::THREADS, 2) k_ip_scan_r3(const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the 256 x 128 tile serves scans with ONE query tile (at most 128 queries): every passage chunk is read exactly once,
  // so its DMA is non-temporal -- the block streams past L2 instead of through it
  constexpr int P_AUX = T::TL <= 128 ? 2 : 0;
  const uint32_t ntiles = (uint32_t)a.nPt * (uint32_t)a.nQt;
  const uint32_t xcd = blockIdx.x & 7u, q8 = ntiles >> 3, r8 = ntiles & 7u;
  const uint32_t chunk_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const uint32_t chunk_len = q8 + (xcd < r8 ? 1u : 0u);
  const uint32_t stride = (gridDim.x + 7u) >> 3;
  uint32_t idx = blockIdx.x >> 3;
  if (idx >= chunk_len) return;
  const WavePos<T> w;
  auto coords = [&](uint32_t i, int& ts, int64_t& m0, int64_t& n0) {
    const uint32_t logical = chunk_base + i;
    ts = (int)(logical / a.nQt);
    const int qt = (int)(logical - (uint32_t)ts * a.nQt);
    m0 = (int64_t)ts * a.pt_stride * T::TR;
    n0 = (int64_t)qt * T::TL;
  };
  int ts;
  int64_t m0, n0;
  coords(idx, ts, m0, n0);
  TileSrcAll<T> src(a.P, a.d, a.n, a.Qb, a.d, a.nq_pad, m0, n0, w);
  R3Slots slots{0, 0};
  gemm_r3_prologue<T, P_AUX>(src, a.d, smem, w, slots);
  float tau_next[T::NT];
  auto load_tau = [&](int64_t n0_) {
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int q = (int)n0_ + w.l_index(nt);
      tau_next[nt] = q < a.nq ? a.tau[q] : INFINITY;
    }
  };
  load_tau(n0);
  EmitPending<T> pend;
  pend.clear();
  for (;;) {
    GemmAcc<T> acc;
    acc.zero();
    float tau_lane[T::NT];
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) tau_lane[nt] = tau_next[nt];
    slots = gemm_nt_mainloop_r3<T, F16, P_AUX>(src, a.d, smem, acc, w, slots, true, false, a.dbg_prelanded != 0);
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) asm volatile("" : "+v"(tau_lane[nt]));
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const WavePos<T> we(tid_e);
    scan_emit_flush<T>(a, we, pend);   // the previous tile's hits: their reservations came back under this main loop
    const uint32_t next = idx + stride;
    const bool has_next = next < chunk_len;
    const int64_t m0_cur = m0, n0_cur = n0;
    if (has_next) {
      coords(next, ts, m0, n0);
      load_tau(n0);
      src = TileSrcAll<T>(a.P, a.d, a.n, a.Qb, a.d, a.nq_pad, m0, n0, w);
      gemm_r3_prologue<T, P_AUX>(src, a.d, smem, w, slots);
    }
    scan_emit_capture<T>(a, acc, we, m0_cur, n0_cur, tau_lane, pend);
    if (!has_next) break;
    idx = next;
  }
  scan_emit_flush<T>(a, w, pend);
}

// ------------------------------------------------------------------------------------------
// Block-wide order statistics in LDS.  The search needs three of them per query -- the r-th largest sample score (the
// threshold), the k-th largest candidate score (the band cut) and the k best re-scored candidates -- and none needs
// the whole list ordered: an MSB-first radix select (one 256-bin histogram per key byte) finds the k-th largest key
// in a handful of barriers where the full bitonic sorts these replaced took 55-78 barrier-separated stages.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_order_key(float f) {   // a > b  <=>  key(a) > key(b)   (-0 orders below +0)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t key) {
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}
__device__ __forceinline__ uint64_t f64_order_key(double f) {
  const uint64_t u = (uint64_t)__double_as_longlong(f);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

struct SelectScratch {   // static LDS of a selecting kernel
  uint32_t hist[256];
  uint32_t wave_sum[4];
  uint32_t bin, rem;
};

// kth-largest (kth = 1: the maximum; 1 <= kth <= c) of key_at(0..c-1); NBITS = significant key bits (32 or 64).
// Every thread returns the key.  blockDim.x >= 256.
template <int NBITS, class KeyAt>
__device__ uint64_t block_kth_largest(KeyAt key_at, int c, uint32_t kth, SelectScratch& sc) {
  uint64_t prefix = 0, mask = 0;
  uint32_t rem = kth;
  for (int shift = NBITS - 8; shift >= 0; shift -= 8) {
    if (threadIdx.x < 256) sc.hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
      const uint64_t x = key_at(i);
      if ((x & mask) == prefix) atomicAdd(&sc.hist[(uint32_t)(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    // the bin holding the rem-th largest of the keys that match the prefix: suffix sums of the histogram, one bin
    // per thread of the first four waves (shuffle scan inside a wave, the four wave totals through LDS)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t here = 0, incl = 0;
    if (threadIdx.x < 256) {
      here = incl = sc.hist[threadIdx.x];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += v;
      }
      if (lane == 0) sc.wave_sum[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      uint32_t above = incl - here;
      for (int v = wave + 1; v < 4; ++v) above += sc.wave_sum[v];
      if (above < rem && rem <= above + here) { sc.bin = threadIdx.x; sc.rem = rem - above; }
    }
    __syncthreads();
    prefix |= (uint64_t)sc.bin << shift;
    mask |= (uint64_t)255 << shift;
    rem = sc.rem;
  }
  return prefix;
}

__device__ __forceinline__ bool cand_before(double sa, uint32_t ia, double sb, uint32_t ib) {
  return sa > sb || (sa == sb && ia < ib);
}

// LDS bitonic sort by (score desc, index asc); n a power of two
__device__ void bitonic_cand(double* s, uint32_t* id, int n) {
  for (int k2 = 2; k2 <= n; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const double x = s[i], y = s[p];
          const uint32_t ix = id[i], iy = id[p];
          const bool fwd = (i & k2) == 0;
          const bool sw = fwd ? cand_before(y, iy, x, ix) : cand_before(x, ix, y, iy);
          if (sw) { s[i] = y; s[p] = x; id[i] = iy; id[p] = ix; }
        }
      }
      __syncthreads();
    }
}

// tau[q] = r-th largest of T[0..nvals) for query q (column q of T, leading dim nq_pad)
__global__ void __launch_bounds__(1024) k_tau_select(const float* __restrict__ T, int64_t nvals, int nq_pad, int r,
                                                     float* __restrict__ tau) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ SelectScratch sc;
  uint32_t* key = (uint32_t*)smem;
  const int q = blockIdx.x;
  for (int i = threadIdx.x; i < nvals; i += blockDim.x) key[i] = f32_order_key(T[(int64_t)i * nq_pad + q]);
  __syncthreads();
  if (r < 1 || r > nvals) {
    if (threadIdx.x == 0) tau[q] = -INFINITY;
    return;
  }
  const uint32_t kk = (uint32_t)block_kth_largest<32>([&](int i) { return (uint64_t)key[i]; }, (int)nvals, (uint32_t)r, sc);
  if (threadIdx.x == 0) tau[q] = f32_from_order_key(kk);
}

// ------------------------------------------------------------------------------------------
// band cut: which candidates must be re-scored exactly?
// Let s~(k) be the k-th best bf16-pass score of the query and eps its error bound
// (|s~ - exact| <= eps for every passage of the block).  The k best by s~ all have exact >= s~(k) - eps,
// so the true k-th exact score t_k >= s~(k) - eps.  A passage with s~ < cut := s~(k) - 2 eps has
// exact < s~(k) - eps <= t_k and cannot be in the true top-k.  Hence re-scoring exactly the band
// {s~ >= cut} yields the exact top-k, PROVIDED the candidate list is complete down to cut, i.e.
// cut >= tau and the list did not overflow.  Otherwise the query is flagged for a retry with a
// threshold that makes the next pass complete.
// Moves the band to the front of the query's candidate list (in no particular order: k_ip_select orders the
// re-scored band by (exact score, index), which does not depend on it) and writes m[q] = band size.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_ip_cut(int64_t n, int k, int cap, const uint32_t* __restrict__ counts,
                                                uint32_t* __restrict__ counts_packed,
                                                uint32_t* __restrict__ cand_id, float* __restrict__ cand_s,
                                                const float* __restrict__ tau, const float* __restrict__ qnorm,
                                                const float* __restrict__ p_max_norm, float eps_coef, float eps_abs,
                                                float p_scale, float norm_limit,
                                                uint32_t* __restrict__ m_out, int32_t* __restrict__ status,
                                                float* __restrict__ tau_retry) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ SelectScratch sc;
  __shared__ uint32_t sh_m;
  const int q = blockIdx.x;
  const uint32_t cnt = counts[(int64_t)q * IP_COUNT_STRIDE];
  if (threadIdx.x == 0) counts_packed[q] = cnt;   // [nq] copy for convdr_ip_debug_counts
  const int c = cnt < (uint32_t)cap ? (int)cnt : cap;
  uint32_t* key = (uint32_t*)smem;
  uint32_t* id = (uint32_t*)(smem + (size_t)cap * 4);
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    key[i] = f32_order_key(cand_s[(int64_t)q * cap + i]);
    id[i] = cand_id[(int64_t)q * cap + i];
  }
  if (threadIdx.x == 0) sh_m = 0;
  __syncthreads();
  const int need = (int64_t)k < n ? k : (int)n;
  const float t = tau[q];
  // (scaled units for the fp16 scan: qnorm is the scaled query norm, pm the scaled largest passage norm)
  const float pm = p_max_norm[0] * p_scale;
  const float eps = (eps_coef * qnorm[q] * pm + eps_abs * (qnorm[q] + pm) + eps_abs * eps_abs) * 1.001f + 1e-30f;
  const bool have_k = need > 0 && c >= need;
  float cut = -INFINITY;
  if (have_k)
    cut = f32_from_order_key((uint32_t)block_kth_largest<32>([&](int i) { return (uint64_t)key[i]; }, c, (uint32_t)need, sc)) -
          2.f * eps;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    const float v = f32_from_order_key(key[i]);
    if (v >= cut) {
      const uint32_t slot = atomicAdd(&sh_m, 1u);
      cand_id[(int64_t)q * cap + slot] = id[i];
      cand_s[(int64_t)q * cap + slot] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int st = CONVDR_IP_OK;
    float retry = -INFINITY;
    if (cnt > (uint32_t)cap) {
      // stored candidates are a subset of {s~ >= tau}: their k-th s~ bounds the true s~(k) from below,
      // so `cut` is a valid (if loose) threshold; when it does not tighten tau the caller raises cap
      st = CONVDR_IP_OVERFLOW;
      const float cand = nextafterf(cut, -INFINITY);
      retry = cand > t ? cand : t;
    } else if (c < need) {
      st = CONVDR_IP_TOO_FEW;
      retry = t - 4.f * eps - 1e-3f * fabsf(t);
    } else if (need > 0 && t > -INFINITY && cut < t) {
      st = CONVDR_IP_UNCERTAIN;  // band reaches below tau: list incomplete in [cut, tau)
      retry = nextafterf(cut, -INFINITY);
    }
    // fp16 scan copy built with a scale too large for this block's norms -- or a query so long that its per-row power-of-two
    // scale hit the clamp (norm > ~8e34) --: elements of an operand may be inf
    if (pm > norm_limit || qnorm[q] > norm_limit) {
      st = CONVDR_IP_RANGE;
      retry = -INFINITY;
    }
    m_out[q] = sh_m;
    status[q] = st;
    tau_retry[q] = retry;
  }
}

// ------------------------------------------------------------------------------------------
// exact rescoring: canonical fp64 inner product (see oracle/search.py: lane l owns elements
// 256 j + 4 l + c, accumulated in (j, c) order; then butterfly 32,16,8,4,2,1)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ip_rescore(const float* __restrict__ Q, const float* __restrict__ P, int d,
                                                    int cap, const uint32_t* __restrict__ m_in,
                                                    const uint32_t* __restrict__ cand_id,
                                                    double* __restrict__ cand_x) {
  const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c = m_in[q];
  const float* qv = Q + (int64_t)q * d;
  for (uint32_t slot = blockIdx.y * 4 + wave; slot < c; slot += gridDim.y * 4) {
    const uint32_t id = cand_id[(int64_t)q * cap + slot];
    const float* pv = P + (int64_t)id * d;
    double acc = 0.0;
    for (int e = lane * 4; e < d; e += 256) {
      const float4 x = *(const float4*)(qv + e);
      const float4 y = *(const float4*)(pv + e);
      acc = fma((double)x.x, (double)y.x, acc);
      acc = fma((double)x.y, (double)y.y, acc);
      acc = fma((double)x.z, (double)y.z, acc);
      acc = fma((double)x.w, (double)y.w, acc);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) cand_x[(int64_t)q * cap + slot] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// per-query top-k of the re-scored band by (exact score desc, index asc): select the k-th largest exact score, keep
// the candidates at or above it (more than k only when scores tie exactly at the boundary -- duplicate passages), and
// sort just those.
// ------------------------------------------------------------------------------------------
constexpr int IP_SELECT_THREADS = 1024;
constexpr int IP_SELECT_PER_THREAD = 8192 / IP_SELECT_THREADS;   // cap <= 8192
__global__ void __launch_bounds__(IP_SELECT_THREADS) k_ip_select(int k, int cap, const uint32_t* __restrict__ m_in,
                                                                 const uint32_t* __restrict__ cand_id,
                                                                 const double* __restrict__ cand_x,
                                                                 float* __restrict__ D, int64_t* __restrict__ I) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ SelectScratch sc;
  __shared__ uint32_t sh_g;
  const int q = blockIdx.x;
  const int c = (int)m_in[q];
  double* s = (double*)smem;
  uint32_t* id = (uint32_t*)(smem + (size_t)cap * 8);
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    s[i] = cand_x[(int64_t)q * cap + i];
    id[i] = cand_id[(int64_t)q * cap + i];
  }
  if (threadIdx.x == 0) sh_g = 0;
  __syncthreads();
  int g = c;
  if (c > k) {
    // Select on the high 32 key bits only (sign, exponent, 20 mantissa bits): a candidate whose high word is below the
    // k-th largest high word is below at least k candidates in the full order too, so the top k all survive; the few
    // extra survivors that share the boundary word are ordered exactly by the sort.
    const uint64_t kth =
        block_kth_largest<32>([&](int i) { return f64_order_key(s[i]) >> 32; }, c, (uint32_t)k, sc);
    // compact {high word >= kth} to the front: every thread lifts its elements into registers first
    double ms[IP_SELECT_PER_THREAD];
    uint32_t mi[IP_SELECT_PER_THREAD];
#pragma unroll
    for (int e = 0; e < IP_SELECT_PER_THREAD; ++e) {
      const int i = threadIdx.x + e * IP_SELECT_THREADS;
      ms[e] = i < c ? s[i] : 0.0;
      mi[e] = i < c ? id[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < IP_SELECT_PER_THREAD; ++e) {
      const int i = threadIdx.x + e * IP_SELECT_THREADS;
      if (i < c && (f64_order_key(ms[e]) >> 32) >= kth) {
        const uint32_t slot = atomicAdd(&sh_g, 1u);
        s[slot] = ms[e];
        id[slot] = mi[e];
      }
    }
    __syncthreads();
    g = (int)sh_g;
  }
  int np2 = 2;
  while (np2 < g) np2 <<= 1;
  for (int i = g + threadIdx.x; i < np2; i += blockDim.x) { s[i] = -INFINITY; id[i] = 0xffffffffu; }
  __syncthreads();
  bitonic_cand(s, id, np2);
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    D[(int64_t)q * k + j] = j < g ? (float)s[j] : -FLT_MAX;
    I[(int64_t)q * k + j] = j < g ? (int64_t)id[j] : -1;
  }
}

// ------------------------------------------------------------------------------------------
// Round 5: cut + re-score + select of ONE query in ONE workgroup (k_ip_cut, k_ip_rescore and k_ip_select chained three
// launches per search with the band making a round trip through HBM between them; at 100 queries the three were 39 us of
// kernel time plus their boundaries around a 290 us scan).  Same arithmetic, phase by phase:
//   1. the query's candidate list into LDS, radix select of S~(k), cut = S~(k) - 2 eps, certificate (status, retry
//      threshold), the band compacted IN LDS (ids);
//   2. canonical fp64 re-score of the band: one wave per candidate, sixteen waves per query, each wave's next candidate's
//      loads in flight while it folds the current one (the same lane / butterfly order as k_ip_rescore: bit-identical);
//   3. radix select of the k-th exact score on the high key word, bitonic sort of the survivors, D / I written.
// LDS: cap x (4 id + 4 key) for phase 1, the key half reused with the id half's neighbour as cap x 8 scores from phase 2 on
// -> cap x 16 bytes (64 KB at the default cap of 4096: two workgroups per CU).  The per-query band size and packed candidate
// count are still written (convdr_ip_debug_*).  convdr_set_option("ip_fused_finish", 0) gives the three launches back.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_ip_finish(int64_t n, int k, int cap, const uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ counts_packed,
                                                   const uint32_t* __restrict__ cand_id, const float* __restrict__ cand_s,
                                                   const float* __restrict__ tau, const float* __restrict__ qnorm,
                                                   const float* __restrict__ p_max_norm, float eps_coef, float eps_abs,
                                                   float p_scale, float norm_limit, const float* __restrict__ Q,
                                                   const float* __restrict__ P, int d, uint32_t* __restrict__ m_out,
                                                   int32_t* __restrict__ status, float* __restrict__ tau_retry,
                                                   float* __restrict__ D, int64_t* __restrict__ I) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ SelectScratch sc;
  __shared__ uint32_t sh_m, sh_g;
  const int q = blockIdx.x;
  const uint32_t cnt = counts[(int64_t)q * IP_COUNT_STRIDE];
  if (threadIdx.x == 0) counts_packed[q] = cnt;
  const int c = cnt < (uint32_t)cap ? (int)cnt : cap;
  double* sx = (double*)smem;                                   // [cap] exact scores (phases 2, 3)
  uint32_t* key = (uint32_t*)smem;                              // [cap] order keys of the scan scores (phase 1; dead before sx is written)
  uint32_t* id = (uint32_t*)(smem + (size_t)cap * 8);           // [cap] candidate ids, then the band's ids
  uint32_t* bid = (uint32_t*)(smem + (size_t)cap * 12);         // [cap] the band, compacted
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    key[i] = f32_order_key(cand_s[(int64_t)q * cap + i]);
    id[i] = cand_id[(int64_t)q * cap + i];
  }
  if (threadIdx.x == 0) { sh_m = 0; sh_g = 0; }
  __syncthreads();
  // ---- phase 1: cut (k_ip_cut) ----
  const int need = (int64_t)k < n ? k : (int)n;
  const float t = tau[q];
  const float pm = p_max_norm[0] * p_scale;
  const float eps = (eps_coef * qnorm[q] * pm + eps_abs * (qnorm[q] + pm) + eps_abs * eps_abs) * 1.001f + 1e-30f;
  const bool have_k = need > 0 && c >= need;
  float cut = -INFINITY;
  if (have_k)
    cut = f32_from_order_key((uint32_t)block_kth_largest<32>([&](int i) { return (uint64_t)key[i]; }, c, (uint32_t)need, sc)) -
          2.f * eps;
  for (int i = threadIdx.x; i < c; i += blockDim.x)
    if (f32_from_order_key(key[i]) >= cut) bid[atomicAdd(&sh_m, 1u)] = id[i];
  __syncthreads();
  const int m = (int)sh_m;
  if (threadIdx.x == 0) {
    int st = CONVDR_IP_OK;
    float retry = -INFINITY;
    if (cnt > (uint32_t)cap) {
      st = CONVDR_IP_OVERFLOW;
      const float cand = nextafterf(cut, -INFINITY);
      retry = cand > t ? cand : t;
    } else if (c < need) {
      st = CONVDR_IP_TOO_FEW;
      retry = t - 4.f * eps - 1e-3f * fabsf(t);
    } else if (need > 0 && t > -INFINITY && cut < t) {
      st = CONVDR_IP_UNCERTAIN;
      retry = nextafterf(cut, -INFINITY);
    }
    if (pm > norm_limit || qnorm[q] > norm_limit) {
      st = CONVDR_IP_RANGE;
      retry = -INFINITY;
    }
    m_out[q] = (uint32_t)m;
    status[q] = st;
    tau_retry[q] = retry;
  }
  // ---- phase 2: canonical fp64 re-score (k_ip_rescore), one wave per band candidate ----
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* qv = Q + (int64_t)q * d;
    for (int slot = wave; slot < m; slot += nw) {
      const float* pv = P + (int64_t)bid[slot] * d;
      double acc = 0.0;
      for (int e = lane * 4; e < d; e += 256) {
        const float4 x = *(const float4*)(qv + e);
        const float4 y = *(const float4*)(pv + e);
        acc = fma((double)x.x, (double)y.x, acc);
        acc = fma((double)x.y, (double)y.y, acc);
        acc = fma((double)x.z, (double)y.z, acc);
        acc = fma((double)x.w, (double)y.w, acc);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (lane == 0) sx[slot] = acc;     // (key[] is dead: every thread passed the barrier after the compaction)
    }
  }
  __syncthreads();
  // ---- phase 3: exact top-k of the band (k_ip_select); id[] now holds the band ----
  for (int i = threadIdx.x; i < m; i += blockDim.x) id[i] = bid[i];
  __syncthreads();
  int g = m;
  if (m > k) {
    const uint64_t kth = block_kth_largest<32>([&](int i) { return f64_order_key(sx[i]) >> 32; }, m, (uint32_t)k, sc);
    double ms[IP_SELECT_PER_THREAD];
    uint32_t mi[IP_SELECT_PER_THREAD];
#pragma unroll
    for (int e = 0; e < IP_SELECT_PER_THREAD; ++e) {
      const int i = threadIdx.x + e * IP_SELECT_THREADS;
      ms[e] = i < m ? sx[i] : 0.0;
      mi[e] = i < m ? id[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < IP_SELECT_PER_THREAD; ++e) {
      const int i = threadIdx.x + e * IP_SELECT_THREADS;
      if (i < m && (f64_order_key(ms[e]) >> 32) >= kth) {
        const uint32_t slot = atomicAdd(&sh_g, 1u);
        sx[slot] = ms[e];
        id[slot] = mi[e];
      }
    }
    __syncthreads();
    g = (int)sh_g;
  }
  int np2 = 2;
  while (np2 < g) np2 <<= 1;
  for (int i = g + threadIdx.x; i < np2; i += blockDim.x) { sx[i] = -INFINITY; id[i] = 0xffffffffu; }
  __syncthreads();
  bitonic_cand(sx, id, np2);
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    D[(int64_t)q * k + j] = j < g ? (float)sx[j] : -FLT_MAX;
    I[(int64_t)q * k + j] = j < g ? (int64_t)id[j] : -1;
  }
}

// ------------------------------------------------------------------------------------------
// host-side plan shared by workspace sizing and the search call
// ------------------------------------------------------------------------------------------
constexpr int IP_TILE_128 = 0, IP_TILE_256 = 1, IP_TILE_TALL = 2;
struct IpPlan {
  int big;           // scan tile class: IP_TILE_256 (more than 128 queries), IP_TILE_TALL (256 passages x 128 queries: the
                     // HBM-bound regime), IP_TILE_128 (CONVDR_DBG_SCAN_TILE128: the round-1/2 small tile)
  int tr, tl;        // tile extent over passages / queries
  int nq_pad, nQt, nPt;
  int mode;          // -1: no threshold pass (n <= cap), else IP_MODE_FULL / IP_MODE_TOP2
  int nSt, stride;   // sampled passage tiles / tile stride
  int64_t nvals;     // values per query handed to k_tau_select
  int npow2;
  size_t o_qb, o_qlo, o_qnorm, o_tau, o_counts, o_counts_packed, o_m, o_T, o_id, o_s, o_x, total;
};

int64_t g_ip_fused_finish = 1;   // convdr_set_option("ip_fused_finish"): 1 = k_ip_finish, 0 = k_ip_cut + k_ip_rescore + k_ip_select

static IpPlan ip_plan(int nq, int64_t n, int d, int k, int cap) {
  IpPlan p;
  static const bool dbg_small = getenv("CONVDR_DBG_SCAN_TILE128") != nullptr;
  p.big = dbg_small ? IP_TILE_128 : (nq > 128 ? IP_TILE_256 : IP_TILE_TALL);
  p.tr = p.big == IP_TILE_128 ? Tile128::TR : Tile256::TR;
  p.tl = p.big == IP_TILE_256 ? Tile256::TL : Tile128::TL;
  p.nq_pad = (nq + p.tl - 1) / p.tl * p.tl;
  p.nQt = p.nq_pad / p.tl;
  p.nPt = (int)ceil_div64(n, p.tr);
  p.nSt = 0; p.stride = 1; p.nvals = 0; p.npow2 = 2;
  if (n <= cap) {
    p.mode = -1;
  } else if (n <= IP_FULL_MAX_N) {
    p.mode = IP_MODE_FULL; p.nSt = p.nPt; p.nvals = n;
  } else {
    p.mode = IP_MODE_TOP2;
    int64_t S = n / 32;
    if (S < IP_SAMPLE_MIN) S = IP_SAMPLE_MIN;
    if (S > IP_SAMPLE_MAX) S = IP_SAMPLE_MAX;
    p.nSt = (int)(S / p.tr);
    if (p.nSt > p.nPt) p.nSt = p.nPt;
    p.stride = p.nPt / p.nSt;
    p.nvals = (int64_t)p.nSt * 8;   // WR * 2 halves * 2 values per tile, WR = 2 for both tile shapes
  }
  while (p.npow2 < p.nvals) p.npow2 <<= 1;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
  p.o_qb = take((size_t)p.nq_pad * d * 2);
  p.o_qlo = take((size_t)p.nq_pad * d * 2);
  p.o_qnorm = take((size_t)p.nq_pad * 4);
  p.o_tau = take((size_t)p.nq_pad * 4);
  p.o_counts = take((size_t)p.nq_pad * IP_COUNT_STRIDE * 4);
  p.o_counts_packed = take((size_t)p.nq_pad * 4);
  p.o_m = take((size_t)p.nq_pad * 4);
  const size_t t_rows = p.mode == IP_MODE_FULL ? (size_t)p.nPt * p.tr : (size_t)p.nvals;
  p.o_T = take(t_rows * p.nq_pad * 4);
  p.o_id = take((size_t)nq * cap * 4);
  p.o_s = take((size_t)nq * cap * 4);
  p.o_x = take((size_t)nq * cap * 8);
  p.total = o;
  (void)k;
  return p;
}

template <int MODE, class T, bool X3, bool F16>
static int launch_scan_x(const ScanArgs& a, hipStream_t st) {
  static DeviceOnce attr_done;  // > 48 KB dynamic LDS needs the opt-in once per kernel and device
  if (attr_done.first())
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_ip_scan<MODE, T, X3, F16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         T::SMEM_BYTES));
  const unsigned tiles = (unsigned)a.nPt * (unsigned)a.nQt;
  static const bool r3 = getenv("CONVDR_DBG_SCAN_NO_R3") == nullptr;   // A/B switch: the two-stage loop
  if constexpr (MODE == IP_MODE_EMIT && !X3 && T::TR == 256) {
    if (r3) {
      constexpr int R3_SMEM = 3 * T::R_BYTES + 2 * T::L_BYTES;
      static DeviceOnce attr3;
      if (attr3.first())
        CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_ip_scan_r3<T, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_SMEM));
      ScanArgs b = a;
#ifdef CONVDR_ENABLE_TRACE   // timing only (every threshold = +inf: results are garbage): `make TRACE=1` library only
      if (getenv("CONVDR_DBG_SCAN_NOEMIT")) b.nq = 0;
      if (getenv("CONVDR_DBG_PRELANDED")) b.dbg_prelanded = 1;
#endif
      ProfScope prof("ip_scan_emit", st);
      hipLaunchKernelGGL((k_ip_scan_r3<T, F16>), dim3(std::min(tiles, (unsigned)device_cu_count())), dim3(T::THREADS), R3_SMEM, st, b);
      CONVDR_CHECK_LAUNCH("k_ip_scan_r3");
      return 0;
    }
  }
  static const bool one_tile_per_wg = getenv("CONVDR_DBG_SCAN_NONPERSISTENT") != nullptr;   // A/B switch for the walk
  const unsigned slots = (unsigned)device_cu_count() * (T::SMEM_BYTES > 80 * 1024 ? 1u : 2u);
  const unsigned grid = one_tile_per_wg ? tiles : std::min(tiles, slots);
  ProfScope prof(MODE == IP_MODE_EMIT ? "ip_scan_emit" : "ip_scan_sample", st);
  ScanArgs b = a;
#ifdef CONVDR_ENABLE_TRACE
  static const bool no_emit = getenv("CONVDR_DBG_SCAN_NOEMIT") != nullptr;   // timing only: every threshold = +inf
  if (no_emit) b.nq = 0;
#endif
  hipLaunchKernelGGL((k_ip_scan<MODE, T, X3, F16>), dim3(grid), dim3(T::THREADS), T::SMEM_BYTES, st, b);
  CONVDR_CHECK_LAUNCH("k_ip_scan");
  return 0;
}

template <int MODE, class T, bool F16>
static int launch_scan_t(const ScanArgs& a, hipStream_t st) {
  return a.Plo ? launch_scan_x<MODE, T, true, F16>(a, st) : launch_scan_x<MODE, T, false, F16>(a, st);
}

template <int MODE, bool F16>
static int launch_scan_k(const ScanArgs& a, int tile, hipStream_t st) {
  if (tile == IP_TILE_256) return launch_scan_t<MODE, Tile256, F16>(a, st);
  if (tile == IP_TILE_TALL) return launch_scan_t<MODE, TileTall, F16>(a, st);
  return launch_scan_t<MODE, Tile128, F16>(a, st);
}
template <int MODE>
static int launch_scan(const ScanArgs& a, int tile, int kind, hipStream_t st) {
  return kind == IP_KIND_F16 ? launch_scan_k<MODE, true>(a, tile, st) : launch_scan_k<MODE, false>(a, tile, st);
}

}  // namespace convdr

using namespace convdr;

// ---- two-way merge of sorted per-query lists (run_convdr_inference.py:213-229) -------------------------------------
// One workgroup per query.  Every element computes its own output slot: its index in its list + the number of
// elements of the other list that precede it (binary search on the descending scores held in LDS) -- A before B on
// ties, so for a in A the B elements strictly greater count, for b in B the A elements greater or equal.
namespace convdr {
__global__ void __launch_bounds__(256) k_topk_merge(const float* __restrict__ Da, const int64_t* __restrict__ Ia, int na,
                                                    int64_t lda, const float* __restrict__ Db,
                                                    const int64_t* __restrict__ Ib, int nb, int64_t ldb, int n_out,
                                                    float* __restrict__ Dout, int64_t* __restrict__ Iout, int64_t ldo) {
  extern __shared__ float sm[];
  float* sa = sm;
  float* sb = sm + na;
  const int q = blockIdx.x;
  Da += q * lda; Ia += q * lda; Db += q * ldb; Ib += q * ldb; Dout += q * ldo; Iout += q * ldo;
  for (int i = threadIdx.x; i < na; i += 256) sa[i] = Da[i];
  for (int i = threadIdx.x; i < nb; i += 256) sb[i] = Db[i];
  __syncthreads();
  for (int i = threadIdx.x; i < na + nb; i += 256) {
    const bool from_a = i < na;
    const int j = from_a ? i : i - na;
    const float v = from_a ? sa[j] : sb[j];
    const float* other = from_a ? sb : sa;
    int lo = 0, hi = from_a ? nb : na;       // first index of `other` that does NOT precede v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const bool precedes = from_a ? (other[mid] > v) : (other[mid] >= v);
      if (precedes) lo = mid + 1; else hi = mid;
    }
    const int pos = j + lo;
    if (pos < n_out) {
      Dout[pos] = v;
      Iout[pos] = from_a ? Ia[j] : Ib[j];
    }
  }
}
}  // namespace convdr

extern "C" int convdr_topk_merge(const float* Da, const int64_t* Ia, int na, int64_t lda, const float* Db, const int64_t* Ib,
                                 int nb, int64_t ldb, int nq, int n_out, float* Dout, int64_t* Iout, int64_t ldo,
                                 convdr_stream_t stream) {
  using namespace convdr;
  CONVDR_REQUIRE(na >= 0 && nb >= 0 && na <= 4096 && nb <= 4096 && n_out >= 0 && n_out <= na + nb && nq >= 0,
                 "convdr_topk_merge: bad sizes na=%d nb=%d n_out=%d nq=%d", na, nb, n_out, nq);
  if (nq == 0 || n_out == 0) return 0;
  hipLaunchKernelGGL(k_topk_merge, dim3(nq), dim3(256), (size_t)(na + nb) * 4, (hipStream_t)stream, Da, Ia, na, lda, Db, Ib,
                     nb, ldb, n_out, Dout, Iout, ldo);
  CONVDR_CHECK_LAUNCH("k_topk_merge");
  return 0;
}

extern "C" int convdr_ip_column_mean(const float* p_f32, int64_t n, int d, float* scratch /* >= 1024 * d floats */,
                                     float* mean, convdr_stream_t stream) {
  CONVDR_REQUIRE(n > 0 && d > 0, "convdr_ip_column_mean: empty block");
  // (64 row chunks x 3 column blocks = 192 workgroups streamed a 3 GB block at 0.5 TB/s: 5.8 ms of every first add();
  //  1024 chunks fill the chip)
  const int chunks = n >= 262144 ? 1024 : (n >= 4096 ? 64 : 1);
  hipLaunchKernelGGL(k_colsum_f32, dim3((d + 255) / 256, chunks), dim3(256), 0, (hipStream_t)stream, p_f32, n, d, scratch);
  hipLaunchKernelGGL(k_colmean_finish, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, chunks, d, n, mean);
  CONVDR_CHECK_LAUNCH("k_colsum_f32");
  return 0;
}

static int ip_prepare_block(int kind, const float* p_f32, int64_t n, int d, const float* centre, float scale, void* p_half,
                            void* p_half_lo, float* max_norm, hipStream_t st) {
  CONVDR_REQUIRE(n >= 0 && d > 0 && d % 64 == 0, "convdr_ip_prepare_block: need d %% 64 == 0 (got n=%lld d=%d)",
                 (long long)n, d);
  if (n == 0) return 0;
  const int64_t blocks = ceil_div64(n, 4);
  const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
  if (kind == IP_KIND_F16) {
    int ex = 0;
    CONVDR_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "convdr_ip_prepare_block_f16: scale must be a power of two (got %g)",
                   (double)scale);
    hipLaunchKernelGGL((k_rows_to_half<IP_KIND_F16, false>), dim3(grid), dim3(256), 0, st, p_f32, n, d, centre, scale,
                       (bf16_t*)p_half, (bf16_t*)p_half_lo, (float*)nullptr, max_norm);
  } else {
    hipLaunchKernelGGL((k_rows_to_half<IP_KIND_BF16, false>), dim3(grid), dim3(256), 0, st, p_f32, n, d, centre, 1.f,
                       (bf16_t*)p_half, (bf16_t*)p_half_lo, (float*)nullptr, max_norm);
  }
  CONVDR_CHECK_LAUNCH("k_rows_to_half");
  return 0;
}

extern "C" int convdr_ip_prepare_block(const float* p_f32, int64_t n, int d, const float* centre, void* p_bf16,
                                       void* p_bf16_lo, float* max_norm, convdr_stream_t stream) {
  return ip_prepare_block(IP_KIND_BF16, p_f32, n, d, centre, 1.f, p_bf16, p_bf16_lo, max_norm, (hipStream_t)stream);
}

extern "C" int convdr_ip_prepare_block_f16(const float* p_f32, int64_t n, int d, const float* centre, float scale, void* p_f16,
                                           void* p_f16_lo, float* max_norm, convdr_stream_t stream) {
  return ip_prepare_block(IP_KIND_F16, p_f32, n, d, centre, scale, p_f16, p_f16_lo, max_norm, (hipStream_t)stream);
}

extern "C" float convdr_ip_f16_scale(float max_norm) {
  // power of two that puts max_norm into [2^12, 2^13): elements stay below 2^13 (fp16 max 65504 leaves a factor 7 for
  // blocks added later), typical elements (norm / sqrt(d)) far above the 2^-14 end of the normal range
  if (!(max_norm > 0.f) || max_norm == INFINITY) return 1.f;
  int ex = 0;
  (void)frexpf(max_norm, &ex);
  int sh = (int)IP_F16_TARGET_EXP + 1 - ex;
  sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  return ldexpf(1.f, sh);
}

extern "C" size_t convdr_ip_workspace_bytes(int nq, int64_t n, int d, int k, int cap) {
  if (nq <= 0 || n < 0 || d <= 0) return 0;
  return ip_plan(nq, n, d, k, cap).total;
}

extern "C" const uint32_t* convdr_ip_debug_counts(const void* workspace, int nq, int64_t n, int d, int k, int cap) {
  return (const uint32_t*)((const char*)workspace + ip_plan(nq, n, d, k, cap).o_counts_packed);
}

extern "C" const uint32_t* convdr_ip_debug_band(const void* workspace, int nq, int64_t n, int d, int k, int cap) {
  return (const uint32_t*)((const char*)workspace + ip_plan(nq, n, d, k, cap).o_m);
}

static int ip_search(int kind, float p_scale, const float* q_f32, int nq, const float* p_f32, const void* p_bf16,
                     const void* p_bf16_lo, int64_t n, int d, int k, const float* p_max_norm, const float* tau_in, int cap,
                     int rank_target, void* workspace, size_t workspace_bytes, float* D, int64_t* I, int32_t* status,
                     float* tau_retry, hipStream_t st) {
  CONVDR_REQUIRE(nq > 0 && k > 0 && n >= 0, "convdr_ip_search: bad sizes nq=%d k=%d n=%lld", nq, k, (long long)n);
  CONVDR_REQUIRE(d > 0 && d % 64 == 0 && d <= 4096, "convdr_ip_search: need d %% 64 == 0 and d <= 4096 (got %d)", d);
  CONVDR_REQUIRE(n < ((int64_t)1 << 31), "convdr_ip_search: block too large (n=%lld >= 2^31)", (long long)n);
  CONVDR_REQUIRE(cap >= 1024 && cap <= 8192 && (cap & (cap - 1)) == 0,
                 "convdr_ip_search: cap must be a power of two in [1024, 8192] (got %d)", cap);
  CONVDR_REQUIRE(k <= cap / 2, "convdr_ip_search: k=%d too large for cap=%d", k, cap);
  const IpPlan p = ip_plan(nq, n, d, k, cap);
  CONVDR_REQUIRE(workspace_bytes >= p.total, "convdr_ip_search: workspace too small (%zu < %zu)", workspace_bytes,
                 p.total);
  char* ws = (char*)workspace;
  bf16_t* qb = (bf16_t*)(ws + p.o_qb);
  float* qnorm = (float*)(ws + p.o_qnorm);
  float* tau = (float*)(ws + p.o_tau);
  uint32_t* counts = (uint32_t*)(ws + p.o_counts);
  float* T = (float*)(ws + p.o_T);
  uint32_t* cand_id = (uint32_t*)(ws + p.o_id);
  float* cand_s = (float*)(ws + p.o_s);
  double* cand_x = (double*)(ws + p.o_x);
  uint32_t* band = (uint32_t*)(ws + p.o_m);

  // queries -> 16-bit operands (+ norms); the kernel also zeroes the padding rows and the candidate counters
  bf16_t* qlo = p_bf16_lo ? (bf16_t*)(ws + p.o_qlo) : nullptr;
  const int64_t n_count = (int64_t)p.nq_pad * IP_COUNT_STRIDE;
  if (kind == IP_KIND_F16)
    hipLaunchKernelGGL((k_rows_to_half<IP_KIND_F16, true>), dim3((p.nq_pad + 3) / 4), dim3(256), 0, st, q_f32, (int64_t)nq, d,
                       (const float*)nullptr, 1.f, qb, qlo, qnorm, (float*)nullptr, (int64_t)p.nq_pad, counts, n_count);
  else
    hipLaunchKernelGGL((k_rows_to_half<IP_KIND_BF16, false>), dim3((p.nq_pad + 3) / 4), dim3(256), 0, st, q_f32, (int64_t)nq, d,
                       (const float*)nullptr, 1.f, qb, qlo, qnorm, (float*)nullptr, (int64_t)p.nq_pad, counts, n_count);
  CONVDR_CHECK_LAUNCH("k_rows_to_half(Q)");

  if (n == 0) {
    hipLaunchKernelGGL(k_fill_f32, dim3((p.nq_pad + 255) / 256), dim3(256), 0, st, tau, p.nq_pad, -INFINITY);
  } else {
    ScanArgs a{};
    a.P = (const bf16_t*)p_bf16; a.Qb = qb; a.Plo = (const bf16_t*)p_bf16_lo; a.Qlo = qlo; a.n = n; a.nq = nq; a.nq_pad = p.nq_pad; a.d = d;
    a.nQt = p.nQt; a.tau = tau; a.counts = counts; a.cand_id = cand_id; a.cand_s = cand_s; a.cap = cap; a.T = T;
    if (tau_in) {
      CONVDR_CHECK_HIP(hipMemcpyAsync(tau, tau_in, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    } else if (p.mode < 0) {
      hipLaunchKernelGGL(k_fill_f32, dim3((p.nq_pad + 255) / 256), dim3(256), 0, st, tau, p.nq_pad, -INFINITY);
      CONVDR_CHECK_LAUNCH("k_fill_f32");
    } else {
      int R = rank_target > 0 ? rank_target : 16 * k;
      if (R > cap / 2) R = cap / 2;
      if (R < k) R = k;
      int r;
      a.nPt = p.nSt; a.pt_stride = p.stride;
      if (p.mode == IP_MODE_FULL) {
        r = (int64_t)R < n ? R : (int)n;
        if (int e = launch_scan<IP_MODE_FULL>(a, p.big, kind, st)) return e;
      } else {
        // The sample keeps the two best scores of every 64 sampled passages, so it can only represent a rank whose expected
        // hits per 64 passages stay well below 2: R <= n / 128 (half a hit per 64).  Blocks of 32 k .. 200 k passages
        // therefore aim at a lower rank than 16 k (n = 47,104 asked for rank 1,113 of a 1,024-value sample: no threshold,
        // every passage emitted, every query overflowed and was re-run); a band that then reaches below the threshold
        // comes back UNCERTAIN with the threshold to retry, as for any clustered block.
        if ((int64_t)R > n / 128) R = (int)(n / 128 > k ? n / 128 : k);
        const double frac = (double)p.nSt * p.tr / (double)n;
        r = (int)lrint(R * frac);
        if (r < 8) r = 8;
        if (r > p.nvals / 4) r = (int)(p.nvals / 4);
        if (int e = launch_scan<IP_MODE_TOP2>(a, p.big, kind, st)) return e;
      }
      static DeviceOnce attr_done;
      if (attr_done.first())
        CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_tau_select, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             IP_FULL_MAX_N * 4));
      hipLaunchKernelGGL(k_tau_select, dim3(nq), dim3(1024), (size_t)p.npow2 * 4, st, T, p.nvals, p.nq_pad, r, tau);
      CONVDR_CHECK_LAUNCH("k_tau_select");
    }
    a.nPt = p.nPt; a.pt_stride = 1;
    if (int e = launch_scan<IP_MODE_EMIT>(a, p.big, kind, st)) return e;
  }
  static DeviceOnce attr_done2;
  if (attr_done2.first()) {
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_ip_cut, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    CONVDR_CHECK_HIP(
        hipFuncSetAttribute((const void*)k_ip_select, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12));
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_ip_finish, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
  }
  if (g_ip_fused_finish && n > 0) {
    ProfScope prof("ip_finish", st);
    hipLaunchKernelGGL(k_ip_finish, dim3(nq), dim3(1024), (size_t)cap * 16, st, n, k, cap, counts,
                       (uint32_t*)(ws + p.o_counts_packed), cand_id, cand_s, tau, qnorm, p_max_norm,
                       ip_eps_coef(d, p_bf16_lo != nullptr, kind), ip_eps_abs(d, p_bf16_lo != nullptr, kind),
                       kind == IP_KIND_F16 ? p_scale : 1.f, kind == IP_KIND_F16 ? IP_F16_NORM_LIMIT : INFINITY, q_f32, p_f32, d,
                       band, status, tau_retry, D, I);
    CONVDR_CHECK_LAUNCH("k_ip_finish");
    return 0;
  }
  {
  ProfScope prof("ip_cut", st);
  hipLaunchKernelGGL(k_ip_cut, dim3(nq), dim3(1024), (size_t)cap * 8, st, n, k, cap, counts,
                     (uint32_t*)(ws + p.o_counts_packed), cand_id, cand_s, tau, qnorm,
                     p_max_norm, ip_eps_coef(d, p_bf16_lo != nullptr, kind), ip_eps_abs(d, p_bf16_lo != nullptr, kind),
                     kind == IP_KIND_F16 ? p_scale : 1.f, kind == IP_KIND_F16 ? IP_F16_NORM_LIMIT : INFINITY, band, status,
                     tau_retry);
  CONVDR_CHECK_LAUNCH("k_ip_cut");
  }
  if (n > 0) {
    ProfScope prof("ip_rescore", st);
    hipLaunchKernelGGL(k_ip_rescore, dim3(nq, 16), dim3(256), 0, st, q_f32, p_f32, d, cap, band, cand_id, cand_x);
    CONVDR_CHECK_LAUNCH("k_ip_rescore");
  }
  ProfScope prof("ip_select", st);
  hipLaunchKernelGGL(k_ip_select, dim3(nq), dim3(IP_SELECT_THREADS), (size_t)cap * 12, st, k, cap, band, cand_id, cand_x, D, I);
  CONVDR_CHECK_LAUNCH("k_ip_select");
  return 0;
}

extern "C" int convdr_ip_search(const float* q_f32, int nq, const float* p_f32, const void* p_bf16, const void* p_bf16_lo,
                                int64_t n, int d, int k, const float* p_max_norm, const float* tau_in, int cap,
                                int rank_target, void* workspace, size_t workspace_bytes, float* D, int64_t* I,
                                int32_t* status, float* tau_retry, convdr_stream_t stream) {
  return ip_search(IP_KIND_BF16, 1.f, q_f32, nq, p_f32, p_bf16, p_bf16_lo, n, d, k, p_max_norm, tau_in, cap, rank_target,
                   workspace, workspace_bytes, D, I, status, tau_retry, (hipStream_t)stream);
}

extern "C" int convdr_ip_search_f16(const float* q_f32, int nq, const float* p_f32, const void* p_f16, const void* p_f16_lo,
                                    float p_scale, int64_t n, int d, int k, const float* p_max_norm, const float* tau_in,
                                    int cap, int rank_target, void* workspace, size_t workspace_bytes, float* D,
                                    int64_t* I, int32_t* status, float* tau_retry, convdr_stream_t stream) {
  int ex = 0;
  CONVDR_REQUIRE(p_scale > 0.f && frexpf(p_scale, &ex) == 0.5f, "convdr_ip_search_f16: p_scale must be a power of two (got %g)",
                 (double)p_scale);
  return ip_search(IP_KIND_F16, p_scale, q_f32, nq, p_f32, p_f16, p_f16_lo, n, d, k, p_max_norm, tau_in, cap, rank_target,
                   workspace, workspace_bytes, D, I, status, tau_retry, (hipStream_t)stream);
}
