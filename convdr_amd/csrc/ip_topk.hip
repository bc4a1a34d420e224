// Exact brute-force inner-product top-k over a resident passage-embedding block (gfx950).
//
// Replaces faiss.IndexFlatIP.add/.search as driven by the reference at
//   /root/reference/drivers/run_convdr_inference.py:180-182  (gpu_index.add(block); D, I = gpu_index.search(Q, topN))
//
// Design (see DESIGN.md §"IP scan"): never materialise [nq, n].
//   prepare   fp32 block -> bf16 scan copy + max row norm                      (HBM-bound, once per .add)
//   sample    bf16 MFMA GEMM over ~1/32 of the block, epilogue keeps per-(query, 32 passages) top-2,
//             per-query LDS bitonic sort -> tau[q] ~ score of rank `rank_target` in the whole block
//   scan      bf16 MFMA GEMM  P_bf16[n,d] * Q_bf16[nq,d]^T  (gemm_nt.hpp), epilogue compares each fp32
//             accumulator with tau[q] and appends (index, score) of the rare hits to a per-query list
//   rescore   one wave per candidate: fp64 dot of the fp32 originals, canonical order (oracle/search.py)
//   select    per query LDS bitonic sort by (exact score desc, index asc), top-k out, certificate:
//             OK iff  kth_exact >= tau + eps,  eps = 0.0079 * |q| * max|p|   (bf16 rounding bound)
#include "gemm_nt.hpp"

#include <float.h>
#include <math.h>

#include "../../include/convdr_hip.h"

namespace convdr {

constexpr int IP_MODE_FULL = 0, IP_MODE_TOP2 = 1, IP_MODE_EMIT = 2;
constexpr int IP_FULL_MAX_N = 32768;      // <= this many passages: "sample" = all scores, exact rank select
constexpr int IP_SAMPLE_MIN = 32768;      // sampled passages (>= 1/32 of the block)
constexpr int IP_SAMPLE_MAX = 262144;
constexpr float IP_EPS_COEF = 0.0079f;    // bf16 scan: 2u + u^2 + K*2^-24 with u = 2^-8, rounded up (K <= 4096)
// split-bf16 scan (hi*hi + hi*lo + lo*hi): dropped terms 3 u^2 = 4.6e-5, fp32 accumulation of 3K products
// 3 * 4096 * 2^-24 = 7.4e-4 worst case -> rounded up
constexpr float IP_EPS_COEF_X3 = 8.0e-4f;

// ------------------------------------------------------------------------------------------
// rows fp32 -> bf16 of (x - centre) (+ per-row L2 norm of the centred row, + global max norm).
// Optional second output: the bf16 of the rounding remainder (x - centre) - hi, for the split-bf16 scan.
// One wave per row, float4 loads.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rows_to_bf16(const float* __restrict__ X, int64_t n, int d,
                                                      const float* __restrict__ centre, bf16_t* __restrict__ Y,
                                                      bf16_t* __restrict__ Ylo, float* __restrict__ row_norm,
                                                      float* __restrict__ max_norm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wmax = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
    const float* x = X + row * d;
    float ss = 0.f;
    for (int e = lane * 4; e < d; e += 256) {
      float4 v = *(const float4*)(x + e);
      if (centre) {
        const float4 c = *(const float4*)(centre + e);
        v.x -= c.x; v.y -= c.y; v.z -= c.z; v.w -= c.w;
      }
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      const bf16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
      uint2 o;
      o.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
      o.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
      *(uint2*)(Y + row * d + e) = o;
      if (Ylo) {
        uint2 l;
        l.x = pack_bf16x2(v.x - bf16_to_f32(h0), v.y - bf16_to_f32(h1));
        l.y = pack_bf16x2(v.z - bf16_to_f32(h2), v.w - bf16_to_f32(h3));
        *(uint2*)(Ylo + row * d + e) = l;
      }
    }
    ss = wave_sum(ss);
    const float nm = sqrtf(ss);
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
  uint32_t* counts;  // EMIT: [nq]
  uint32_t* cand_id; // EMIT: [nq, cap]
  float* cand_s;     // EMIT: [nq, cap]
  int cap;
  float* T;          // FULL: [nPt*128, nq_pad]; TOP2: [nPt*8, nq_pad]
};

template <int MODE, class T>
__global__ void __launch_bounds__(T::THREADS, 2) k_ip_scan(const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ts = logical / a.nQt, qt = logical - ts * a.nQt;
  const int64_t m0 = (int64_t)ts * a.pt_stride * T::TR;
  const int64_t n0 = (int64_t)qt * T::TL;

  const WavePos<T> w;
  GemmAcc<T> acc;
  acc.zero();
  gemm_nt_mainloop<T>(a.P, a.d, a.n, a.Qb, a.d, a.nq_pad, a.d, m0, n0, smem, acc, w);
  if (a.Plo) {  // S~ = Ph Qh + Ph Ql + Pl Qh: fp32-class scores from three bf16 passes into the same accumulators
    __syncthreads();
    gemm_nt_mainloop<T>(a.P, a.d, a.n, a.Qlo, a.d, a.nq_pad, a.d, m0, n0, smem, acc, w);
    __syncthreads();
    gemm_nt_mainloop<T>(a.Plo, a.d, a.n, a.Qb, a.d, a.nq_pad, a.d, m0, n0, smem, acc, w);
  }

  if constexpr (MODE == IP_MODE_EMIT) {
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
      const int q = (int)n0 + w.l_index(nt);
      const float tau = q < a.nq ? a.tau[q] : INFINITY;
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt) {
        const f32x16 v = acc.c[mt][nt];
        float mx = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, v[r]);
        if (mx >= tau) {  // rare: ~rank_target hits per query in the whole block
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + w.r_index(mt, r);
            if (v[r] >= tau && row < a.n) {
              const uint32_t slot = atomicAdd(&a.counts[q], 1u);
              if (slot < (uint32_t)a.cap) {
                a.cand_id[(int64_t)q * a.cap + slot] = (uint32_t)row;
                a.cand_s[(int64_t)q * a.cap + slot] = v[r];
              }
            }
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

// ------------------------------------------------------------------------------------------
// LDS bitonic sorts (block-wide)
// ------------------------------------------------------------------------------------------
__device__ void bitonic_desc_f32(float* s, int n) {
  for (int k2 = 2; k2 <= n; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const float x = s[i], y = s[p];
          const bool desc = (i & k2) == 0;
          if (desc ? (x < y) : (x > y)) { s[i] = y; s[p] = x; }
        }
      }
      __syncthreads();
    }
}

__device__ __forceinline__ bool cand_before(double sa, uint32_t ia, double sb, uint32_t ib) {
  return sa > sb || (sa == sb && ia < ib);
}

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
__global__ void __launch_bounds__(1024) k_tau_select(const float* __restrict__ T, int64_t nvals, int nq_pad,
                                                     int npow2, int r, float* __restrict__ tau) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s = (float*)smem;
  const int q = blockIdx.x;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) s[i] = i < nvals ? T[(int64_t)i * nq_pad + q] : -INFINITY;
  __syncthreads();
  bitonic_desc_f32(s, npow2);
  if (threadIdx.x == 0) tau[q] = (r >= 1 && r <= nvals) ? s[r - 1] : -INFINITY;
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
// Sorts the query's candidates by (s~ desc, index asc) in place and writes m[q] = band size.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ip_cut(int64_t n, int k, int cap, const uint32_t* __restrict__ counts,
                                                uint32_t* __restrict__ cand_id, float* __restrict__ cand_s,
                                                const float* __restrict__ tau, const float* __restrict__ qnorm,
                                                const float* __restrict__ p_max_norm, float eps_coef,
                                                uint32_t* __restrict__ m_out, int32_t* __restrict__ status,
                                                float* __restrict__ tau_retry) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int sh_m;
  const int q = blockIdx.x;
  const uint32_t cnt = counts[q];
  const int c = cnt < (uint32_t)cap ? (int)cnt : cap;
  int np2 = 2;
  while (np2 < c) np2 <<= 1;
  float* s = (float*)smem;
  uint32_t* id = (uint32_t*)(smem + (size_t)np2 * 4);
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    s[i] = i < c ? cand_s[(int64_t)q * cap + i] : -INFINITY;
    id[i] = i < c ? cand_id[(int64_t)q * cap + i] : 0xffffffffu;
  }
  if (threadIdx.x == 0) sh_m = 0;
  __syncthreads();
  for (int k2 = 2; k2 <= np2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const float x = s[i], y = s[p];
          const uint32_t ix = id[i], iy = id[p];
          const bool fwd = (i & k2) == 0;
          const bool y_first = y > x || (y == x && iy < ix);
          const bool x_first = x > y || (x == y && ix < iy);
          if (fwd ? y_first : x_first) { s[i] = y; s[p] = x; id[i] = iy; id[p] = ix; }
        }
      }
      __syncthreads();
    }
  const int need = (int64_t)k < n ? k : (int)n;
  const float t = tau[q];
  const float eps = eps_coef * qnorm[q] * p_max_norm[0] * 1.001f + 1e-30f;
  const bool have_k = need > 0 && c >= need;
  const float cut = have_k ? s[need - 1] - 2.f * eps : -INFINITY;
  for (int i = threadIdx.x; i < c; i += blockDim.x)
    if (s[i] >= cut && (i + 1 == c || !(s[i + 1] >= cut))) sh_m = i + 1;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    cand_id[(int64_t)q * cap + i] = id[i];
    cand_s[(int64_t)q * cap + i] = s[i];
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
    m_out[q] = (uint32_t)sh_m;
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
// per-query final sort of the re-scored band by (exact score desc, index asc) -> top-k
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ip_select(int k, int cap, const uint32_t* __restrict__ m_in,
                                                   const uint32_t* __restrict__ cand_id,
                                                   const double* __restrict__ cand_x, float* __restrict__ D,
                                                   int64_t* __restrict__ I) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int q = blockIdx.x;
  const int c = (int)m_in[q];
  int np2 = 2;
  while (np2 < c) np2 <<= 1;
  double* s = (double*)smem;
  uint32_t* id = (uint32_t*)(smem + (size_t)np2 * 8);
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    s[i] = i < c ? cand_x[(int64_t)q * cap + i] : -INFINITY;
    id[i] = i < c ? cand_id[(int64_t)q * cap + i] : 0xffffffffu;
  }
  __syncthreads();
  bitonic_cand(s, id, np2);
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    D[(int64_t)q * k + j] = j < c ? (float)s[j] : -FLT_MAX;
    I[(int64_t)q * k + j] = j < c ? (int64_t)id[j] : -1;
  }
}

// ------------------------------------------------------------------------------------------
// host-side plan shared by workspace sizing and the search call
// ------------------------------------------------------------------------------------------
struct IpPlan {
  bool big;          // 256 x 256 scan tiles (more than 128 queries), else 128 x 128
  int tr, tl;        // tile extent over passages / queries
  int nq_pad, nQt, nPt;
  int mode;          // -1: no threshold pass (n <= cap), else IP_MODE_FULL / IP_MODE_TOP2
  int nSt, stride;   // sampled passage tiles / tile stride
  int64_t nvals;     // values per query handed to k_tau_select
  int npow2;
  size_t o_qb, o_qlo, o_qnorm, o_tau, o_counts, o_m, o_T, o_id, o_s, o_x, total;
};

static IpPlan ip_plan(int nq, int64_t n, int d, int k, int cap) {
  IpPlan p;
  p.big = nq > 128;
  p.tr = p.big ? Tile256::TR : Tile128::TR;
  p.tl = p.big ? Tile256::TL : Tile128::TL;
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
  p.o_counts = take((size_t)p.nq_pad * 4);
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

template <int MODE, class T>
static int launch_scan_t(const ScanArgs& a, hipStream_t st) {
  static bool attr_done = false;  // > 48 KB dynamic LDS needs the opt-in once per kernel
  if (!attr_done) {
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_ip_scan<MODE, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         T::SMEM_BYTES));
    attr_done = true;
  }
  const unsigned grid = (unsigned)a.nPt * (unsigned)a.nQt;
  ProfScope prof(MODE == IP_MODE_EMIT ? "ip_scan_emit" : "ip_scan_sample", st);
  hipLaunchKernelGGL((k_ip_scan<MODE, T>), dim3(grid), dim3(T::THREADS), T::SMEM_BYTES, st, a);
  CONVDR_CHECK_LAUNCH("k_ip_scan");
  return 0;
}

template <int MODE>
static int launch_scan(const ScanArgs& a, bool big, hipStream_t st) {
  return big ? launch_scan_t<MODE, Tile256>(a, st) : launch_scan_t<MODE, Tile128>(a, st);
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

extern "C" int convdr_ip_column_mean(const float* p_f32, int64_t n, int d, float* scratch /* >= 64 * d floats */,
                                     float* mean, convdr_stream_t stream) {
  CONVDR_REQUIRE(n > 0 && d > 0, "convdr_ip_column_mean: empty block");
  const int chunks = n >= 4096 ? 64 : 1;
  hipLaunchKernelGGL(k_colsum_f32, dim3((d + 255) / 256, chunks), dim3(256), 0, (hipStream_t)stream, p_f32, n, d, scratch);
  hipLaunchKernelGGL(k_colmean_finish, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, chunks, d, n, mean);
  CONVDR_CHECK_LAUNCH("k_colsum_f32");
  return 0;
}

extern "C" int convdr_ip_prepare_block(const float* p_f32, int64_t n, int d, const float* centre, void* p_bf16,
                                       void* p_bf16_lo, float* max_norm, convdr_stream_t stream) {
  CONVDR_REQUIRE(n >= 0 && d > 0 && d % 64 == 0, "convdr_ip_prepare_block: need d %% 64 == 0 (got n=%lld d=%d)",
                 (long long)n, d);
  if (n == 0) return 0;
  const int64_t blocks = ceil_div64(n, 4);
  const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
  hipLaunchKernelGGL(k_rows_to_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, p_f32, n, d, centre, (bf16_t*)p_bf16,
                     (bf16_t*)p_bf16_lo, (float*)nullptr, max_norm);
  CONVDR_CHECK_LAUNCH("k_rows_to_bf16");
  return 0;
}

extern "C" size_t convdr_ip_workspace_bytes(int nq, int64_t n, int d, int k, int cap) {
  if (nq <= 0 || n < 0 || d <= 0) return 0;
  return ip_plan(nq, n, d, k, cap).total;
}

extern "C" const uint32_t* convdr_ip_debug_counts(const void* workspace, int nq, int64_t n, int d, int k, int cap) {
  return (const uint32_t*)((const char*)workspace + ip_plan(nq, n, d, k, cap).o_counts);
}

extern "C" const uint32_t* convdr_ip_debug_band(const void* workspace, int nq, int64_t n, int d, int k, int cap) {
  return (const uint32_t*)((const char*)workspace + ip_plan(nq, n, d, k, cap).o_m);
}

extern "C" int convdr_ip_search(const float* q_f32, int nq, const float* p_f32, const void* p_bf16, const void* p_bf16_lo,
                                int64_t n, int d,
                                int k, const float* p_max_norm, const float* tau_in, int cap, int rank_target,
                                void* workspace, size_t workspace_bytes, float* D, int64_t* I, int32_t* status,
                                float* tau_retry, convdr_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
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

  // queries -> bf16 (+ norms); padded rows stay zero
  CONVDR_CHECK_HIP(hipMemsetAsync(qb, 0, (size_t)p.nq_pad * d * 2, st));
  CONVDR_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)p.nq_pad * 4, st));
  bf16_t* qlo = p_bf16_lo ? (bf16_t*)(ws + p.o_qlo) : nullptr;
  if (qlo) CONVDR_CHECK_HIP(hipMemsetAsync(qlo, 0, (size_t)p.nq_pad * d * 2, st));
  hipLaunchKernelGGL(k_rows_to_bf16, dim3((nq + 3) / 4), dim3(256), 0, st, q_f32, (int64_t)nq, d, (const float*)nullptr, qb,
                     qlo, qnorm, (float*)nullptr);
  CONVDR_CHECK_LAUNCH("k_rows_to_bf16(Q)");

  if (n == 0) {
    hipLaunchKernelGGL(k_fill_f32, dim3((p.nq_pad + 255) / 256), dim3(256), 0, st, tau, p.nq_pad, -INFINITY);
  } else {
    ScanArgs a;
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
        if (int e = launch_scan<IP_MODE_FULL>(a, p.big, st)) return e;
      } else {
        const double frac = (double)p.nSt * p.tr / (double)n;
        r = (int)lrint(R * frac);
        if (r < 8) r = 8;
        if (int e = launch_scan<IP_MODE_TOP2>(a, p.big, st)) return e;
      }
      static bool attr_done = false;
      if (!attr_done) {
        CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_tau_select, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             IP_FULL_MAX_N * 4));
        attr_done = true;
      }
      hipLaunchKernelGGL(k_tau_select, dim3(nq), dim3(1024), (size_t)p.npow2 * 4, st, T, p.nvals, p.nq_pad, p.npow2,
                         r, tau);
      CONVDR_CHECK_LAUNCH("k_tau_select");
    }
    a.nPt = p.nPt; a.pt_stride = 1;
    if (int e = launch_scan<IP_MODE_EMIT>(a, p.big, st)) return e;
  }
  static bool attr_done2 = false;
  if (!attr_done2) {
    CONVDR_CHECK_HIP(hipFuncSetAttribute((const void*)k_ip_cut, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    CONVDR_CHECK_HIP(
        hipFuncSetAttribute((const void*)k_ip_select, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12));
    attr_done2 = true;
  }
  hipLaunchKernelGGL(k_ip_cut, dim3(nq), dim3(256), (size_t)cap * 8, st, n, k, cap, counts, cand_id, cand_s, tau, qnorm,
                     p_max_norm, p_bf16_lo ? IP_EPS_COEF_X3 : IP_EPS_COEF, band, status, tau_retry);
  CONVDR_CHECK_LAUNCH("k_ip_cut");
  if (n > 0) {
    ProfScope prof("ip_rescore", st);
    hipLaunchKernelGGL(k_ip_rescore, dim3(nq, 16), dim3(256), 0, st, q_f32, p_f32, d, cap, band, cand_id, cand_x);
    CONVDR_CHECK_LAUNCH("k_ip_rescore");
  }
  hipLaunchKernelGGL(k_ip_select, dim3(nq), dim3(256), (size_t)cap * 12, st, k, cap, band, cand_id, cand_x, D, I);
  CONVDR_CHECK_LAUNCH("k_ip_select");
  return 0;
}
