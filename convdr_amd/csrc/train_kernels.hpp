// Backward-pass kernels of the dual-encoder training step on gfx950.
//
// Replaces what `loss.backward()` (/root/reference/drivers/run_convdr_train.py:178) runs through torch autograd
// for the student encoder: LayerNorm / GELU / attention / embedding backward and the data-layout helpers the
// dgrad / wgrad MFMA GEMMs need.  All dense contractions reuse the NT tile engine (gemm_nt.hpp):
//   dgrad  dX[t,k]  = sum_n dY[t,n] W[n,k]    -> operands dY [rows,N] and W^T [K,N]   (W^T kept packed)
//   wgrad  dW[n,k]  = sum_t dY[t,n] X[t,k]    -> TN engine (gemm_tn.hpp), straight from the token-major operands
#pragma once
#include "encoder_kernels.hpp"

namespace convdr {

// fp32 [n, k] -> bf16 [k, n]  (weight packing for dgrad: W^T)
__global__ void __launch_bounds__(256) k_transpose_f32_bf16(const float* __restrict__ in, int n, int k,
                                                            bf16_t* __restrict__ out) {
  __shared__ float tile[64][65];
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (n0 + r < n && k0 + c < k) ? in[(int64_t)(n0 + r) * k + k0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;  // out row = k0 + r, col = n0 + c
    if (k0 + r < k && n0 + c < n) out[(int64_t)(k0 + r) * n + n0 + c] = f32_to_bf16(tile[c][r]);
  }
}

// The same for up to TR_MAX_JOBS matrices in ONE launch (the 49 transposed weight copies of a roberta-base student were 49
// launches of ~11 us on the side stream of every training step): blockIdx.x walks the concatenated 64 x 64 tile lists.
constexpr int TR_MAX_JOBS = 64;
struct TransposeJob { const void* in; bf16_t* out; int n, k, tiles_k, tile_end; };   // in: fp32 (SRC_BF16 = false) or bf16
struct TransposeJobs { TransposeJob j[TR_MAX_JOBS]; int count; };
// NO LDS, on purpose: this launch (20,880 tiles for a roberta-base student) runs on a side stream UNDER the forward, whose
// 256 x 256 GEMM workgroups need all 160 KB of a CU's LDS to start -- a flood of small workgroups that each hold a 16 KB tile keeps
// every CU partly occupied and the forward's next GEMM waits for the flood to drain (the LDS form cost the step 0.48 ms for 0.17 ms
// of kernel: profiles/r06_ab_transpose.txt).  A wave owns 64 rows n x 16 columns k of a 64 x 64 tile: each lane reads 64
// contiguous bytes of its row (the four waves of the workgroup take the four quarters of the rows' 256-byte spans, so every
// 128-byte line is consumed by two neighbouring waves) and the wave stores sixteen 128-byte rows of the transposed matrix.
// SRC_BF16: the source is the bf16 copy of the same weights (the one the optimizer step keeps current for the forward GEMMs:
// convdr_adamw_step_packed) -- a lane then reads 32 bytes of its row and moves bits: a third less traffic for the same result.
template <bool SRC_BF16>
__global__ void __launch_bounds__(256) k_transpose_bf16_batch(const TransposeJobs a) {
  int ji = 0;
  for (int i = 0; i + 1 < a.count; ++i)
    if ((int)blockIdx.x >= a.j[i].tile_end) ji = i + 1;
  const TransposeJob& q = a.j[ji];
  const int t = (int)blockIdx.x - (ji ? a.j[ji - 1].tile_end : 0);
  const int n = q.n, k = q.k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = (t % q.tiles_k) * 64 + 16 * wave, row = (t / q.tiles_k) * 64 + lane;
  bf16_t* __restrict__ out = q.out + (int64_t)k0 * n + row;
  if (row >= n || k0 >= k) return;
  if constexpr (SRC_BF16) {
    const bf16_t* __restrict__ in = (const bf16_t*)q.in + (int64_t)row * k + k0;
    if (k0 + 16 <= k && (((uintptr_t)in) & 15) == 0) {
      const uint4 x0 = *(const uint4*)in, x1 = *(const uint4*)(in + 8);
      const uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        out[(int64_t)(2 * j) * n] = (bf16_t)(w[j] & 0xffffu);
        out[(int64_t)(2 * j + 1) * n] = (bf16_t)(w[j] >> 16);
      }
    } else {
      for (int j = 0; j < 16 && k0 + j < k; ++j) out[(int64_t)j * n] = in[j];
    }
  } else {
    const float* __restrict__ in = (const float*)q.in + (int64_t)row * k + k0;
    float v[16];
    if (k0 + 16 <= k && (((uintptr_t)in) & 15) == 0) {
      const float4 x0 = *(const float4*)in, x1 = *(const float4*)(in + 4), x2 = *(const float4*)(in + 8), x3 = *(const float4*)(in + 12);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      v[8] = x2.x; v[9] = x2.y; v[10] = x2.z; v[11] = x2.w; v[12] = x3.x; v[13] = x3.y; v[14] = x3.z; v[15] = x3.w;
#pragma unroll
      for (int j = 0; j < 16; ++j) out[(int64_t)j * n] = f32_to_bf16(v[j]);
    } else {
      for (int j = 0; j < 16 && k0 + j < k; ++j) out[(int64_t)j * n] = f32_to_bf16(in[j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (grid-stride), H <= 1024.
//   xhat = (y - mean) * rstd;  gdy = g * dy;  dx = rstd * (gdy - mean(gdy) - xhat * mean(gdy * xhat))
// Per-workgroup partial sums go to part[block][3][H] = (sum dx, dgamma = sum dy * xhat, dbeta = sum dy); sum dx is the
// bias gradient of the dense layer whose output (+ residual) this LayerNorm normalises -- the arena keeps
// [dense.bias, LayerNorm.weight, LayerNorm.bias] adjacent, so one k_reduce_multi job finishes all three
// (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------
// The incoming gradient is dYa (fp32, optional) + dYb (bf16, optional): the residual stream stays fp32 while the
// branch that comes out of a dgrad GEMM arrives as that GEMM's bf16 tile output (whole-row stores) instead of being
// added inside its epilogue (16-byte pieces of 32 rows per instruction, read and written in fp32).
__device__ __forceinline__ float4 grad_in(const float* __restrict__ a, const bf16_t* __restrict__ b, int64_t off) {
  float4 d = a ? *(const float4*)(a + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (b) {
    const uint2 v = *(const uint2*)(b + off);
    d.x += __uint_as_float(v.x << 16); d.y += __uint_as_float(v.x & 0xffff0000u);
    d.z += __uint_as_float(v.y << 16); d.w += __uint_as_float(v.y & 0xffff0000u);
  }
  return d;
}

__global__ void __launch_bounds__(256) k_layernorm_bwd(const float* __restrict__ dY, const bf16_t* __restrict__ dYb,
                                                       const float* __restrict__ Yin,
                                                       int64_t rows, int H, const float* __restrict__ g, float eps,
                                                       float* __restrict__ dXf, bf16_t* __restrict__ dXb,
                                                       float* __restrict__ part, const DropSite drop,
                                                       const int32_t* __restrict__ row_map) {
  // row_map (optional): the rows are a compact selection (the CLS rows of the last layer); the dropout mask of row r is
  // that of packed row row_map[r]
  // (cross-wave sums of the three column partials: one [3][1024] image that the four waves add into in turn.  The round-2
  //  form kept one image per wave, 48 KB: three workgroups per CU for a kernel that is bound by loads in flight -- 512
  //  workgroups measured 1.11 ms per configs[2] step, 768 0.94, and with 12 KB the launch below takes 1024.)
  __shared__ float red[3][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ag[4], ab[4], ax[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { ag[j] = make_float4(0, 0, 0, 0); ab[j] = make_float4(0, 0, 0, 0); ax[j] = make_float4(0, 0, 0, 0); }
  // A wave owns rows first, first + stride, ...: the loads of its NEXT row are issued before the arithmetic of the current
  // one (four dependent wave reductions: ~1.5 us of latency chain per row that used to sit between two ~2 us load round
  // trips; at the 9 k rows of a configs[2] step a wave has 4-5 rows and the kernel ran at 3.1 TB/s)
  const int64_t stride = (int64_t)gridDim.x * 4;
  float4 yn[4], dn[4];
  auto load_row = [&](int64_t row) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        yn[j] = *(const float4*)(Yin + row * H + e0);
        dn[j] = grad_in(dY, dYb, row * H + e0);
      }
    }
  };
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row < rows) load_row(row);
  for (; row < rows; row += stride) {
    float4 y[4], d[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        y[j] = yn[j];
        d[j] = dn[j];
        s += y[j].x + y[j].y + y[j].z + y[j].w;
      }
    }
    if (row + stride < rows) load_row(row + stride);
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (256 * j + 4 * lane < H) {
        y[j].x -= mean; y[j].y -= mean; y[j].z -= mean; y[j].w -= mean;
        q += y[j].x * y[j].x + y[j].y * y[j].y + y[j].z * y[j].z + y[j].w * y[j].w;
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        const float4 gg = *(const float4*)(g + e0);
        y[j].x *= rstd; y[j].y *= rstd; y[j].z *= rstd; y[j].w *= rstd;  // xhat
        ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
        ag[j].x += d[j].x * y[j].x; ag[j].y += d[j].y * y[j].y; ag[j].z += d[j].z * y[j].z; ag[j].w += d[j].w * y[j].w;
        d[j].x *= gg.x; d[j].y *= gg.y; d[j].z *= gg.z; d[j].w *= gg.w;  // g * dy
        m1 += d[j].x + d[j].y + d[j].z + d[j].w;
        m2 += d[j].x * y[j].x + d[j].y * y[j].y + d[j].z * y[j].z + d[j].w * y[j].w;
      }
    }
    m1 = wave_sum(m1) / (float)H;
    m2 = wave_sum(m2) / (float)H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        float4 o;
        o.x = rstd * (d[j].x - m1 - y[j].x * m2);
        o.y = rstd * (d[j].y - m1 - y[j].y * m2);
        o.z = rstd * (d[j].z - m1 - y[j].z * m2);
        o.w = rstd * (d[j].w - m1 - y[j].w * m2);
        if (dXf) *(float4*)(dXf + row * H + e0) = o;   // residual branch: not dropped
        if (drop.thresh) {   // the dense output feeding this LayerNorm went through dropout: its gradient (and its bias') is masked
          float k0, k1, k2, k3;
          drop_hidden4(drop, row_map ? (int64_t)row_map[row] : row, e0, H, k0, k1, k2, k3);
          o.x *= k0; o.y *= k1; o.z *= k2; o.w *= k3;
        }
        ax[j].x += o.x; ax[j].y += o.y; ax[j].z += o.z; ax[j].w += o.w;
        if (dXb) {
          uint2 p;
          p.x = pack_bf16x2(o.x, o.y);
          p.y = pack_bf16x2(o.z, o.w);
          *(uint2*)(dXb + row * H + e0) = p;
        }
      }
    }
  }
  auto f4add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
#pragma unroll
  for (int w = 0; w < 4; ++w) {   // fixed order (wave 0 + 1 + 2 + 3): deterministic
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e0 = 256 * j + 4 * lane;
        if (e0 < H) {
          if (w > 0) {
            f4add(ax[j], *(const float4*)&red[0][e0]);
            f4add(ag[j], *(const float4*)&red[1][e0]);
            f4add(ab[j], *(const float4*)&red[2][e0]);
          }
          if (w < 3) {
            *(float4*)&red[0][e0] = ax[j];
            *(float4*)&red[1][e0] = ag[j];
            *(float4*)&red[2][e0] = ab[j];
          } else {
            *(float4*)(part + ((int64_t)blockIdx.x * 3 + 0) * H + e0) = ax[j];
            *(float4*)(part + ((int64_t)blockIdx.x * 3 + 1) * H + e0) = ag[j];
            *(float4*)(part + ((int64_t)blockIdx.x * 3 + 2) * H + e0) = ab[j];
          }
        }
      }
    }
    if (w < 3) __syncthreads();
  }
}

// The encoder layers' form of the kernel above: H == 256 J exactly, both halves of the incoming gradient and both outputs
// present, rows indexed directly.  Same formulas in the same order (results equal up to hipcc's per-kernel choice of FMA
// contractions: tests/test_train_gpu.py compares the two inside one backward); what changes is the code shape.
// The general kernel guards every 16-byte group with `e0 < H` and every optional operand with a pointer test, and hipcc turns
// each guard into its own basic block with its own s_waitcnt: the 3 J loads of a row went out one dependent round trip after
// the other -- a latency chain per row, which is why a register prefetch of the NEXT row paid there.  Here the 3 J loads of a
// row are issued back to back (ISA: nine global_load in a row, counted vmcnt).  Where it showed: the second LayerNorm backward
// of a layer runs while the weight-gradient branch of the layer above owns 108 of the 256 CUs, the 768 workgroups (3 per CU at
// 168 VGPRs) need two rounds on the 148 left, and two rounds of latency chains were 45 us against 25 us for the same kernel
// one LayerNorm earlier (profiles/r05_train_kd.dispatches.txt).  Measured (profiles/r05_ab_ln_bwd_rows.txt): the 24 launches of
// a configs[2] step 0.94 -> 0.74 ms, the step -0.2 ms; PREFETCH (3 workgroups per CU) and no prefetch at WPE = 4 waves per SIMD
// (128 VGPRs, no spill; 5 or 6 would spill 40-100 registers) measure the same, as do grids of 384-768 workgroups.
template <int J, bool PREFETCH, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_layernorm_bwd_rows(const float* __restrict__ dY, const bf16_t* __restrict__ dYb,
                                                            const float* __restrict__ Yin, int64_t rows,
                                                            const float* __restrict__ g, float eps, float* __restrict__ dXf,
                                                            bf16_t* __restrict__ dXb, float* __restrict__ part,
                                                            const DropSite drop) {
  constexpr int H = 256 * J;
  __shared__ float red[3][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ag[J], ab[J], ax[J], gg[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    ag[j] = make_float4(0, 0, 0, 0); ab[j] = make_float4(0, 0, 0, 0); ax[j] = make_float4(0, 0, 0, 0);
    gg[j] = *(const float4*)(g + 256 * j + 4 * lane);
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  float4 y[J], d[J];
  uint2 b[J];
  float4 yn[J], dn[J];
  uint2 bn[J];
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (PREFETCH && row < rows) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t off = row * H + 256 * j + 4 * lane;
      yn[j] = *(const float4*)(Yin + off); dn[j] = *(const float4*)(dY + off); bn[j] = *(const uint2*)(dYb + off);
    }
  }
#pragma unroll 1
  for (; row < rows; row += stride) {
    if (PREFETCH) {
#pragma unroll
      for (int j = 0; j < J; ++j) { y[j] = yn[j]; d[j] = dn[j]; b[j] = bn[j]; }
      // the next row's loads go out before this row's four dependent wave reductions; past the end: this row again (the
      // address stays valid and no branch splits the block)
      const int64_t nrow = row + stride < rows ? row + stride : row;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int64_t off = nrow * H + 256 * j + 4 * lane;
        yn[j] = *(const float4*)(Yin + off); dn[j] = *(const float4*)(dY + off); bn[j] = *(const uint2*)(dYb + off);
      }
    } else {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int64_t off = row * H + 256 * j + 4 * lane;
        y[j] = *(const float4*)(Yin + off); d[j] = *(const float4*)(dY + off); b[j] = *(const uint2*)(dYb + off);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      d[j].x += __uint_as_float(b[j].x << 16); d[j].y += __uint_as_float(b[j].x & 0xffff0000u);
      d[j].z += __uint_as_float(b[j].y << 16); d[j].w += __uint_as_float(b[j].y & 0xffff0000u);
      s += y[j].x + y[j].y + y[j].z + y[j].w;
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      y[j].x -= mean; y[j].y -= mean; y[j].z -= mean; y[j].w -= mean;
      q += y[j].x * y[j].x + y[j].y * y[j].y + y[j].z * y[j].z + y[j].w * y[j].w;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      y[j].x *= rstd; y[j].y *= rstd; y[j].z *= rstd; y[j].w *= rstd;  // xhat
      ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
      ag[j].x += d[j].x * y[j].x; ag[j].y += d[j].y * y[j].y; ag[j].z += d[j].z * y[j].z; ag[j].w += d[j].w * y[j].w;
      d[j].x *= gg[j].x; d[j].y *= gg[j].y; d[j].z *= gg[j].z; d[j].w *= gg[j].w;  // g * dy
      m1 += d[j].x + d[j].y + d[j].z + d[j].w;
      m2 += d[j].x * y[j].x + d[j].y * y[j].y + d[j].z * y[j].z + d[j].w * y[j].w;
    }
    m1 = wave_sum(m1) / (float)H;
    m2 = wave_sum(m2) / (float)H;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int e0 = 256 * j + 4 * lane;
      float4 o;
      o.x = rstd * (d[j].x - m1 - y[j].x * m2);
      o.y = rstd * (d[j].y - m1 - y[j].y * m2);
      o.z = rstd * (d[j].z - m1 - y[j].z * m2);
      o.w = rstd * (d[j].w - m1 - y[j].w * m2);
      *(float4*)(dXf + row * H + e0) = o;   // residual branch: not dropped
      if (drop.thresh) {
        float k0, k1, k2, k3;
        drop_hidden4(drop, row, e0, H, k0, k1, k2, k3);
        o.x *= k0; o.y *= k1; o.z *= k2; o.w *= k3;
      }
      ax[j].x += o.x; ax[j].y += o.y; ax[j].z += o.z; ax[j].w += o.w;
      uint2 p;
      p.x = pack_bf16x2(o.x, o.y);
      p.y = pack_bf16x2(o.z, o.w);
      *(uint2*)(dXb + row * H + e0) = p;
    }
  }
  auto f4add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
#pragma unroll
  for (int w = 0; w < 4; ++w) {   // fixed order (wave 0 + 1 + 2 + 3): deterministic
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int e0 = 256 * j + 4 * lane;
        if (w > 0) {
          f4add(ax[j], *(const float4*)&red[0][e0]);
          f4add(ag[j], *(const float4*)&red[1][e0]);
          f4add(ab[j], *(const float4*)&red[2][e0]);
        }
        if (w < 3) {
          *(float4*)&red[0][e0] = ax[j];
          *(float4*)&red[1][e0] = ag[j];
          *(float4*)&red[2][e0] = ab[j];
        } else {
          *(float4*)(part + ((int64_t)blockIdx.x * 3 + 0) * H + e0) = ax[j];
          *(float4*)(part + ((int64_t)blockIdx.x * 3 + 1) * H + e0) = ag[j];
          *(float4*)(part + ((int64_t)blockIdx.x * 3 + 2) * H + e0) = ab[j];
        }
      }
    }
    if (w < 3) __syncthreads();
  }
}

// out[e] (+)= sum_p part[p * stride + e], p in fixed order (deterministic).  16 bytes per thread, four partials in
// flight per accumulator chain (the one-load-at-a-time form of round 1 spent its time in dependent L2 round trips).
__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ part, int nparts, int64_t stride,
                                                         int64_t n, float* __restrict__ out, int accumulate,
                                                         const float* __restrict__ addend = nullptr) {
  // addend (optional, with accumulate = 0): out = addend + sum of the partials
  const int64_t n4 = n >> 2;   // n % 4 == 0 (weight matrices)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = 0;
    for (; p + 4 <= nparts; p += 4) {
      const float4 a = *(const float4*)(part + (int64_t)p * stride + 4 * i);
      const float4 b = *(const float4*)(part + (int64_t)(p + 1) * stride + 4 * i);
      const float4 c = *(const float4*)(part + (int64_t)(p + 2) * stride + 4 * i);
      const float4 d = *(const float4*)(part + (int64_t)(p + 3) * stride + 4 * i);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; p < nparts; ++p) {
      const float4 a = *(const float4*)(part + (int64_t)p * stride + 4 * i);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    if (accumulate || addend) {
      const float4 o = *(const float4*)((accumulate ? out : addend) + 4 * i);
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *(float4*)(out + 4 * i) = s;
  }
}

// column sums of a bf16 matrix [rows, C] (C % 8 == 0): part[chunk][C]; grid (ceil(C / 256), chunks).
// 256 threads = 32 column groups of 8 (one 16-byte load each) x 8 row lanes, two rows in flight per lane; each block sums
// its row chunk, then folds the 8 row lanes in LDS in a fixed order.
static __global__ void __launch_bounds__(256) k_colsum_bf16(const bf16_t* __restrict__ in, int64_t rows, int C,
                                                            float* __restrict__ part) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + 8 * cg;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t t0 = per * blockIdx.y, t1 = t0 + per < rows ? t0 + per : rows;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  auto add = [&](const uint4& v) {
    s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
    s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
    s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
    s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
  };
  if (c < C) {
    int64_t t = t0 + rl;
    for (; t + 8 < t1; t += 16) {
      const uint4 a = *(const uint4*)(in + t * C + c), b = *(const uint4*)(in + (t + 8) * C + c);
      add(a); add(b);
    }
    for (; t < t1; t += 8) add(*(const uint4*)(in + t * C + c));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][8 * cg + j] = s[j];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    part[(int64_t)blockIdx.y * C + cc] = t;
  }
}

// FFN1 activation backward fused with the bias column sums:  g[t, c] <- g[t, c] * gelu'(pre[t, c])  (in place: g is the
// bf16 tile output of the FFN2 dgrad GEMM), part[chunk][C] = column sums of the result (= d b1).  Same geometry as
// k_colsum_bf16.  As a separate memory-bound pass the ~30 VALU slots per element of gelu' run under the loads; inside
// the GEMM epilogue they ran with the matrix pipe idle (126 -> ~55 + ~40 us per layer at 9 k rows).
static __global__ void __launch_bounds__(256) k_dgelu_colsum(bf16_t* __restrict__ g, const bf16_t* __restrict__ pre, int64_t rows,
                                                             int C, float* __restrict__ part) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + 8 * cg;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t t0 = per * blockIdx.y, t1 = t0 + per < rows ? t0 + per : rows;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  auto one = [&](const uint4& gv, const uint4& pv, int64_t t) {
    const uint32_t* a = (const uint32_t*)&gv;
    const uint32_t* b = (const uint32_t*)&pv;
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float lo = __uint_as_float(a[q] << 16) * gelu_grad(__uint_as_float(b[q] << 16));
      const float hi = __uint_as_float(a[q] & 0xffff0000u) * gelu_grad(__uint_as_float(b[q] & 0xffff0000u));
      o[q] = pack_bf16x2(lo, hi);
      s[2 * q] += __uint_as_float(o[q] << 16);          // sums of the ROUNDED values: db1 matches the operand wgrad reads
      s[2 * q + 1] += __uint_as_float(o[q] & 0xffff0000u);
    }
    *(uint4*)(g + t * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
  };
  if (c < C) {
    int64_t t = t0 + rl;
    for (; t + 8 < t1; t += 16) {
      const uint4 g0 = *(const uint4*)(g + t * C + c), p0 = *(const uint4*)(pre + t * C + c);
      const uint4 g1 = *(const uint4*)(g + (t + 8) * C + c), p1 = *(const uint4*)(pre + (t + 8) * C + c);
      one(g0, p0, t);
      one(g1, p1, t + 8);
    }
    for (; t < t1; t += 8) one(*(const uint4*)(g + t * C + c), *(const uint4*)(pre + t * C + c), t);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][8 * cg + j] = s[j];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    part[(int64_t)blockIdx.y * C + cc] = t;
  }
}

// Several small partial-sum reductions in one launch: 16 outputs per block, the partials split over 16 thread rows (four
// loads in flight each) and folded in LDS in a fixed order -- deterministic (all of a layer's LayerNorm / bias partial sums are finished by ONE
// kernel on the weight-gradient stream instead of one launch each on the activation-gradient chain).
constexpr int REDUCE_MAX_JOBS = 8;
struct ReduceJob {
  const float* part;
  float* out;
  int64_t stride;
  int nparts, n;
  int block_end;    // one past this job's last block (16 outputs per block)
};
struct ReduceJobs {
  ReduceJob j[REDUCE_MAX_JOBS];
  int count;
  int overwrite;    // out = sum instead of out += sum (convdr_encoder_backward_fresh)
};
static __global__ void __launch_bounds__(256) k_reduce_multi(const ReduceJobs a) {
  __shared__ float red[16][17];
  int ji = 0;
#pragma unroll
  for (int i = 0; i < REDUCE_MAX_JOBS - 1; ++i)
    if (i + 1 < a.count && (int)blockIdx.x >= a.j[i].block_end) ji = i + 1;
  const ReduceJob& q = a.j[ji];
  const int blk = (int)blockIdx.x - (ji ? a.j[ji - 1].block_end : 0);
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int64_t i = (int64_t)blk * 16 + c;
  float s = 0.f;
  if (i < q.n) {
    int p = r;
    for (; p + 48 < q.nparts; p += 64) {
      const float v0 = q.part[(int64_t)p * q.stride + i], v1 = q.part[(int64_t)(p + 16) * q.stride + i];
      const float v2 = q.part[(int64_t)(p + 32) * q.stride + i], v3 = q.part[(int64_t)(p + 48) * q.stride + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < q.nparts; p += 16) s += q.part[(int64_t)p * q.stride + i];
  }
  red[r][c] = s;
  __syncthreads();
  if (r == 0 && i < q.n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    q.out[i] = a.overwrite ? t : q.out[i] + t;
  }
}

// backward of k_masked_mean: dX[row, :] = dpool[b, :] / len[b] for the sequence's tokens, 0 for its alignment rows
static __global__ void __launch_bounds__(256) k_masked_mean_bwd(const float* __restrict__ dpool, const int32_t* __restrict__ cu,
                                                                const int32_t* __restrict__ lens, int H, float* __restrict__ dX) {
  const int b = blockIdx.x, c = 4 * threadIdx.x;
  if (c >= H) return;
  const int64_t base = cu[b], end = cu[b + 1];
  const int len = lens[b];
  const float inv = 1.f / (float)len;
  float4 g = *(const float4*)(dpool + (int64_t)b * H + c);
  g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
  for (int64_t r = base; r < end; ++r)
    *(float4*)(dX + r * H + c) = r < base + len ? g : make_float4(0.f, 0.f, 0.f, 0.f);
}

// y = bf16(x * dropout mask) for a [rows, H] matrix: the last layer's FFN-output gradient (only its CLS rows are non-zero)
// on its way to the FFN2 dgrad / wgrad operands when hidden dropout is on
static __global__ void __launch_bounds__(256) k_cast_drop_f32_bf16(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t rows,
                                                                   int H, const DropSite drop, const int32_t* __restrict__ row_map) {
  const int64_t n4 = rows * H / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = (4 * i) / H;
    const int col = (int)(4 * i - row * H);
    float4 v = *(const float4*)(x + 4 * i);
    float m0, m1, m2, m3;
    drop_hidden4(drop, row_map ? (int64_t)row_map[row] : row, col, H, m0, m1, m2, m3);
    uint2 o;
    o.x = pack_bf16x2(v.x * m0, v.y * m1);
    o.y = pack_bf16x2(v.z * m2, v.w * m3);
    *(uint2*)(y + 4 * i) = o;
  }
}

// Both gathers of the last layer's CLS tail in one launch: ctx_c[b] = ctx[cu[b]], xin_c[b] = xin[cu[b]] (bf16 rows).
__global__ void __launch_bounds__(256) k_gather_cls2(const int32_t* __restrict__ cu, int B, int H, const bf16_t* __restrict__ a,
                                                     const bf16_t* __restrict__ b, bf16_t* __restrict__ oa, bf16_t* __restrict__ ob) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int64_t row = cu[i];
  for (int e0 = 8 * lane; e0 < H; e0 += 512) {
    *(uint4*)(oa + (int64_t)i * H + e0) = *(const uint4*)(a + row * H + e0);
    *(uint4*)(ob + (int64_t)i * H + e0) = *(const uint4*)(b + row * H + e0);
  }
}

// The hand-over from the last layer's CLS-row backward to the full-size kernels, one launch instead of three memsets and two
// scatters.  For sequence b (rows [cu[b], cu[b + 1])):
//   dctx  [rows, H]  bf16: row cu[b] = dctx_c[b]; the other rows the attention backward reads (its first query tile: 128) = 0
//   dQKV  [rows, 3H] bf16: the Q third of the rows beyond that tile = 0 (the dQ kernel writes only its tile's rows)
//   G     [rows, H]  fp32: row cu[b] = dY1_c[b], every other row of the sequence = 0 (the residual-branch gradient below)
// grid (B, 8): the rows of a sequence are dealt to 8 blocks, one wave per row.
__global__ void __launch_bounds__(256) k_cls_tail_scatter(const int32_t* __restrict__ cu, int B, int H, const bf16_t* __restrict__ dctx_c,
                                                          bf16_t* __restrict__ dctx, bf16_t* __restrict__ dQKV,
                                                          const float* __restrict__ dY1_c, float* __restrict__ G,
                                                          int dq_rows) {
  // dq_rows: rows of a sequence whose dQ the attention backward that follows writes itself (its first query tile: 128 for the
  // dQ kernel, 64 for the one-workgroup form)
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const int64_t base = cu[b], end = cu[b + 1];
  for (int64_t row = base + blockIdx.y * 4 + (threadIdx.x >> 6); row < end; row += 4 * gridDim.y) {
    const bool cls = row == base;
    for (int e0 = 4 * lane; e0 < H; e0 += 256)
      *(float4*)(G + row * H + e0) = cls ? *(const float4*)(dY1_c + (int64_t)b * H + e0) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < base + 128)
      for (int e0 = 8 * lane; e0 < H; e0 += 512)
        *(uint4*)(dctx + row * H + e0) = cls ? *(const uint4*)(dctx_c + (int64_t)b * H + e0) : make_uint4(0u, 0u, 0u, 0u);
    if (row >= base + dq_rows)
      for (int e0 = 8 * lane; e0 < H; e0 += 512) *(uint4*)(dQKV + row * 3 * H + e0) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// Finish of a split-K projection on a few compact rows (the CLS rows of the last layer, training forward):
//   Y[r, f] = dropout(sum_s slab[s][r][f] + bias[f]; packed row row_map[r]) + resid[r, f]
// i.e. exactly what the EPI_RESID_F32 epilogue computes for a whole-contraction tile.  One thread per 4 features.
__global__ void __launch_bounds__(256) k_slab_finish(const float* __restrict__ slab, int nsplit, int rows, int N,
                                                     const float* __restrict__ bias, const bf16_t* __restrict__ resid,
                                                     const int32_t* __restrict__ row_map, const DropSite drop,
                                                     float* __restrict__ Y) {
  const int64_t n4 = (int64_t)rows * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int r = (int)((4 * i) / N), f = (int)(4 * i - (int64_t)r * N);
    float4 y = bias ? *(const float4*)(bias + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nsplit; ++s) {
      const float4 v = *(const float4*)(slab + ((int64_t)s * rows + r) * N + f);
      y.x += v.x; y.y += v.y; y.z += v.z; y.w += v.w;
    }
    if (drop.thresh) {
      float m0, m1, m2, m3;
      drop_hidden4(drop, row_map ? (int64_t)row_map[r] : (int64_t)r, f, N, m0, m1, m2, m3);
      y.x *= m0; y.y *= m1; y.z *= m2; y.w *= m3;
    }
    if (resid) {
      const uint2 q = *(const uint2*)(resid + (int64_t)r * N + f);
      y.x += __uint_as_float(q.x << 16); y.y += __uint_as_float(q.x & 0xffff0000u);
      y.z += __uint_as_float(q.y << 16); y.w += __uint_as_float(q.y & 0xffff0000u);
    }
    *(float4*)(Y + (int64_t)r * N + f) = y;
  }
}

// ---------------------------------------------------------------------------------------------
// embeddings backward: dX0 -> LayerNorm backward (embedding sum recomputed from the tables) -> atomic adds into
// d_word[id], d_pos[p]; d_type[0] and the LayerNorm dgamma / dbeta go through per-block partials.
// part[block][3][H] = (dgamma, dbeta, dtype0)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_embed_bwd(const float* __restrict__ dX, const bf16_t* __restrict__ dXb,
                                                   const int32_t* __restrict__ tok_id,
                                                   const int32_t* __restrict__ tok_pos, int64_t rows, int H,
                                                   const float* __restrict__ word, const float* __restrict__ pos,
                                                   const float* __restrict__ type0, const float* __restrict__ g,
                                                   float eps, float* __restrict__ d_word, float* __restrict__ d_pos,
                                                   float* __restrict__ part, const DropSite drop, float* __restrict__ o_rows) {
  // o_rows non-null (convdr_set_option "embed_bwd_deterministic"): the row's gradient is written to o_rows[row] instead of
  // being added into the tables with atomics; k_embed_scatter_det then adds the rows in row order (below)
  __shared__ float red[4][3][1024];
  // the wave-private staging strip of the atomics lives in the wave's own slice of `red`, which is written only after the
  // row loop (the two arrays together were exactly the 64 KB static-LDS limit)
  float* const stage_w = &red[threadIdx.x >> 6][0][0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ag[4], ab[4], at[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { ag[j] = make_float4(0, 0, 0, 0); ab[j] = ag[j]; at[j] = ag[j]; }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int id = tok_id[row];
    if (id < 0) continue;  // alignment row: no parameters behind it
    const int pp = tok_pos[row];
    float4 y[4], d[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        const float4 a = *(const float4*)(word + (int64_t)id * H + e0), c = *(const float4*)(pos + (int64_t)pp * H + e0),
                     t = *(const float4*)(type0 + e0);
        y[j] = make_float4(a.x + c.x + t.x, a.y + c.y + t.y, a.z + c.z + t.z, a.w + c.w + t.w);
        d[j] = grad_in(dX, dXb, row * H + e0);
        if (drop.thresh) {   // embedding dropout sits between this LayerNorm and the first layer
          float k0, k1, k2, k3;
          drop_hidden4(drop, row, e0, H, k0, k1, k2, k3);
          d[j].x *= k0; d[j].y *= k1; d[j].z *= k2; d[j].w *= k3;
        }
        s += y[j].x + y[j].y + y[j].z + y[j].w;
      }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (256 * j + 4 * lane < H) {
        y[j].x -= mean; y[j].y -= mean; y[j].z -= mean; y[j].w -= mean;
        q += y[j].x * y[j].x + y[j].y * y[j].y + y[j].z * y[j].z + y[j].w * y[j].w;
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        const float4 gg = *(const float4*)(g + e0);
        y[j].x *= rstd; y[j].y *= rstd; y[j].z *= rstd; y[j].w *= rstd;
        ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
        ag[j].x += d[j].x * y[j].x; ag[j].y += d[j].y * y[j].y; ag[j].z += d[j].z * y[j].z; ag[j].w += d[j].w * y[j].w;
        d[j].x *= gg.x; d[j].y *= gg.y; d[j].z *= gg.z; d[j].w *= gg.w;
        m1 += d[j].x + d[j].y + d[j].z + d[j].w;
        m2 += d[j].x * y[j].x + d[j].y * y[j].y + d[j].z * y[j].z + d[j].w * y[j].w;
      }
    }
    m1 = wave_sum(m1) / (float)H;
    m2 = wave_sum(m2) / (float)H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = 256 * j + 4 * lane;
      if (e0 < H) {
        float4 o;
        o.x = rstd * (d[j].x - m1 - y[j].x * m2);
        o.y = rstd * (d[j].y - m1 - y[j].y * m2);
        o.z = rstd * (d[j].z - m1 - y[j].z * m2);
        o.w = rstd * (d[j].w - m1 - y[j].w * m2);
        at[j].x += o.x; at[j].y += o.y; at[j].z += o.z; at[j].w += o.w;
        if (o_rows) *(float4*)(o_rows + row * H + e0) = o;
        else *(float4*)&stage_w[e0] = o;
      }
    }
    if (o_rows) continue;   // (wave-uniform)
    // one atomic instruction = 64 consecutive floats (two whole 128-byte lines), not 64 floats 16 bytes apart (eight
    // quarter-filled lines): the row goes through a wave-private LDS strip to change the lane -> column map.  Measured
    // on a configs[2] step: the kernel's 14 M lane-atomics cost 0.2 ms in the strided form and nothing measurable here.
    float* dw = d_word + (int64_t)id * H;
    float* dp = d_pos + (int64_t)pp * H;
    for (int e = lane; e < H; e += 64) {
      const float o = stage_w[e];
      atomicAdd(dw + e, o);
      atomicAdd(dp + e, o);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = 256 * j + 4 * lane;
    if (e0 < H) {
      *(float4*)&red[wave][0][e0] = ag[j];
      *(float4*)&red[wave][1][e0] = ab[j];
      *(float4*)&red[wave][2][e0] = at[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * H; i += 256) {
    const int k = i / H, e = i - k * H;
    part[((int64_t)blockIdx.x * 3 + k) * H + e] = red[0][k][e] + red[1][k][e] + red[2][k][e] + red[3][k][e];
  }
}

// Deterministic form of the scatter above (the "deterministic re-run diff" is this build's race detector, and
// fp32 atomics made the word / position gradients the two tensors it could not cover: 3e-5 of 495 between identical runs).
// Workgroup w OWNS the table rows [w * per, (w + 1) * per): it walks the token rows in ascending order, 256 at a time, compacts
// the ones that hit its range in order (ballot + prefix), and adds them to the table row by row -- element e of a table row is
// only ever touched by thread e % 256 of its owner, in token-row order: no atomics, one fixed summation order.
// Called once for the word table (ids = tok_id, alignment rows carry -1) and once for the position table.
__global__ void __launch_bounds__(256) k_embed_scatter_det(const float* __restrict__ o_rows, const int32_t* __restrict__ ids,
                                                           const int32_t* __restrict__ valid, int64_t rows, int H, int n_table,
                                                           float* __restrict__ d_table) {
  __shared__ int32_t list_row[256], list_id[256], wave_cnt[4];
  const int per = (n_table + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = lo + per < n_table ? lo + per : n_table;
  if (lo >= hi) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r0 = 0; r0 < rows; r0 += 256) {
    const int64_t row = r0 + threadIdx.x;
    int id = -1;
    if (row < rows && valid[row] >= 0) id = ids[row];
    const bool hit = id >= lo && id < hi;
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = __popcll(bal & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wave_cnt[w];
      before += w < wave ? c : 0;
      total += c;
    }
    if (hit) { list_row[before] = (int32_t)(row - r0); list_id[before] = id; }
    __syncthreads();
    for (int i = 0; i < total; ++i) {
      const float* src = o_rows + (r0 + list_row[i]) * H;
      float* dst = d_table + (int64_t)list_id[i] * H;
      for (int e = threadIdx.x; e < H; e += 256) dst[e] += src[e];
    }
    __syncthreads();   // the lists are rewritten by the next chunk
  }
}

// ---------------------------------------------------------------------------------------------
// attention backward
// ---------------------------------------------------------------------------------------------
// D[h, t] = sum_d dO[t, 64 h + d] * O[t, 64 h + d]; one wave per token (lane owns 256 j + 4 lane + c: each
// 16-lane group covers one head per j)
__global__ void __launch_bounds__(256) k_attn_rowdot(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                     int64_t rows, int H, float* __restrict__ D, int64_t ldt) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int e0 = 4 * lane; e0 < H; e0 += 256) {
    const uint2 a = *(const uint2*)(dO + row * H + e0), b = *(const uint2*)(O + row * H + e0);
    float s = __uint_as_float(a.x << 16) * __uint_as_float(b.x << 16) +
              __uint_as_float(a.x & 0xffff0000u) * __uint_as_float(b.x & 0xffff0000u) +
              __uint_as_float(a.y << 16) * __uint_as_float(b.y << 16) +
              __uint_as_float(a.y & 0xffff0000u) * __uint_as_float(b.y & 0xffff0000u);
    s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    if ((lane & 15) == 0) D[(int64_t)(e0 >> 6) * ldt + row] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// losses (run_convdr_train.py:114-115, :160-171) and the optimizer (utils/dpr_utils.py:80-87 -> HF AdamW)
// ---------------------------------------------------------------------------------------------
// loss = mean((s - t)^2) over n elements; ds = 2 (s - t) / n * gscale.  Single workgroup: n = B * 768 is small.
__global__ void __launch_bounds__(1024) k_mse_fwd_bwd(const float* __restrict__ s, const float* __restrict__ t,
                                                      int64_t n, float gscale, float* __restrict__ loss,
                                                      float* __restrict__ ds) {
  __shared__ float red[16];
  float acc = 0.f;
  const float k = 2.f / (float)n * gscale;
  // one workgroup (the loss is one scalar): 16-byte accesses, four of them in flight per thread -- the scalar loop was a
  // 48-deep chain of dependent round trips at the configs[2] size (27 us on the step's critical path)
  const bool vec = ((((uintptr_t)s) | ((uintptr_t)t) | ((uintptr_t)ds)) & 15) == 0;
  const int64_t n4 = vec ? (n >> 2) : 0;
#pragma unroll 4
  for (int64_t i = threadIdx.x; i < n4; i += 1024) {
    const float4 a = ((const float4*)s)[i], b = ((const float4*)t)[i];
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    if (ds) ((float4*)ds)[i] = make_float4(k * d.x, k * d.y, k * d.z, k * d.w);
  }
  for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 1024) {
    const float d = s[i] - t[i];
    acc += d * d;
    if (ds) ds[i] = k * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += red[i];
    *loss = tot / (float)n;
  }
}

// ranking loss: logits[b, k] = <e_b, d_{b,k}>, loss = mean_b(-log_softmax(logits[b])[0]);
// de_b (+)= gscale / B * sum_k (softmax_k - [k == 0]) d_{b,k}.  One workgroup (256 threads) per b; K <= 64.
__global__ void __launch_bounds__(256) k_rank_ce_fwd_bwd(const float* __restrict__ e, const float* __restrict__ docs,
                                                         int B, int K, int E, float gscale,
                                                         float* __restrict__ loss_per_b, float* __restrict__ de,
                                                         int accumulate) {
  __shared__ float logit[64];
  __shared__ float prob[64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* eb = e + (int64_t)b * E;
  for (int k = wave; k < K; k += 4) {
    const float* d = docs + ((int64_t)b * K + k) * E;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += eb[i] * d[i];
    s = wave_sum(s);
    if (lane == 0) logit[k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mx = logit[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, logit[k]);
    float z = 0.f;
    for (int k = 0; k < K; ++k) z += expf(logit[k] - mx);
    const float lz = logf(z) + mx;
    for (int k = 0; k < K; ++k) prob[k] = expf(logit[k] - lz);
    loss_per_b[b] = lz - logit[0];
  }
  __syncthreads();
  if (de) {
    const float sc = gscale / (float)B;
    for (int i = threadIdx.x; i < E; i += 256) {
      float g = 0.f;
      for (int k = 0; k < K; ++k) g += (prob[k] - (k == 0 ? 1.f : 0.f)) * docs[((int64_t)b * K + k) * E + i];
      g *= sc;
      de[(int64_t)b * E + i] = accumulate ? de[(int64_t)b * E + i] + g : g;
    }
  }
}

// Pairwise NLL of the model surface: NLL.forward's triple branch (models.py:66-75) and NLL_MultiChunk's MaxP form
// (models.py:92-126) in one kernel.  Per query b, with C chunks per document (C = 1 and no bias: the plain pairwise form):
//   s_x = max_c ( <q[b], x[b, c]> + bias_x[b, c] ),  x in {a, b}        (first maximal chunk, like torch.max)
//   loss[b] = -log_softmax([s_a, s_b])[0]
//   dq[b] = g ((p_a - 1) a[b, c*_a] + p_b b[b, c*_b]),  da[b, c*_a] = g (p_a - 1) q[b],  db[b, c*_b] = g p_b q[b],  g = gscale / B,
//   zero for every other chunk.  One workgroup per query; C <= 32.
__global__ void __launch_bounds__(256) k_pair_nll_fwd_bwd(const float* __restrict__ q, const float* __restrict__ a,
                                                          const float* __restrict__ b, const float* __restrict__ bias_a,
                                                          const float* __restrict__ bias_b, int B, int C, int E, float gscale,
                                                          float* __restrict__ loss_per_b, float* __restrict__ dq,
                                                          float* __restrict__ da, float* __restrict__ db) {
  __shared__ float sc[2][32];
  __shared__ float pr[2];
  __shared__ int arg[2];
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qi = q + (int64_t)i * E;
  for (int k = wave; k < 2 * C; k += 4) {
    const int side = k / C, c = k - side * C;
    const float* d = (side ? b : a) + ((int64_t)i * C + c) * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += qi[e] * d[e];
    s = wave_sum(s);
    const float* bias = side ? bias_b : bias_a;
    if (lane == 0) sc[side][c] = s + (bias ? bias[(int64_t)i * C + c] : 0.f);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float best[2];
    for (int side = 0; side < 2; ++side) {
      int am = 0;
      for (int c = 1; c < C; ++c)
        if (sc[side][c] > sc[side][am]) am = c;
      arg[side] = am;
      best[side] = sc[side][am];
    }
    const float mx = fmaxf(best[0], best[1]);
    const float lz = logf(expf(best[0] - mx) + expf(best[1] - mx)) + mx;
    pr[0] = expf(best[0] - lz);
    pr[1] = expf(best[1] - lz);
    loss_per_b[i] = lz - best[0];
  }
  __syncthreads();
  const float g = gscale / (float)B;
  const float ca = g * (pr[0] - 1.f), cb = g * pr[1];
  const float* as = a + ((int64_t)i * C + arg[0]) * E;
  const float* bs = b + ((int64_t)i * C + arg[1]) * E;
  if (dq)
    for (int e = threadIdx.x; e < E; e += 256) dq[(int64_t)i * E + e] = ca * as[e] + cb * bs[e];
  for (int k = 0; k < C; ++k) {
    if (da)
      for (int e = threadIdx.x; e < E; e += 256) da[((int64_t)i * C + k) * E + e] = k == arg[0] ? ca * qi[e] : 0.f;
    if (db)
      for (int e = threadIdx.x; e < E; e += 256) db[((int64_t)i * C + k) * E + e] = k == arg[1] ? cb * qi[e] : 0.f;
  }
}

// In-batch-negative ranking loss (BASELINE configs[4]; not in the reference -- oracle/train.py:inbatch_rank_loss is the
// definition): every query scores ALL N gathered documents, the target is the index of its own positive.
//   logits[n] = <e_b, docs[n]>,  loss_b = logsumexp(logits) - logits[pos_b],
//   de_b = gscale / B * sum_n (softmax_n - [n == pos_b]) docs[n]
// One workgroup per query; logits live in LDS (dynamic, N floats).  Summation orders are fixed (deterministic).
__global__ void __launch_bounds__(256) k_inbatch_ce_fwd_bwd(const float* __restrict__ e, const float* __restrict__ docs,
                                                            int B, int N, int E, const int32_t* __restrict__ pos,
                                                            float gscale, float* __restrict__ loss_per_b,
                                                            float* __restrict__ de, int accumulate) {
  extern __shared__ float lg[];          // [N] logits, then probabilities
  __shared__ float red[8];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* eb = e + (int64_t)b * E;
  for (int n0 = wave * 4; n0 < N; n0 += 16) {   // four documents per wave per round: independent loads in flight
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < E; i += 64) {
      const float q = eb[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += q * docs[(int64_t)(n0 + j < N ? n0 + j : N - 1) * E + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = wave_sum(s[j]);
      if (lane == 0 && n0 + j < N) lg[n0 + j] = t;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int n = threadIdx.x; n < N; n += 256) mx = fmaxf(mx, lg[n]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float z = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) z += expf(lg[n] - mx);
  z = wave_sum(z);
  if (lane == 0) red[4 + wave] = z;
  __syncthreads();
  const float lz = logf((red[4] + red[5]) + (red[6] + red[7])) + mx;
  const int p = pos[b];
  if (threadIdx.x == 0) loss_per_b[b] = lz - lg[p];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) lg[n] = expf(lg[n] - lz) - (n == p ? 1.f : 0.f);
  __syncthreads();
  if (de) {
    const float sc = gscale / (float)B;
    for (int i = threadIdx.x; i < E; i += 256) {
      float g = 0.f;
#pragma unroll 8
      for (int n = 0; n < N; ++n) g += lg[n] * docs[(int64_t)n * E + i];   // fixed order: deterministic
      g *= sc;
      de[(int64_t)b * E + i] = accumulate ? de[(int64_t)b * E + i] + g : g;
    }
  }
}

// sum of squares, two stages (deterministic): part[block]
__global__ void __launch_bounds__(256) k_sumsq_partial(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  // 16 bytes per lane, four loads in flight per thread (the scalar form streamed the 0.5 GB gradient arena at 2.4 TB/s)
  const int64_t n4 = ((uintptr_t)x & 15) == 0 ? n >> 2 : 0;
  const int64_t step4 = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step4 < n4; i += 4 * step4) {
    const float4 a = *(const float4*)(x + 4 * i), b = *(const float4*)(x + 4 * (i + step4)),
                 c = *(const float4*)(x + 4 * (i + 2 * step4)), d = *(const float4*)(x + 4 * (i + 3 * step4));
    acc += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w) +
           (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) + (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
  }
  for (; i < n4; i += step4) {
    const float4 a = *(const float4*)(x + 4 * i);
    acc += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  for (int64_t t = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += step4) acc += x[t] * x[t];   // tail / unaligned
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// norm_out[0] = pre_scale * sqrt(sum part); norm_out[1] = pre_scale * min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_
// of the gradients scaled by pre_scale)
__global__ void k_norm_finish(const float* __restrict__ part, int nparts, float max_norm, float pre_scale,
                              float* __restrict__ out) {
  // one wave; lane l folds partials l, l + 64, ... in fp64, then a fixed butterfly (deterministic)
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) s += (double)part[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float nm = pre_scale * (float)sqrt(s);   // norm of the gradients AFTER the pending scale (1 / world size)
    out[0] = nm;
    const float cf = max_norm / (nm + 1e-6f);
    out[1] = pre_scale * (cf < 1.f ? cf : 1.f);    // what the raw gradients still have to be multiplied by
  }
}

// HF transformers==2.3.0 AdamW (not torch.optim.AdamW):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps);  p -= lr wd p
// g is multiplied by *gscale (the clip coefficient, device scalar) first.
// pb (optional): the packed bf16 copy of the weights that the GEMMs read, pb[i - pb_first] = bf16(p[i]) for i >= pb_first
// (pb_first % 4 == 0): written here, from the registers that hold the new weight, instead of by a cast pass over the arena at
// the top of the next step (0.5 GB read + 0.17 GB written, 93 us of the configs[2] step's serial tail).
__global__ void __launch_bounds__(256) k_adamw_hf(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                  float b1, float b2, float eps, float wd, float step_size,
                                                  const float* __restrict__ gscale, bf16_t* __restrict__ pb,
                                                  int64_t pb_first) {
  const float gs = gscale ? gscale[0] : 1.f;
  auto upd = [&](float& pi, const float g0, float& mi, float& vi) {
    const float gi = g0 * gs;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi = pi - step_size * mi / (sqrtf(vi) + eps);
    if (wd > 0.f) pi -= lr * wd * pi;
  };
  // 16-byte accesses when the four arrays allow it (the flat arenas always do): 28 bytes per element of pure streaming
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const int64_t n4 = vec ? (n >> 2) : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = ((const float4*)p)[i], mm = ((const float4*)m)[i], vv = ((const float4*)v)[i];
    const float4 gg = ((const float4*)g)[i];
    upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
    ((float4*)m)[i] = mm; ((float4*)v)[i] = vv; ((float4*)p)[i] = pp;
    if (pb && 4 * i >= pb_first) {
      uint2 o;
      o.x = pack_bf16x2(pp.x, pp.y);
      o.y = pack_bf16x2(pp.z, pp.w);
      *(uint2*)(pb + (4 * i - pb_first)) = o;
    }
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (pb && i >= pb_first) pb[i - pb_first] = f32_to_bf16(pi);
  }
}

__global__ void __launch_bounds__(256) k_scale_inplace(float* __restrict__ x, int64_t n, const float* __restrict__ s) {
  const float f = s[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= f;
}

}  // namespace convdr
