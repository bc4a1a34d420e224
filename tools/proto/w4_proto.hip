// Prototype (NOT product): main loop of a 256 x 256 tile with ONE wave per SIMD (4 waves, 128 x 128 per wave, 256
// accumulator registers of a 512-register budget) against the product geometry (8 waves, 128 x 64 per wave), same R3
// K step (3 R slots + 2 L slots), FFN1 shape, no epilogue (the accumulators are folded into one store so that nothing
// is dead code).  Question: is the one-wave-per-SIMD loop within ~10 % of the two-wave loop?  If so the previous tile's
// epilogue can be interleaved into its MFMA shadow (512 registers hold both tiles); round 1 measured -40 % for a
// straightforward port of the two-stage loop.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I convdr_amd/csrc tools/proto/w4_proto.hip -o gpurun_out/w4_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_nt.hpp"
using namespace convdr;

struct Args { const bf16_t* W; const bf16_t* X; float* Y; int64_t rows; int N, K; int tilesN, tilesT; unsigned long long* trace; };

// VARIANT 0: DMA blocks as in gemm_nt_mainloop_r3 (L chunk after the first fragment reads, R chunk after the MFMAs)
// VARIANT 1: the DMA instructions ride between the MFMAs of sub-steps 0 (L) and 2-3 (R), one per MFMA pair
using T8 = TileCfg<2, 4, 4, 2>;
using T4 = TileCfg<2, 2, 4, 4>;
template <int W8> struct Pick { using type = T8; };
template <> struct Pick<0> { using type = T4; };
template <int W8, int VARIANT>
__device__ __forceinline__ void loop_body(const Args& a) {
  using T = typename Pick<W8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sR = smem;
  char* sL = smem + 3 * T::R_BYTES;
  const WavePos<T> w;
  const uint32_t ntiles = (uint32_t)a.tilesN * a.tilesT;
  const int sw = (w.lane >> 1) & 7;
  const int offR = (w.wr * T::MT * 32 + w.li) * 128;
  const int offL = (w.wl * T::NT * 32 + w.li) * 128;
  constexpr int R_DPW = T::TR / (8 * T::WAVES), L_DPW = T::TL / (8 * T::WAVES);
  int rs = 0, ls = 0;
  float sink = 0.f;
  int tcount = 0;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++tcount) {
    const int tt = tile / a.tilesN, tn = tile - tt * a.tilesN;
    const int64_t t0 = (int64_t)tt * T::TL;
    const int n0 = tn * T::TR;
    const TileSrcAll<T> src(a.W, a.K, a.N, a.X, a.K, a.rows, n0, t0, w);
    GemmAcc<T> acc;
    acc.zero();
    const int nk = a.K / GEMM_BK;
    __syncthreads();
    if (a.trace && threadIdx.x == 0 && tcount == 2) a.trace[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
    gemm_stage<T::TR, T::WAVES, 0>(src.R, 0, sR + rs * T::R_BYTES, w.wave);
    gemm_stage<T::TL, T::WAVES, 0>(src.L, 0, sL + ls * T::L_BYTES, w.wave);
    gemm_stage<T::TR, T::WAVES, 0>(src.R, 1, sR + ((rs + 1) % 3) * T::R_BYTES, w.wave);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R_DPW) : "memory");
      else lds_dma_wait_all();
      lds_barrier();
      const char* tR = sR + rs * T::R_BYTES + offR;
      const char* tL = sL + ls * T::L_BYTES + offL;
      bf16x8 fa[2][T::MT], fb[2][T::NT];
#define load_frags(s_, set_)                                                                          \
  do {                                                                                              \
    const int ch_ = ((2 * (s_) + w.hi) ^ sw) * 16;                                                  \
    _Pragma("unroll") for (int j = 0; j < T::NT; ++j) fb[set_][j] = *(const bf16x8*)(tL + j * 32 * 128 + ch_); \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i) fa[set_][i] = *(const bf16x8*)(tR + i * 32 * 128 + ch_); \
  } while (0)
      load_frags(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const bool issue_l = kt + 1 < nk, issue_r = kt + 2 < nk;
      char* l_dst = sL + (ls ^ 1) * T::L_BYTES;
      const int rnext = rs == 0 ? 2 : rs - 1;
      char* r_dst = sR + rnext * T::R_BYTES;
      if (VARIANT == 0 && issue_l) gemm_stage<T::TL, T::WAVES, 0>(src.L, kt + 1, l_dst, w.wave);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        int dma = 0;
#pragma unroll
        for (int i = 0; i < T::MT; ++i)
#pragma unroll
          for (int j = 0; j < T::NT; ++j) {
            acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc.c[i][j], 0, 0, 0);
            if (VARIANT == 1 && (j & 1) == 1) {   // one DMA instruction per MFMA pair
              if (s == 0 && issue_l && dma < L_DPW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src.L.rsrc, (lptr_t)(l_dst + (dma * T::WAVES + w.wave) * 8 * 128), 16,
                                                         src.L.voff, dma * src.L.round_pitch + (kt + 1) * (GEMM_BK * 2), 0, 0);
              if (s >= 2 && issue_r && dma + (s - 2) * (R_DPW / 2) < R_DPW && dma < R_DPW / 2) {
                const int d2 = dma + (s - 2) * (R_DPW / 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src.R.rsrc, (lptr_t)(r_dst + (d2 * T::WAVES + w.wave) * 8 * 128), 16,
                                                         src.R.voff, d2 * src.R.round_pitch + (kt + 2) * (GEMM_BK * 2), 0, 0);
              }
              ++dma;
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (VARIANT == 0 && issue_r) gemm_stage<T::TR, T::WAVES, 0>(src.R, kt + 2, r_dst, w.wave);
      rs = rs == 2 ? 0 : rs + 1;
      ls ^= 1;
    }
    if (a.trace && threadIdx.x == 0 && tcount == 2) a.trace[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
      for (int j = 0; j < T::NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc.c[i][j][r];
  }
  a.Y[(size_t)blockIdx.x * T::THREADS + threadIdx.x] = sink;
}

template <int W8, int VARIANT> struct Kern;
#define DEF_KERN(W8_, V_, THREADS_)                                                      \
  __global__ void __launch_bounds__(THREADS_) k_loop_##W8_##_##V_(const Args a) { loop_body<W8_, V_>(a); } \
  template <> struct Kern<W8_, V_> { static constexpr auto fn = k_loop_##W8_##_##V_; };
DEF_KERN(1, 0, 512)
DEF_KERN(1, 1, 512)
DEF_KERN(0, 0, 256)
DEF_KERN(0, 1, 256)

template <int W8, int VARIANT>
static float run(const Args& a, int iters, const char* name) {
  using T = typename Pick<W8>::type;
  constexpr int SMEM = 3 * T::R_BYTES + 2 * T::L_BYTES;
  hipFuncSetAttribute((const void*)Kern<W8, VARIANT>::fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)Kern<W8, VARIANT>::fn);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((Kern<W8, VARIANT>::fn), dim3(256), dim3(T::THREADS), SMEM, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((Kern<W8, VARIANT>::fn), dim3(256), dim3(T::THREADS), SMEM, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  std::vector<unsigned long long> tr(1024);
  hipMemcpy(tr.data(), a.trace, 8192, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (int b = 0; b < 256; ++b) cyc += (double)(tr[b * 4 + 1] - tr[b * 4 + 0]);
  printf("%-44s %.3f ms  %.0f TF  regs %d  spill(scratch B) %zu  main loop of tile 2: %.0f memtime ticks\n", name, ms,
         2.0 * a.rows * a.N * a.K / ms / 1e9, fa.numRegs, (size_t)fa.localSizeBytes, cyc / 256);
  return ms;
}

int main() {
  const int64_t rows = 65536;
  const int N = 3072, K = 768;
  bf16_t *W, *X;
  float* Y;
  unsigned long long* trace;
  hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&X, (size_t)rows * K * 2); hipMalloc(&Y, 256 * 512 * 4); hipMalloc(&trace, 8192);
  std::vector<bf16_t> h((size_t)rows * K);
  srand(1);
  for (auto& v : h) v = (bf16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(X, h.data(), (size_t)rows * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  Args a{W, X, Y, rows, N, K, N / 256, (int)(rows / 256), trace};
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 0>(a, 10, "8 waves (2/SIMD), 128x64 per wave, blocks");
    run<0, 0>(a, 10, "4 waves (1/SIMD), 128x128 per wave, blocks");
    run<0, 1>(a, 10, "4 waves (1/SIMD), DMA between MFMA pairs");
    run<1, 1>(a, 10, "8 waves (2/SIMD), DMA between MFMA pairs");
  }
  return 0;
}
