// Prototype (NOT product): main loop of a 256 x 256 tile with FOUR waves per SIMD (16 waves, 64 x 64 per wave, 64
// accumulator registers of a 128-register budget) against the product geometry (8 waves, 128 x 64 per wave), same R3
// K step (3 R slots + 2 L slots), FFN1 shape, no epilogue.  Question (round 5): two waves per SIMD leave the matrix pipe
// idle ~30 % of a K step (barrier -> first fragments, the younger wave's tail); do four waves per SIMD -- more
// independent instruction streams to cover each other's LDS round trips and barrier bubbles -- close that gap?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I convdr_amd/csrc tools/proto/w16_proto.hip -o tools/proto/bin/w16_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_nt.hpp"
using namespace convdr;

struct Args { const bf16_t* W; const bf16_t* X; float* Y; int64_t rows; int N, K; int tilesN, tilesT; unsigned long long* trace; };

using T8 = TileCfg<2, 4, 4, 2>;
using T16 = TileCfg<4, 4, 2, 2>;
using T16b = TileCfg<2, 8, 4, 1>;   // 16 waves of 128 x 32: the wave keeps 4 R fragments, 1 L fragment (5 reads per 4 MFMAs)

// ROLES 0: every wave issues its share of both chunks (L under the first fragment reads, R after the MFMAs)
// ROLES 1: the first half of the waves issues the whole R chunk after its MFMAs, the second half the whole L chunk at the top
// FRAG3: fragments prefetched one sub-step ahead (two register sets) as in the product; 0 = one set, read just in time
template <class T, int ROLES, int MODE = 0>
__device__ __forceinline__ void loop_body(const Args& a) {
  constexpr bool NO_DMA = MODE == 1 || MODE == 3 || MODE == 4, NO_FRAG = MODE == 2 || MODE == 3 || MODE == 4, NO_BAR = MODE == 4;
  constexpr bool ILV = MODE >= 5;   // fragment reads of sub-step s + 1 threaded between the MFMAs of sub-step s (5: 1 per MFMA, 6: 2 per 2 MFMAs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sR = smem;
  char* sL = smem + 3 * T::R_BYTES;
  const WavePos<T> w;
  const uint32_t ntiles = (uint32_t)a.tilesN * a.tilesT;
  const int sw = (w.lane >> 1) & 7;
  const int offR = (w.wr * T::MT * 32 + w.li) * 128;
  const int offL = (w.wl * T::NT * 32 + w.li) * 128;
  constexpr int RW = ROLES ? T::WAVES / 2 : T::WAVES, LW = RW, LFIRST = ROLES ? T::WAVES / 2 : 0;
  constexpr int R_DPW = T::TR / (8 * RW);
  const bool r_wave = !ROLES || w.wave < RW;
  int rs = 0, ls = 0;
  float sink = 0.f;
  int tcount = 0;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++tcount) {
    const int tt = tile / a.tilesN, tn = tile - tt * a.tilesN;
    const int64_t t0 = (int64_t)tt * T::TL;
    const int n0 = tn * T::TR;
    const StageSrc srcR = gemm_stage_src<RW, 0>(a.W, a.K, n0, a.N, w.wave, w.lane);
    const StageSrc srcL = gemm_stage_src<LW, LFIRST>(a.X, a.K, t0, a.rows, w.wave, w.lane);
    GemmAcc<T> acc;
    acc.zero();
    const int nk = a.K / GEMM_BK;
    __syncthreads();
    if (a.trace && threadIdx.x == 0 && tcount == 2) a.trace[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
    if (r_wave) gemm_stage<T::TR, RW, 0>(srcR, 0, sR + rs * T::R_BYTES, w.wave);
    gemm_stage<T::TL, LW, LFIRST>(srcL, 0, sL + ls * T::L_BYTES, w.wave);
    if (r_wave) gemm_stage<T::TR, RW, 0>(srcR, 1, sR + ((rs + 1) % 3) * T::R_BYTES, w.wave);
    for (int kt = 0; kt < nk; ++kt) {
      if (NO_DMA) { if (kt == 0) lds_dma_wait_all(); }
      else if (kt + 1 < nk && r_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R_DPW) : "memory");
      else lds_dma_wait_all();
      if (!NO_BAR || kt == 0) lds_barrier();
      const char* tR = sR + rs * T::R_BYTES + offR;
      const char* tL = sL + ls * T::L_BYTES + offL;
      bf16x8 fa[2][T::MT], fb[2][T::NT];
#define load_frags(s_, set_)                                                                          \
  do {                                                                                              \
    const int ch_ = ((2 * (s_) + w.hi) ^ sw) * 16;                                                  \
    _Pragma("unroll") for (int j = 0; j < T::NT; ++j) fb[set_][j] = *(const bf16x8*)(tL + j * 32 * 128 + ch_); \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i) fa[set_][i] = *(const bf16x8*)(tR + i * 32 * 128 + ch_); \
  } while (0)
      if (!NO_FRAG || kt == 0) load_frags(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const bool issue_l = kt + 1 < nk, issue_r = kt + 2 < nk;
      char* l_dst = sL + (ls ^ 1) * T::L_BYTES;
      const int rnext = rs == 0 ? 2 : rs - 1;
      char* r_dst = sR + rnext * T::R_BYTES;
      if (issue_l && !NO_DMA && MODE != 7 && MODE != 8) gemm_stage<T::TL, LW, LFIRST>(srcL, kt + 1, l_dst, w.wave);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (MODE == 7 && s == 1 && issue_l) { gemm_stage<T::TL, LW, LFIRST>(srcL, kt + 1, l_dst, w.wave); __builtin_amdgcn_sched_barrier(0); }
        if (MODE == 8 && s == 1 && issue_l) gemm_stage<T::TL, LW, LFIRST>(srcL, kt + 1, l_dst, w.wave);
        if (s + 1 < 4 && (!NO_FRAG || kt == 0)) load_frags(s + 1, (s + 1) & 1);
        if (!ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < T::MT; ++i)
#pragma unroll
          for (int j = 0; j < T::NT; ++j)
            acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc.c[i][j], 0, 0, 0);
        if (ILV && s + 1 < 4) {
          if (MODE == 8 && s == 1) {
#pragma unroll
            for (int r = 0; r < T::MT + T::NT; ++r) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one VMEM read (LDS-DMA piece)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          } else if (MODE == 5 || MODE == 7 || MODE == 8) {
#pragma unroll
            for (int r = 0; r < T::MT + T::NT; ++r) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
            }
          } else {
#pragma unroll
            for (int r = 0; r < (T::MT + T::NT) / 2; ++r) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x008, T::MT * T::NT, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (issue_r && r_wave && !NO_DMA) gemm_stage<T::TR, RW, 0>(srcR, kt + 2, r_dst, w.wave);
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

__global__ void __launch_bounds__(512) k8_0(const Args a) { loop_body<T8, 0>(a); }
__global__ void __launch_bounds__(512) k8_1(const Args a) { loop_body<T8, 1>(a); }
__global__ void __launch_bounds__(512) k8_nodma(const Args a) { loop_body<T8, 1, 1>(a); }
__global__ void __launch_bounds__(512) k8_nofrag(const Args a) { loop_body<T8, 1, 2>(a); }
__global__ void __launch_bounds__(512) k8_neither(const Args a) { loop_body<T8, 1, 3>(a); }
__global__ void __launch_bounds__(512) k8_bare(const Args a) { loop_body<T8, 1, 4>(a); }
__global__ void __launch_bounds__(512) k8_ilv1(const Args a) { loop_body<T8, 1, 5>(a); }
__global__ void __launch_bounds__(512) k8_ilv2(const Args a) { loop_body<T8, 1, 6>(a); }
__global__ void __launch_bounds__(512) k8_ilv_l1(const Args a) { loop_body<T8, 1, 7>(a); }
__global__ void __launch_bounds__(512) k8_ilv_lt(const Args a) { loop_body<T8, 1, 8>(a); }
__global__ void __launch_bounds__(1024) k16_ilv(const Args a) { loop_body<T16, 1, 5>(a); }
__global__ void __launch_bounds__(1024) k16_0(const Args a) { loop_body<T16, 0>(a); }
__global__ void __launch_bounds__(1024) k16_1(const Args a) { loop_body<T16, 1>(a); }
__global__ void __launch_bounds__(1024) k16b_0(const Args a) { loop_body<T16b, 0>(a); }
__global__ void __launch_bounds__(1024) k16b_1(const Args a) { loop_body<T16b, 1>(a); }

template <class T, class F>
static float run(F fn, const Args& a, int iters, const char* name) {
  constexpr int SMEM = 3 * T::R_BYTES + 2 * T::L_BYTES;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
    printf("%-52s attribute failed\n", name);
    return 0.f;
  }
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)fn);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(T::THREADS), SMEM, 0, a);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
    printf("%-52s launch failed (regs %d)\n", name, fa.numRegs);
    return 0.f;
  }
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(T::THREADS), SMEM, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  std::vector<unsigned long long> tr(1024);
  hipMemcpy(tr.data(), a.trace, 8192, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (int b = 0; b < 256; ++b) cyc += (double)(tr[b * 4 + 1] - tr[b * 4 + 0]);
  printf("%-52s %.3f ms  %.0f TF  regs %d  scratch %zu B  main loop of tile 2: %.0f ticks (%.0f per K step)\n", name, ms,
         2.0 * a.rows * a.N * a.K / ms / 1e9, fa.numRegs, (size_t)fa.localSizeBytes, cyc / 256, cyc / 256 / (a.K / 64));
  return ms;
}

int main(int argc, char** argv) {
  const int64_t rows = 65536;
  const int N = 3072;
  const int K = argc > 1 ? atoi(argv[1]) : 768;
  bf16_t *W, *X;
  float* Y;
  unsigned long long* trace;
  hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&X, (size_t)rows * K * 2); hipMalloc(&Y, 256 * 1024 * 4); hipMalloc(&trace, 8192);
  std::vector<bf16_t> h((size_t)rows * K);
  srand(1);
  for (auto& v : h) v = (bf16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(X, h.data(), (size_t)rows * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  Args a{W, X, Y, rows, N, K, N / 256, (int)(rows / 256), trace};
  for (int rep = 0; rep < 3; ++rep) {
    run<T8>(k8_1, a, 10, "8 waves (2/SIMD) 128x64 per wave, roles (product)");
    run<T8>(k8_0, a, 10, "8 waves (2/SIMD) 128x64 per wave, shared issue");
    run<T8>(k8_ilv1, a, 10, "8 waves roles, fragment reads 1 per MFMA (interleaved)");
    run<T8>(k8_ilv2, a, 10, "8 waves roles, fragment reads 2 per 2 MFMAs");
    run<T8>(k8_ilv_l1, a, 10, "8 waves roles, ILV + L chunk issued after sub-step 0");
    run<T8>(k8_ilv_lt, a, 10, "8 waves roles, ILV + L chunk threaded through sub-step 1");
    run<T16>(k16_ilv, a, 10, "16 waves (4/SIMD) 64x64, roles, ILV");
    run<T8>(k8_nodma, a, 10, "8 waves roles, NO DMA in the loop (timing only)");
    run<T8>(k8_nofrag, a, 10, "8 waves roles, NO fragment reads in the loop");
    run<T8>(k8_neither, a, 10, "8 waves roles, neither (MFMA + barrier)");
    run<T8>(k8_bare, a, 10, "8 waves roles, neither, no barrier (bare MFMA)");
    run<T16>(k16_0, a, 10, "16 waves (4/SIMD) 64x64 per wave, shared issue");
    run<T16>(k16_1, a, 10, "16 waves (4/SIMD) 64x64 per wave, roles");
    run<T16b>(k16b_0, a, 10, "16 waves (4/SIMD) 128x32 per wave, shared issue");
    run<T16b>(k16b_1, a, 10, "16 waves (4/SIMD) 128x32 per wave, roles");
  }
  return 0;
}
