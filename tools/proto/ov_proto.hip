// Prototype (NOT product): does a 256 x 128 tile with two accumulator sets hide the GELU / convert / park epilogue of
// tile i under the MFMAs of tile i + 1?  Synthetic FFN1-shaped GEMM, R3 K step, three epilogue schedules:
//   mode 0  serial epilogue after each tile's main loop
//   mode 1  epilogue of the previous tile issued as slices inside the next tile's K loop (compiler's schedule)
//   mode 2  the same with sched_group_barrier patterns (1 MFMA : N VALU)
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I convdr_amd/csrc tools/proto/ov_proto.hip -o gpurun_out/ov_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_nt.hpp"
using namespace convdr;

using TP = TileCfg<4, 2, 2, 2>;   // TR = 256 features, TL = 128 tokens, wave tile 64 x 64
constexpr int PARK_BYTES = 64 * 256 * 2;                       // one pass: 64 tokens x 256 features bf16
constexpr int SMEM = 3 * TP::R_BYTES + 2 * TP::L_BYTES + PARK_BYTES;   // 96 + 32 + 32 = 160 KB

__device__ __forceinline__ float gelu_tail(float x) {
  const float t = fminf(fabsf(x), 9.f);
  float p = fmaf(t, 0.0041585f, -0.04571999f);
  p = fmaf(p, t, -0.46495319f);
  p = fmaf(p, t, -1.14955714f);
  const float q = __builtin_amdgcn_exp2f(fmaf(p, t, -1.f));
  return fmaf(-t, q, fmaxf(x, 0.f));
}

struct Args { const bf16_t* W; const bf16_t* X; bf16_t* Y; int64_t rows; int N, K; int tilesN, tilesT; };

// park address: row = token within pass (0..63), 512 B per row, 16-byte chunk swizzled by the row
__device__ __forceinline__ uint32_t park_addr(int row, int feat) {
  const int ch = feat >> 3;
  return row * 512 + ((ch ^ (row & 31)) << 4) + (feat & 7) * 2;
}

// convert one MFMA tile (mt, nt) of `acc` and park it (pass = nt)
template <class T>
__device__ __forceinline__ void convert_park(const f32x16& v, const WavePos<T>& w, int mt, char* park) {
  const int row = w.wl * 32 + w.li;   // token within the pass
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int f = w.r_base(mt, g);
    u32x2_t o;
    o.x = pack_bf16x2(gelu_tail(v[4 * g + 0]), gelu_tail(v[4 * g + 1]));
    o.y = pack_bf16x2(gelu_tail(v[4 * g + 2]), gelu_tail(v[4 * g + 3]));
    lds_write_b64_hidden(lds_off(park) + park_addr(row, f), o);
  }
}
// cooperative store of a parked pass: 64 rows x 512 B = 2048 chunks of 16 B, 4 per thread
template <class T>
__device__ __forceinline__ void store_pass(const char* park, bf16_t* Y, int64_t ldy, int64_t t0, int n0, int pass, int tid) {
  u32x4_t v[4];
  uint32_t ad[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 512 + tid, row = idx >> 5, c = idx & 31;
    ad[i] = lds_off(park) + row * 512 + ((c ^ (row & 31)) << 4);
  }
  lds_read4_b128_hidden(ad[0], ad[1], ad[2], ad[3], v[0], v[1], v[2], v[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 512 + tid, row = idx >> 5, c = idx & 31;
    const int tok = (row >> 5) * 64 + pass * 32 + (row & 31);   // wl * 64 + nt * 32 + li
    *(u32x4_t*)(Y + (t0 + tok) * ldy + n0 + c * 8) = v[i];
  }
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) k_proto(const Args a) {
  using T = TP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sR = smem;
  char* sL = smem + 3 * T::R_BYTES;
  char* park = smem + 3 * T::R_BYTES + 2 * T::L_BYTES;
  const WavePos<T> w;
  const int tid = threadIdx.x;
  const uint32_t ntiles = (uint32_t)a.tilesN * a.tilesT;
  const int sw = (w.lane >> 1) & 7;
  const int offR = (w.wr * T::MT * 32 + w.li) * 128;
  const int offL = (w.wl * T::NT * 32 + w.li) * 128;
  constexpr int R_DPW = T::TR / 64, L_DPW = T::TL / 64;
  GemmAcc<T> acc, prev;
  bool have_prev = false;
  int64_t pt0 = 0;
  int pn0 = 0;
  int rs = 0, ls = 0;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tt = tile / a.tilesN, tn = tile - tt * a.tilesN;
    const int64_t t0 = (int64_t)tt * T::TL;
    const int n0 = tn * T::TR;
    const TileSrcAll<T> src(a.W, a.K, a.N, a.X, a.K, a.rows, n0, t0, w);
    acc.zero();
    const int nk = a.K / GEMM_BK;
    __syncthreads();
    gemm_stage<T::TR, 8, 0>(src.R, 0, sR + rs * T::R_BYTES, w.wave);
    gemm_stage<T::TL, 8, 0>(src.L, 0, sL + ls * T::L_BYTES, w.wave);
    gemm_stage<T::TR, 8, 0>(src.R, 1, sR + ((rs + 1) % 3) * T::R_BYTES, w.wave);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R_DPW) : "memory");
      else lds_dma_wait_all();
      lds_barrier();
      const char* tR = sR + rs * T::R_BYTES + offR;
      const char* tL = sL + ls * T::L_BYTES + offL;
      bf16x8 fa[2][T::MT], fb[2][T::NT];
      auto load_frags = [&](int s, int set) {
        const int ch = ((2 * s + w.hi) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < T::NT; ++j) fb[set][j] = *(const bf16x8*)(tL + j * 32 * 128 + ch);
#pragma unroll
        for (int i = 0; i < T::MT; ++i) fa[set][i] = *(const bf16x8*)(tR + i * 32 * 128 + ch);
      };
      load_frags(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) gemm_stage<T::TL, 8, 0>(src.L, kt + 1, sL + (ls ^ 1) * T::L_BYTES, w.wave);
      // stores of a parked pass of the previous tile: after this step's barrier every wave's park writes are visible
      if (MODE >= 1 && have_prev && (kt == 3 || kt == 7)) store_pass<T>(park, a.Y, a.N, pt0, pn0, kt == 3 ? 0 : 1, tid);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
        if (MODE != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < T::MT; ++i)
#pragma unroll
          for (int j = 0; j < T::NT; ++j)
            acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc.c[i][j], 0, 0, 0);
        // one quarter of an MFMA tile's conversion (4 values... here: a whole register quad group g = s) per sub-step
        if (MODE >= 1 && have_prev) {
          const int item = kt == 1 ? 0 : kt == 2 ? 1 : kt == 5 ? 2 : kt == 6 ? 3 : -1;   // (mt, nt) = (item & 1, item >> 1)
          if (item >= 0) {
            auto conv = [&](const f32x16& v, int mt) {
              const int row = w.wl * 32 + w.li;
              const int f = w.r_base(mt, s);
              u32x2_t o;
              o.x = pack_bf16x2(gelu_tail(v[4 * s + 0]), gelu_tail(v[4 * s + 1]));
              o.y = pack_bf16x2(gelu_tail(v[4 * s + 2]), gelu_tail(v[4 * s + 3]));
              lds_write_b64_hidden(lds_off(park) + park_addr(row, f), o);
            };
            if (item == 0) conv(prev.c[0][0], 0);
            else if (item == 1) conv(prev.c[1][0], 1);
            else if (item == 2) conv(prev.c[0][1], 0);
            else conv(prev.c[1][1], 1);
            if (MODE == 2) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);  // 12 VALU
              }
            }
          }
        }
        if (MODE != 2) __builtin_amdgcn_sched_barrier(0);
      }
      const int rnext = rs == 0 ? 2 : rs - 1;
      if (kt + 2 < nk) gemm_stage<T::TR, 8, 0>(src.R, kt + 2, sR + rnext * T::R_BYTES, w.wave);
      rs = rs == 2 ? 0 : rs + 1;
      ls ^= 1;
    }
    if (MODE == 0) {   // serial epilogue
      for (int pass = 0; pass < 2; ++pass) {
        lds_barrier();
        convert_park<T>(acc.c[0][pass], w, 0, park);
        convert_park<T>(acc.c[1][pass], w, 1, park);
        lds_barrier();
        store_pass<T>(park, a.Y, a.N, t0, n0, pass, tid);
      }
    } else {
      prev = acc;
      have_prev = true;
      pt0 = t0;
      pn0 = n0;
    }
  }
  if (MODE >= 1 && have_prev) {   // flush the last tile
    for (int pass = 0; pass < 2; ++pass) {
      lds_barrier();
      convert_park<T>(prev.c[0][pass], w, 0, park);
      convert_park<T>(prev.c[1][pass], w, 1, park);
      lds_barrier();
      store_pass<T>(park, a.Y, a.N, pt0, pn0, pass, tid);
    }
  }
}

template <int MODE>
static float run(const Args& a, int iters) {
  hipFuncSetAttribute((const void*)k_proto<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_proto<MODE>, dim3(256), dim3(512), SMEM, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_proto<MODE>, dim3(256), dim3(512), SMEM, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int64_t rows = 262144;
  const int N = 3072, K = 768;
  bf16_t *W, *X, *Y;
  hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&X, (size_t)rows * K * 2); hipMalloc(&Y, (size_t)rows * N * 2);
  std::vector<bf16_t> h((size_t)rows * K);
  srand(1);
  for (auto& v : h) v = (bf16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // ~ +-0.01..0.03
  hipMemcpy(X, h.data(), (size_t)rows * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  Args a{W, X, Y, rows, N, K, N / 256, (int)(rows / 128)};
  std::vector<bf16_t> y0(4096), y1(4096), y2(4096);
  const double flop = 2.0 * rows * N * K;
  float t0 = run<0>(a, 10); hipMemcpy(y0.data(), Y + 12345 * 3072, 8192, hipMemcpyDeviceToHost);
  hipMemset(Y, 0, (size_t)rows * N * 2);
  float t1 = run<1>(a, 10); hipMemcpy(y1.data(), Y + 12345 * 3072, 8192, hipMemcpyDeviceToHost);
  hipMemset(Y, 0, (size_t)rows * N * 2);
  float t2 = run<2>(a, 10); hipMemcpy(y2.data(), Y + 12345 * 3072, 8192, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 4096; ++i) { bad1 += y0[i] != y1[i]; bad2 += y0[i] != y2[i]; }
  printf("256x128 tiles, FFN1 shape: serial epilogue %.3f ms (%.0f TF) | overlapped %.3f ms (%.0f TF), mismatches %d | "
         "overlapped + sched_group_barrier %.3f ms (%.0f TF), mismatches %d\n",
         t0, flop / t0 / 1e9, t1, flop / t1 / 1e9, bad1, t2, flop / t2 / 1e9, bad2);
  return 0;
}
