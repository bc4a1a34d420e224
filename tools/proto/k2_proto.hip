// Prototype (NOT product): can the epilogue of a short-K projection (FFN1: K = 768, 28 % of a 256 x 256 tile's time)
// be hidden by giving every SIMD two INDEPENDENT waves instead of two waves of one workgroup?
//   product (k_gemm<Tile256>): 1 workgroup of 8 waves per CU, 256 x 256 tile, 128 x 64 per wave, BK = 64, 160 KB LDS;
//                              both waves of a SIMD reach the epilogue together -> the matrix pipe idles through it
//   here:                      2 workgroups of 4 waves per CU, 256 x 128 tile each, the same 128 x 64 per wave,
//                              BK = 32 stages of 24 KB in a ring of three (72 KB per workgroup), the K loop runs
//                              continuously across tiles (the ring never drains), the finished tile parks in the slot its
//                              last step read.  The two workgroups of a CU are started half a tile apart, so one is in
//                              its main loop while the other converts / stores.
// Costs: 1.5 x the L2 -> LDS bytes per FLOP of the 256 x 256 tile, a barrier per 512 MFMA cycles (4 waves), 6 DMA
// instructions per wave per step.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I convdr_amd/csrc tools/proto/k2_proto.hip -o gpurun_out/k2_proto
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_nt.hpp"
using namespace convdr;

struct Args {
  const bf16_t* W;   // [N, K]  -> R operand (features on accumulator registers)
  const bf16_t* X;   // [rows, K] -> L operand (tokens on lanes)
  bf16_t* C;         // [rows, N] = gelu(X W^T)
  int64_t rows;
  int N, K;
  int tilesN, tilesT;
  int stagger;       // 1: workgroups of the second dispatch round start half a tile late
  int epi;           // 0: no epilogue (sink), 1: gelu + park + store
};

constexpr int TR = 256, TL = 128, WAVES = 4, THREADS = 256;
constexpr int BKB = 64;                       // bytes of K per stage row (BK = 32 bf16)
constexpr int R_BYTES = TR * BKB, L_BYTES = TL * BKB, STAGE = R_BYTES + L_BYTES;   // 16 + 8 = 24 KB
constexpr int SMEM = 3 * STAGE;               // 72 KB
constexpr int R_DPW = TR / 16 / WAVES, L_DPW = TL / 16 / WAVES;   // DMA instructions per wave per stage: 4 + 2
constexpr int DPW = R_DPW + L_DPW;

struct Src {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff, pitch16;   // this lane's offset inside a 16-row piece; bytes between pieces
};
__device__ __forceinline__ Src make_src(const bf16_t* G, int64_t ld, int64_t row0, int64_t nrows, int lane) {
  Src s;
  int64_t bytes = (nrows - row0) * ld * 2;
  bytes = bytes < 0 ? 0 : (bytes > 0xffffffffll ? 0xffffffffll : bytes);
  const uint64_t base = (uint64_t)(G + row0 * ld);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
  const int row = lane >> 2;                                  // 16 rows of 64 bytes per instruction
  const int gch = (lane & 3) ^ ((row >> 2) & 3);              // LDS chunk (lane & 3) of this row holds source chunk gch
  s.voff = (uint32_t)(row * ld * 2) + gch * 16;
  s.pitch16 = __builtin_amdgcn_readfirstlane((uint32_t)(16 * ld * 2));
  return s;
}
template <int PIECES_PER_WAVE>
__device__ __forceinline__ void issue(const Src& s, int kt, char* lds_tile, int wave) {
#pragma unroll
  for (int j = 0; j < PIECES_PER_WAVE; ++j) {
    const int piece = j * WAVES + wave;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lptr_t)(lds_tile + piece * 1024), 16, s.voff, piece * s.pitch16 + kt * BKB, 0, 0);
  }
}

__device__ __forceinline__ void gelu2(float& x0, float& x1) {   // csrc/encoder_kernels.hpp:gelu_tail2
  f32x2_t t, r;
  float a, b;
  const float nine = 9.f;
  asm("v_min_f32 %0, |%1|, %2" : "=v"(a) : "v"(x0), "s"(nine));
  asm("v_min_f32 %0, |%1|, %2" : "=v"(b) : "v"(x1), "s"(nine));
  t.x = a; t.y = b;
  asm("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(x0));
  asm("v_max_f32 %0, 0, %1" : "=v"(b) : "v"(x1));
  r.x = a; r.y = b;
  const f32x2_t c3 = {0.0041585f, 0.0041585f}, c2 = {-0.04571999f, -0.04571999f}, c1 = {-0.46495319f, -0.46495319f},
                c0 = {-1.14955714f, -1.14955714f}, m1 = {-1.f, -1.f};
  f32x2_t p = __builtin_elementwise_fma(t, c3, c2);
  p = __builtin_elementwise_fma(p, t, c1);
  p = __builtin_elementwise_fma(p, t, c0);
  p = __builtin_elementwise_fma(p, t, m1);
  f32x2_t q;
  q.x = __builtin_amdgcn_exp2f(p.x);
  q.y = __builtin_amdgcn_exp2f(p.y);
  r = __builtin_elementwise_fma(-t, q, r);
  x0 = r.x; x1 = r.y;
}

template <int VAR>
__global__ void __launch_bounds__(THREADS, 2) k2(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wl = wave & 1, hi = lane >> 5, li = lane & 31;
  const uint32_t ntiles = (uint32_t)a.tilesN * a.tilesT;
  // XCD x owns a contiguous chunk of the tile order (feature tile fastest: the token tile stays in that XCD's L2)
  const uint32_t xcd = blockIdx.x & 7u, q8 = ntiles >> 3, r8 = ntiles & 7u;
  const uint32_t chunk_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const uint32_t chunk_len = q8 + (xcd < r8 ? 1u : 0u);
  const uint32_t stride = (gridDim.x + 7u) >> 3;
  uint32_t idx = blockIdx.x >> 3;
  if (idx >= chunk_len) return;
  if (a.stagger && blockIdx.x >= gridDim.x / 2) {
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);   // ~2 x 8 k cycles: half a tile
  }
  const int nk = a.K / 32;
  const int sw = (li >> 2) & 3;
  const int offR = (wr * 128 + li) * BKB, offL = R_BYTES + (wl * 64 + li) * BKB;
  auto coords = [&](uint32_t i, int64_t& t0, int& n0) {
    const uint32_t logical = chunk_base + i;
    const int tt = logical / a.tilesN, tn = logical - tt * a.tilesN;
    t0 = (int64_t)tt * TL; n0 = tn * TR;
  };
  int64_t t0; int n0;
  coords(idx, t0, n0);
  Src sR = make_src(a.W, a.K, n0, a.N, lane), sL = make_src(a.X, a.K, t0, a.rows, lane);
  // ring position of the NEXT stage to issue and of the stage to consume
  int slot_c = 0;
  issue<R_DPW>(sR, 0, smem + 0 * STAGE, wave); issue<L_DPW>(sL, 0, smem + 0 * STAGE + R_BYTES, wave);
  issue<R_DPW>(sR, 1, smem + 1 * STAGE, wave); issue<L_DPW>(sL, 1, smem + 1 * STAGE + R_BYTES, wave);
  float sink = 0.f;
  int landed = 0;   // leading stages of the coming tile that are known to have landed (waited for inside the epilogue)
  for (;;) {
    GemmAcc<TileCfg<2, 2, 4, 2>> acc;
    acc.zero();
    const uint32_t next = idx + stride;
    const bool has_next = next < chunk_len;
    int64_t t0n = 0; int n0n = 0;
    Src sRn = sR, sLn = sL;
    if (has_next) {
      coords(next, t0n, n0n);
      sRn = make_src(a.W, a.K, n0n, a.N, lane);
      sLn = make_src(a.X, a.K, t0n, a.rows, lane);
    }
    for (int kt = 0; kt < nk; ++kt) {
      // stage kt has landed (this wave's share; the barrier extends that to every wave), stage kt + 1 may be in flight
      // (after an epilogue the queue also holds its stores: the first two stages were waited for before those were
      //  issued, and by step 2 a counted wait no longer sits behind a fresh store)
      const bool more = kt + 1 < nk || has_next;
      if (kt >= landed) {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
        else lds_dma_wait_all();
      }
      lds_barrier();
      // stage kt + 2 -> the slot step kt - 1 read (every wave is past that step's reads: it has passed this barrier)
      // VAR 0: issued first; 1: after the MFMAs; 2: under the fragment reads' LDS round trip; 3: as 1 with s_setprio
      const int slot_i = slot_c == 0 ? 2 : slot_c - 1;
      auto dma = [&]() {
        if (kt + 2 < nk) {
          issue<R_DPW>(sR, kt + 2, smem + slot_i * STAGE, wave); issue<L_DPW>(sL, kt + 2, smem + slot_i * STAGE + R_BYTES, wave);
        } else if (has_next) {   // the first two stages of the next tile: the ring never drains
          issue<R_DPW>(sRn, kt + 2 - nk, smem + slot_i * STAGE, wave); issue<L_DPW>(sLn, kt + 2 - nk, smem + slot_i * STAGE + R_BYTES, wave);
        }
      };
      if (VAR == 0) dma();
      const char* tR = smem + slot_c * STAGE + offR;
      const char* tL = smem + slot_c * STAGE + offL;
      bf16x8 fa[2][4], fb[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ch = ((2 * s + hi) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[s][j] = *(const bf16x8*)(tL + j * 32 * BKB + ch);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[s][i] = *(const bf16x8*)(tR + i * 32 * BKB + ch);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (VAR == 2) { dma(); __builtin_amdgcn_sched_barrier(0); }
      if (VAR == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc.c[i][j], 0, 0, 0);
      }
      if (VAR == 3) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (VAR == 1 || VAR == 3) { dma(); __builtin_amdgcn_sched_barrier(0); }
      slot_c = slot_c == 2 ? 0 : slot_c + 1;
    }
    // ---- epilogue: the slot the last step read is dead once every wave has left the loop ----
    if (a.epi) {
      const int dead = slot_c == 0 ? 2 : slot_c - 1;
      lds_barrier();
      const uint32_t wb = lds_off(smem + dead * STAGE) + wave * 4096;   // 4 KB per wave: 32 tokens x 64 features (128 B rows)
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane_e = tid & 63, li_e = lane_e & 31, hi_e = lane_e >> 5;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // features [64 half, 64 half + 64) of the wave's 128
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) {
            const int mt = 2 * half + m2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v0 = acc.c[mt][nt][4 * g], v1 = acc.c[mt][nt][4 * g + 1], v2 = acc.c[mt][nt][4 * g + 2], v3 = acc.c[mt][nt][4 * g + 3];
              gelu2(v0, v1); gelu2(v2, v3);
              const uint32_t p0 = pack_bf16x2(v0, v1), p1 = pack_bf16x2(v2, v3);
              // row = token li, 8 chunks of 16 B per 128-byte row: chunk (m2 * 4 + g) ^ (row & 7), 8 bytes at hi * 8
              lds_write_b64_hidden(wb + li_e * 128 + (((m2 * 4 + g) ^ (li_e & 7)) << 4) + hi_e * 8, (u32x2_t){p0, p1});
            }
          }
          // the next tile's first two stages must have landed BEFORE the first store enters the (in-order) queue: a
          // counted wait behind stores would wait for their acknowledgements
          if (nt == 0 && half == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // 32 rows x 128 B: 8 rows per instruction
          u32x4_t v[4];
          const int r0 = lane_e >> 3, c = lane_e & 7;
          auto at = [&](int i) { const int row = i * 8 + r0; return wb + row * 128 + ((c ^ (row & 7)) << 4); };
          lds_read4_b128_hidden(at(0), at(1), at(2), at(3), v[0], v[1], v[2], v[3]);
          bf16_t* dst = a.C + (t0 + wl * 64 + nt * 32) * a.N + n0 + wr * 128 + half * 64;
#pragma unroll
          for (int i = 0; i < 4; ++i) *(u32x4_t*)(dst + (int64_t)(i * 8 + r0) * a.N + c * 8) = v[i];
        }
      // the next loop's first barrier orders the parked slot against its re-use as a stage (issued after that barrier)
      landed = 2;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sink += acc.c[i][j][r];
    }
    if (!has_next) break;
    idx = next; t0 = t0n; n0 = n0n; sR = sRn; sL = sLn;
  }
  if (!a.epi) a.C[(size_t)blockIdx.x * THREADS + threadIdx.x] = (bf16_t)(int)sink;
}

static float bf(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 262144;
  const int N = 3072, K = 768;
  bf16_t *W, *X, *C;
  hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&X, (size_t)rows * K * 2); hipMalloc(&C, (size_t)rows * N * 2);
  std::vector<bf16_t> hx((size_t)rows * K), hw((size_t)N * K);
  srand(1);
  auto rnd = [] { return (bf16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15) - ((rand() & 3) << 7)); };   // +-[0.25, 1.0)
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = (bf16_t)(rnd() - 0x0280);   // ~ /32
  hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, int var) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)kern);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, THREADS, SMEM);
    printf("variant %d: regs %d, scratch %zu B, LDS %d B, workgroups per CU %d\n", var, fa.numRegs, (size_t)fa.localSizeBytes, SMEM, occ);
    for (int epi = 0; epi < 2; ++epi)
      for (int stagger = 0; stagger < 2; ++stagger) {
        const int grid = 512;
        Args a{W, X, C, rows, N, K, N / TR, (int)(rows / TL), stagger, epi};
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), SMEM, 0, a);
        hipEventRecord(e0);
        const int iters = 10;
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), SMEM, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= iters;
        printf("  variant %d epilogue %d  stagger %d  grid %3d: %.3f ms  %.0f TF\n", var, epi, stagger, grid, ms, 2.0 * rows * N * K / ms / 1e9);
      }
  };
  for (int rep = 0; rep < 2; ++rep) {
    run(k2<0>, 0); run(k2<1>, 1); run(k2<2>, 2); run(k2<3>, 3);
  }
  // spot check of the stored tile values (last configuration ran with the epilogue)
  std::vector<bf16_t> hc((size_t)4096 * N);
  hipMemcpy(hc.data(), C + (size_t)(rows - 4096) * N, hc.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 2000; ++s) {
    const int64_t t = rows - 4096 + rand() % 4096;
    const int n = rand() % N;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (double)bf(hx[t * K + k]) * bf(hw[(size_t)n * K + k]);
    const double ref = acc * 0.5 * (1.0 + erf(acc / sqrt(2.0)));
    const double got = bf(hc[(size_t)(t - (rows - 4096)) * N + n]);
    worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 0.05));
  }
  printf("spot check vs fp64 gelu(x.w): worst relative error %.4f (bf16 output: expect < 0.01)\n", worst);
  return 0;
}
