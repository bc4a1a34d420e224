// Row-complete GEMM with fused residual + LayerNorm epilogue (inference path, hidden size 768):
//     X_out[t, :] = LayerNorm( A[t, :] . W^T + bias + R[t, :] ) * gamma + beta        (bf16 out)
// replaces  k_gemm<EPI_RESID_F32> (fp32 pre-LN sums to HBM) + k_layernorm (read them back): one workgroup owns
// 128 tokens x ALL 768 output features, so the row statistics are available on chip and the 3 KB/token fp32
// round trip (27 % of the encoder's HBM traffic) disappears.
//   HF call sites: BertSelfOutput / BertOutput  (dense -> dropout -> LayerNorm(x + residual)), reached from
//   /root/reference/model/models.py:141-142.
// Geometry: 8 waves = 4 (features) x 2 (tokens); wave tile 192 features x 64 tokens = 6 x 2 MFMA 32x32x16 tiles
// (192 accumulator VGPRs).  Operands staged by LDS-DMA in 32-wide K slices, double buffered:
// W slice 768 x 64 B = 48 KB, A slice 128 x 64 B = 8 KB  -> 112 KB LDS, one workgroup per CU.
// 64-byte LDS rows: 16-byte chunk index XOR (row >> 2) & 3 keeps the 32-row ds_read_b128 fragment reads
// conflict-free (applied to the DMA source address, LDS-DMA writes lane-linear).
#pragma once
#include "gemm_nt.hpp"

namespace convdr {

using TileLN = TileCfg<4, 2, 6, 2>;   // TR = 768 features, TL = 128 tokens
constexpr int LN_SLICE = 32;
constexpr int LN_R_BYTES = TileLN::TR * 64, LN_L_BYTES = TileLN::TL * 64;
constexpr int LN_SMEM_BYTES = 3 * LN_R_BYTES + 2 * LN_L_BYTES;   // 160 KB: three weight slots + two activation slots

struct GemmLnArgs {
  const bf16_t* W;      // [768, K]
  const bf16_t* A;      // [rows, K]
  int64_t rows;
  int K;
  const float* bias;    // [768]
  const bf16_t* R;      // residual [rows, 768]
  const float *gamma, *beta;
  float eps;
  bf16_t* X;            // out [rows, 768] (may alias R: a workgroup reads its residual rows before it writes them)
  unsigned long long* trace;   // experiment: [workgroup][16] s_memtime stamps of thread 0 (or null)
  const bf16_t* Wks;    // W in K-slice-major order [K / 32][768][32] (or null: stream the row-major W)
  int a_blocked;        // A is in the blocked layout [rows / 32][K / 8][32][8] the FFN1 epilogue EPI_GELU_BLK writes
  int dbg_skip_epi;     // timing experiment (TRACE library only): the kernel ends after its main loop (garbage results)
};
#ifdef CONVDR_ENABLE_TRACE   // make TRACE=1: phase stamps for tools/gemm_trace_ln.py
#define CONVDR_LN_TRACE(ph) \
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + (ph)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_LN_TRACE(ph)
#endif

// 32-wide K slices: 64-byte LDS rows, 16 rows per wave per round, rounds 128 rows apart (so the swizzle term
// (row >> 2) & 3 does not depend on the round).  Same buffer-descriptor addressing as gemm_nt.hpp's gemm_stage.
// Who issues the LDS-DMA, as R3Issue in gemm_nt.hpp: the older wave of each SIMD (waves 0-3), which the matrix pipe serves
// first and which then idles in the barrier, issues the WHOLE weight slice of step t + 2 after its MFMAs; the younger
// (waves 4-7) issues the activation slice of step t + 1 at the top of the step (FFN2 12.95 -> 12.44 ms per 12 layers).
constexpr int LN_DMA_WAVES = 8, LN_DMA_FIRST = 0;                 // cooperative loads outside the main loop
constexpr int LN_W_WAVES = 4, LN_W_FIRST = 0;                     // weight slices
constexpr int LN_A_WAVES = 4, LN_A_FIRST = 4;                     // activation slices
constexpr int LN_A_AUX = 2;   // cache policy of the activation-slice DMA (2 = nt: every activation row is read by ONE workgroup)

template <int WAVES = LN_DMA_WAVES, int FIRST = LN_DMA_FIRST>   // WAVES issuing waves, the first of which is wave FIRST
__device__ __forceinline__ StageSrc ln_stage_src(const bf16_t* __restrict__ G, int64_t ld, int64_t row0, int64_t nrows,
                                                  int wave, int lane) {
  StageSrc s;
  wave = wave >= FIRST ? wave - FIRST : 0;
  int64_t bytes = (nrows - row0) * ld * 2;
  bytes = bytes < 0 ? 0 : (bytes > 0xffffffffll ? 0xffffffffll : bytes);
  const uint64_t base = (uint64_t)(G + row0 * ld);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
  const int row = wave * 16 + (lane >> 2);
  const int gch = (lane & 3) ^ ((row >> 2) & 3);
  s.voff = (uint32_t)(row * ld * 2) + gch * 16;
  s.round_pitch = __builtin_amdgcn_readfirstlane((uint32_t)(16 * WAVES * ld * 2));
  return s;
}

template <int ROWS, int WAVES = LN_DMA_WAVES, int FIRST = LN_DMA_FIRST, int AUX = 0>
__device__ __forceinline__ void ln_stage32(const StageSrc& s, int ks, char* lds_tile, int wave,
                                           uint32_t slice_stride = LN_SLICE * 2) {
  if (wave < FIRST || wave >= FIRST + WAVES) return;   // wave-uniform
  wave -= FIRST;
  constexpr int ROUNDS = ROWS / (16 * WAVES);
  static_assert(ROUNDS * 16 * WAVES == ROWS, "rows must be a multiple of 16 x issuing waves");
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lptr_t)(lds_tile + (i * WAVES + wave) * 16 * 64), 16, s.voff,
                                             i * s.round_pitch + ks * slice_stride, 0, AUX);
}

__global__ void __launch_bounds__(512, 2) k_gemm_resid_ln(const GemmLnArgs a) {
  using T = TileLN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WavePos<T> w;
  const int64_t t0 = (int64_t)blockIdx.x * T::TL;
  char* sW = smem;                       // 3 weight slots
  char* sA = smem + 3 * LN_R_BYTES;      // 2 activation slots
  GemmAcc<T> acc;
  acc.zero();
  CONVDR_LN_TRACE(0)
  const int nk = a.K / LN_SLICE;
  const int sw = (w.li >> 2) & 3;
  const int offR = (w.wr * T::MT * 32 + w.li) * 64;
  const int offL = (w.wl * T::NT * 32 + w.li) * 64;

  // Weights come from the K-slice-major copy when there is one: a 32-wide slice of the row-major [768, K] matrix is
  // 768 half cache lines, every line is fetched twice (once per slice) and the 112 KB that leaves L2 per CU per step
  // made this kernel L2-bandwidth-bound (5.2 k cycles per 1,536-cycle step); in slice-major order a slice is 48 KB of
  // whole lines.
  StageSrc srcW = a.Wks ? ln_stage_src<LN_W_WAVES, LN_W_FIRST>(a.Wks, LN_SLICE, 0, (int64_t)T::TR * (a.K / LN_SLICE), w.wave, w.lane)
                        : ln_stage_src<LN_W_WAVES, LN_W_FIRST>(a.W, a.K, 0, T::TR, w.wave, w.lane);
  const uint32_t w_slice_stride = a.Wks ? T::TR * LN_SLICE * 2 : LN_SLICE * 2;
  StageSrc srcA = ln_stage_src<LN_A_WAVES, LN_A_FIRST>(a.A, a.K, t0, a.rows, w.wave, w.lane);
  uint32_t a_slice_stride = LN_SLICE * 2;
  if (a.a_blocked) {
    // blocked activations: a 16-token x 4-octet DMA instruction reads 4 runs of 256 contiguous bytes (whole lines; the
    // row-major form reads 16 half lines).  Window: from this tile's first 32-token block to the end of the last block.
    const int64_t rows32 = (a.rows + 31) & ~(int64_t)31;
    const int64_t blk_elems = (int64_t)(a.K >> 3) * 256;            // one 32-token block, all K
    int64_t bytes = (rows32 - t0) / 32 * blk_elems * 2;
    bytes = bytes < 0 ? 0 : (bytes > 0xffffffffll ? 0xffffffffll : bytes);
    const uint64_t base = (uint64_t)(a.A + (t0 >> 5) * blk_elems);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
    srcA.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    const int wv = w.wave >= LN_A_FIRST ? w.wave - LN_A_FIRST : 0;
    const int row = wv * 16 + (w.lane >> 2);                        // token inside a round of 16 x LN_A_WAVES tokens
    const int gch = (w.lane & 3) ^ ((row >> 2) & 3);                // source octet of the slice for LDS chunk (lane & 3)
    srcA.voff = (uint32_t)((row >> 5) * blk_elems * 2 + gch * 512 + (row & 31) * 16);
    // rounds are 16 * LN_A_WAVES tokens apart: a multiple of 32, so (row & 31) is round-independent and a round advances
    // whole 32-token blocks
    srcA.round_pitch = __builtin_amdgcn_readfirstlane((uint32_t)((16 * LN_A_WAVES / 32) * blk_elems * 2));
    a_slice_stride = 4 * 512;                                       // four octets per 32-wide slice
  }
  static_assert((16 * LN_A_WAVES) % 32 == 0 && TileLN::TL % (16 * LN_A_WAVES) == 0, "blocked A staging: rounds of whole 32-token blocks");
  // Three weight slots, two activation slots: the weight slice of step t + 2 is issued at step t.  With one slice in
  // flight (two stages) the stream was bound by bytes in flight / latency -- 56 KB per CU over a ~2 us loaded L2 round
  // trip = 28 GB/s per CU, 4.5 k cycles per 1,536-cycle step -- so the 48 KB of LDS this kernel left unused buy a
  // second weight slice in flight.  Issue order per step: A(t+1), then W(t+2); completion is in issue order, so
  // "W(t), A(t) landed" = all but the newest weight group (LN_W_DPW instructions per issuing wave) retired.
  constexpr int LN_W_DPW = T::TR / (16 * LN_W_WAVES);   // weight DMA instructions per issuing wave per slice
  const bool w_wave = w.wave >= LN_W_FIRST && w.wave < LN_W_FIRST + LN_W_WAVES;   // (wave-uniform) this wave has weight slices in flight
  ln_stage32<T::TR, LN_W_WAVES, LN_W_FIRST>(srcW, 0, sW, w.wave, w_slice_stride);
  ln_stage32<T::TL, LN_A_WAVES, LN_A_FIRST, LN_A_AUX>(srcA, 0, sA, w.wave, a_slice_stride);
  if (nk > 1) ln_stage32<T::TR, LN_W_WAVES, LN_W_FIRST>(srcW, 1, sW + LN_R_BYTES, w.wave, w_slice_stride);
#ifdef CONVDR_ENABLE_TRACE   // per-wave stamps of K step 8 (and the top of step 9): a.trace[2048 * 16 + wg * 64 + wave * 8 + i]
#define CONVDR_LN_STEP(i)                                                                                     \
  if (a.trace && kt == 8 + (i) / 5 && w.lane == 0 && blockIdx.x < 256)                                        \
    a.trace[2048 * 16 + blockIdx.x * 64 + w.wave * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_LN_STEP(i)
#endif
  int wslot = 0;   // kt % 3
  for (int kt = 0; kt < nk; ++kt) {
    CONVDR_LN_STEP(5)
    CONVDR_LN_STEP(0)
    if (kt + 1 < nk && w_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LN_W_DPW) : "memory");
    else lds_dma_wait_all();   // (a wave without weight slices has only this step's activation slice outstanding)
    CONVDR_LN_STEP(1)
    lds_barrier();   // NOT __syncthreads(): its fence would add vmcnt(0) and drain the slice that must stay in flight
    CONVDR_LN_STEP(2)
    if (kt + 1 < nk) ln_stage32<T::TL, LN_A_WAVES, LN_A_FIRST, LN_A_AUX>(srcA, kt + 1, sA + ((kt + 1) & 1) * LN_L_BYTES, w.wave, a_slice_stride);
    const bool issue_w = kt + 2 < nk;
    char* w_dst = sW + (wslot == 0 ? 2 : wslot - 1) * LN_R_BYTES;   // slot (kt + 2) % 3
    CONVDR_LN_STEP(3)
    const char* tR = sW + wslot * LN_R_BYTES + offR;
    const char* tL = sA + (kt & 1) * LN_L_BYTES + offL;
    wslot = wslot == 2 ? 0 : wslot + 1;
#ifdef CONVDR_LN_FRAG_JIT   // the round 1-4 form (A/B builds): hipcc read each weight fragment right before the MFMA pair that uses it
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ch = ((2 * s + w.hi) ^ sw) * 16;
      bf16x8 fa[T::MT], fb[T::NT];
#pragma unroll
      for (int j = 0; j < T::NT; ++j) fb[j] = *(const bf16x8*)(tL + j * 32 * 64 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i) fa[i] = *(const bf16x8*)(tR + i * 32 * 64 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i) {
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
          acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc.c[i][j], 0, 0, 0);
      }
    }
#else
    // Round 5: ROLLING fragment prefetch.  A slice is 12 (weight fragment, MFMA pair) units (2 sub-steps x 6 feature blocks);
    // with 192 accumulators there is no room for a second fragment set, and left to itself hipcc read each weight fragment
    // right before its MFMA pair and waited for it at once (`ds_read x 2; s_waitcnt lgkmcnt(1); mfma x 2`): one exposed LDS
    // round trip per pair, six per sub-step.  Here THREE weight-fragment registers rotate -- the fragment of unit u + 2 is
    // read before the MFMAs of unit u -- and the two activation fragments of sub-step 1 are read (into their own registers)
    // during sub-step 0: 12 more VGPRs in the loop, one exposed round trip per slice (right after the barrier, where nothing
    // can be prefetched).  sched_group_barrier pins the order.
    {
      constexpr int UNITS = 2 * T::MT;
      static_assert(T::NT == 2, "TileLN: two activation fragments per sub-step");
      const int ch0 = ((0 + w.hi) ^ sw) * 16, ch1 = ((2 + w.hi) ^ sw) * 16;
      bf16x8 fb[2][T::NT], fa[3];
      auto load_a = [&](int u) { fa[u % 3] = *(const bf16x8*)(tR + (u % T::MT) * 32 * 64 + (u < T::MT ? ch0 : ch1)); };
#pragma unroll
      for (int j = 0; j < T::NT; ++j) fb[0][j] = *(const bf16x8*)(tL + j * 32 * 64 + ch0);
      load_a(0);
      load_a(1);
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + 2 < UNITS) load_a(u + 2);
        if (u == 2) {
#pragma unroll
          for (int j = 0; j < T::NT; ++j) fb[1][j] = *(const bf16x8*)(tL + j * 32 * 64 + ch1);
        }
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
          acc.c[u % T::MT][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u % 3], fb[u / T::MT][j], acc.c[u % T::MT][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, T::NT + 2, 0);          // fb[0], fa(0), fa(1)
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + 2 < UNITS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // fa(u + 2)
        if (u == 2) __builtin_amdgcn_sched_group_barrier(0x100, T::NT, 0);          // fb of sub-step 1
        __builtin_amdgcn_sched_group_barrier(0x008, T::NT, 0);                      // the unit's MFMA pair
      }
    }
#endif
    // the weight slice of step t + 2 once this wave's MFMAs of the step are in the pipe: the ~60-cycle issue stalls then
    // cost no matrix-pipe time (two steps of slack for the landing)
    if (issue_w) ln_stage32<T::TR, LN_W_WAVES, LN_W_FIRST>(srcW, kt + 2, w_dst, w.wave, w_slice_stride);
    CONVDR_LN_STEP(4)
  }

  if (a.dbg_skip_epi) {   // (keeps the accumulators alive; never true in the product)
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc.c[mt][nt][r];
    if (s == 12345.678f) a.X[0] = f32_to_bf16(s);
    return;
  }
  // ---------------- epilogue: + bias + residual, LayerNorm over the 768 features of each token ----------------
  // In the accumulator layout a lane holds 4 features of one token, so direct residual loads / output stores touch
  // 32 rows x 16-32 B per instruction (measured: 23 k + 19 k cycles of a 130 k-cycle workgroup at K = 768).  Both
  // tiles therefore pass through LDS in full rows: the residual half-tile (64 tokens x 1536 B) arrives by LDS-DMA,
  // the normalised half-tile is parked in the same image and leaves 16 B per lane.  Image: row lr = wl * 32 + li,
  // 96 chunks of 16 B, chunk position p holds feature chunk (p & ~31) | ((p ^ lr) & 31) -- the swizzle spreads the
  // per-lane 8-byte accesses (32 rows x the same feature chunk) over the banks.
  CONVDR_LN_TRACE(1)
  __syncthreads();   // operand buffers are dead
  float* sBias = (float*)smem;           // [768]
  float* sGam = sBias + 768;
  float* sBet = sGam + 768;
  float* sRed = sBet + 768;              // [128 tokens][8 partial slots]
  float* sStat = sRed + 128 * 8;         // [128] mean, [128] rstd
  char* sC = smem + 16384;               // [64][1536 B] residual / output image (96 KB; 112 KB in all)
  const int slot = w.wr * 2 + w.hi;
  // residual window [t0, rows) as a buffer: rows past the end read as zeros
  const StageSrc srcR = ln_stage_src(a.R, 768, t0, a.rows, w.wave, w.lane);   // (only .rsrc is used)
  const __amdgpu_buffer_rsrc_t dstX = ln_stage_src(a.X, 768, t0, a.rows, w.wave, w.lane).rsrc;   // output window [t0, rows)
  // byte offset of this lane's 8 bytes of feature chunk `chunk` in the image.  lrl / tid are handed in as opaque
  // per-phase copies: with 192 accumulators live, addresses that the compiler CSEs across phases end up in scratch
  // = lrl * 1536 + (((chunk & ~31) | ((chunk ^ lrl) & 31)) << 4) + hi * 8, evaluated as ONE v_xor per access: the row base
  // lrl * 1536 and the image base have zeros in bits 4-8, so with B = base + lrl * 1536 + ((lrl & 31) << 4) + hi * 8 the
  // address is (B ^ ((chunk & 31) << 4)) + ((chunk & ~31) << 4), the second term a compile-time DS offset.  (The closed
  // form cost ~5 integer instructions per access, 96 accesses per wave: the epilogue is VALU-bound.)
  typedef __attribute__((address_space(3))) u32x2_t lds_u2_t;
  static_assert((16384 & 511) == 0, "the image base must leave bits 4-8 to the swizzle");
  auto img_base = [&](int lrl) { return lds_off(sC) + (uint32_t)(lrl * 1536 + ((lrl & 31) << 4) + w.hi * 8); };
  auto img_at = [&](uint32_t B, int chunk) {
    return (lds_u2_t*)(uintptr_t)((B ^ (uint32_t)((chunk & 31) << 4)) + (uint32_t)((chunk & ~31) << 4));
  };
  auto opaque = [](int x) {
    asm volatile("" : "+v"(x));
    return x;
  };
  CONVDR_LN_TRACE(2)
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    if (nt) __syncthreads();   // every lane has consumed the previous half
    // 6144 chunks: thread t fills positions i * 512 + t (lane-linear per wave instruction, as LDS-DMA requires)
    // (position idx = i * 512 + t is row idx / 96, chunk idx % 96: one division for i = 0, then 512 = 5 * 96 + 32)
    const int tid_a = opaque((int)threadIdx.x);
    int lr = tid_a / 96, pp = tid_a - lr * 96;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int c = (pp & ~31) | ((pp ^ lr) & 31);
      const int tok = (lr >> 5) * 64 + nt * 32 + (lr & 31);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcR.rsrc, (lptr_t)(sC + (i * 512 + w.wave * 64) * 16), 16,
                                               (uint32_t)(tok * 1536 + c * 16), 0, 0, 0);
      pp += 32; lr += 5;
      if (pp >= 96) { pp -= 96; lr += 1; }
    }
    if (nt == 0) {
      // the LayerNorm parameters ride under the first residual half's flight (staged before it, their loads and the
      // residual's round trip were two exposed latencies in a row)
      for (int i = threadIdx.x; i < 768; i += 512) { sBias[i] = a.bias[i]; sGam[i] = a.gamma[i]; sBet[i] = a.beta[i]; }
      CONVDR_LN_TRACE(8)
    }
    lds_dma_wait_all();
    if (nt == 0) { CONVDR_LN_TRACE(9) }
    __syncthreads();           // (first half: also publishes the LayerNorm parameters)
    if (nt == 0) { CONVDR_LN_TRACE(10) }
    float s = 0.f;
    const uint32_t imgB = img_base(opaque(w.wl * 32 + w.li));
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) {
      u32x2_t res[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) res[g] = *img_at(imgB, (w.wr * T::MT + mt) * 4 + g);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x16& v = acc.c[mt][nt];
        const float4 bv = *(const float4*)(sBias + w.r_base(mt, g));
        const u32x2_t r = res[g];
        v[4 * g + 0] += bv.x + __uint_as_float(r.x << 16);
        v[4 * g + 1] += bv.y + __uint_as_float(r.x & 0xffff0000u);
        v[4 * g + 2] += bv.z + __uint_as_float(r.y << 16);
        v[4 * g + 3] += bv.w + __uint_as_float(r.y & 0xffff0000u);
        s += (v[4 * g + 0] + v[4 * g + 1]) + (v[4 * g + 2] + v[4 * g + 3]);
      }
    }
    sRed[(w.wl * 64 + nt * 32 + w.li) * 8 + slot] = s;
    if (nt == 0) { CONVDR_LN_TRACE(11) }
  }
  CONVDR_LN_TRACE(3)
  __syncthreads();
  CONVDR_LN_TRACE(4)
  if (threadIdx.x < 128) {
    const float* p = sRed + threadIdx.x * 8;
    sStat[threadIdx.x] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * (1.f / 768.f);
  }
  __syncthreads();
  float mean[T::NT];
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    mean[nt] = sStat[w.wl * 64 + nt * 32 + w.li];
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc.c[mt][nt][r] - mean[nt];
        q += d * d;
      }
    sRed[(w.wl * 64 + nt * 32 + w.li) * 8 + slot] = q;
  }
  CONVDR_LN_TRACE(5)
  __syncthreads();
  if (threadIdx.x < 128) {
    const float* p = sRed + threadIdx.x * 8;
    const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * (1.f / 768.f);
    sStat[128 + threadIdx.x] = rsqrtf(var + a.eps);
  }
  __syncthreads();
  CONVDR_LN_TRACE(6)
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    if (nt) __syncthreads();   // the previous half has been read out
    const float rstd = sStat[128 + w.wl * 64 + nt * 32 + w.li];
    const uint32_t imgB = img_base(opaque(w.wl * 32 + w.li));
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = w.r_base(mt, g);
        const float4 gg = *(const float4*)(sGam + f), bb = *(const float4*)(sBet + f);
        const f32x16& v = acc.c[mt][nt];
        u32x2_t o;
        o.x = pack_bf16x2((v[4 * g + 0] - mean[nt]) * rstd * gg.x + bb.x, (v[4 * g + 1] - mean[nt]) * rstd * gg.y + bb.y);
        o.y = pack_bf16x2((v[4 * g + 2] - mean[nt]) * rstd * gg.z + bb.z, (v[4 * g + 3] - mean[nt]) * rstd * gg.w + bb.w);
        *img_at(imgB, (w.wr * T::MT + mt) * 4 + g) = o;
      }
    if (nt == 0) { CONVDR_LN_TRACE(12) }
    __syncthreads();
    if (nt == 0) { CONVDR_LN_TRACE(13) }
    const int tid_s = opaque((int)threadIdx.x);
    int lr = tid_s / 96, pp = tid_s - lr * 96;
#pragma unroll
    for (int i0 = 0; i0 < 12; i0 += 4) {
      u32x4_t v4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v4[j] = *(const u32x4_t*)(sC + ((i0 + j) * 512 + tid_s) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (pp & ~31) | ((pp ^ lr) & 31);
        const int tok = (lr >> 5) * 64 + nt * 32 + (lr & 31);
        // through the output window [t0, rows): rows past the end are dropped by the bounds check; one 32-bit offset
        __builtin_amdgcn_raw_buffer_store_b128(v4[j], dstX, (uint32_t)(tok * 1536 + c * 16), 0, NT_LN ? 2 : 0);
        pp += 32; lr += 5;
        if (pp >= 96) { pp -= 96; lr += 1; }
      }
    }
    if (nt == 0) { CONVDR_LN_TRACE(14) }
  }
  CONVDR_LN_TRACE(7)
}

}  // namespace convdr
