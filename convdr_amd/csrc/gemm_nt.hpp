// bf16 MFMA "NT" tile engine for gfx950:  acc[TR x TL] = R[r0.., :] * L[l0.., :]^T
// with both operands row-major bf16 [rows, K] (K contiguous), fp32 accumulate.
//
// This one main loop serves every dense contraction on the ConvDR hot path:
//   * similarity scan     S = P_block * Q^T          (ip_topk.hip; replaces the SGEMM inside
//                                                     faiss.IndexFlatIP.search, run_convdr_inference.py:182)
//   * encoder projections Y = X * W^T (+ epilogue)    (encoder*.hip; nn.Linear weights are [out, in])
//
// The "R" operand's row index lands on accumulator REGISTERS (4 consecutive rows per register quad), the "L"
// operand's row index on LANES (v_mfma_f32_32x32x16_bf16: D[i][j], j = lane & 31, i = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)).
//
// Geometry is a template: WR x WL waves, each owning MT x NT MFMA 32x32 tiles, BK = 64.
//   Tile128: 2 x 2 waves, 2 x 2 tiles -> 128 x 128, 256 threads, 64 KB LDS, 2 workgroups per CU
//   Tile256: 2 x 4 waves, 4 x 2 tiles -> 256 x 256, 512 threads, 128 KB LDS, 1 workgroup per CU; halves the
//            L2->LDS bytes per FLOP and makes one K step long enough (2048 MFMA cycles per SIMD) that the
//            one-step-ahead LDS-DMA prefetch covers HBM latency without a second resident workgroup
// Operand tiles are staged HBM/L2 -> LDS with 16-byte LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ...
// lds), double buffered, one barrier per K step: the loads for step t+1 fly under the MFMAs of step t; in 8-wave
// tiles only the younger wave of each SIMD issues them (TileCfg::DMA_WAVES).  LDS rows are 128 B; the 16-byte chunk
// index is XOR-swizzled with (row >> 1) & 7 so that the ds_read_b128 fragment reads (32 rows x one chunk) are
// bank-conflict free.  LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE offset (it stays
// inside the row's 128-byte line, coalescing is unchanged).
#pragma once
#include "common.hpp"

namespace convdr {

constexpr int GEMM_BK = 64;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
// 16-byte global store, optionally non-temporal (activations that the next kernel streams from HBM anyway: keeping them
// out of L2 leaves it to the weights)
template <bool NT, class V>
__device__ __forceinline__ void store16(void* p, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte store");
  typedef uint32_t raw4_t __attribute__((ext_vector_type(4)));
  if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(raw4_t, v), (raw4_t*)p);
  else *(V*)p = v;
}
// Cache policy of the streamed activation stores (measured one at a time, DESIGN.md section 5.0: 45.22 -> 44.45 ms per
// 2048-passage forward together): what the next kernel streams from HBM anyway stays out of L2, which is left to the
// weights and the activation tile the column tiles of one token tile share.
constexpr bool NT_GEMM_BLK = true;   // blocked GEMM outputs (FFN1's Hm, Q / K)
constexpr bool NT_CTILE = true;      // parked bf16 tile outputs of 128 MB and more (V^T, ...)
constexpr bool NT_ATT = true;        // attention context
constexpr bool NT_LN = true;         // LayerNorm output of the projection + LayerNorm kernel
// the same with a cache policy (aux 2 = nt: data that ONE workgroup reads once)
template <int AUX>
__device__ __forceinline__ void glds16_aux(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, AUX);
}

// all of this wave's outstanding LDS-DMA (and other vector-memory) operations have completed
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() additionally drains the vector-memory queue
// (vmcnt(0)) whenever an LDS-DMA is in flight, which is exactly what must NOT happen in an epilogue that runs under
// the next tile's prefetch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS accesses the compiler must not see.  While an LDS-DMA is in flight hipcc guards every LDS access it knows about
// with s_waitcnt vmcnt(0) (it cannot prove the DMA targets another region), which serialises an epilogue that runs
// under the next tile's prefetch.  These wrappers take 32-bit LDS byte addresses; reads wait for their own data.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_write_b64_hidden(uint32_t addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");   // retired by lds_barrier()'s lgkmcnt(0)
}
__device__ __forceinline__ void lds_read4_b128_hidden(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, u32x4_t& v0,
                                                      u32x4_t& v1, u32x4_t& v2, u32x4_t& v3) {
  asm volatile(
      "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
      : "memory");
}

// One MFMA of the engine: 32x32x16, fp32 accumulate; the operands are 16-bit words the staging never interprets --
// bf16 everywhere except the fp16 rung of the similarity scan (ip_topk.hip), which reads the same LDS image as halfs
// (v_mfma_f32_32x32x16_f16: same rate and register layout, 11 significand bits instead of 8).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int WR_, int WL_, int MT_, int NT_>
struct TileCfg {
  static constexpr int WR = WR_, WL = WL_, MT = MT_, NT = NT_;
  static constexpr int TR = WR * MT * 32, TL = WL * NT * 32;   // tile extent on the R / L operand
  static constexpr int WAVES = WR * WL, THREADS = WAVES * 64;
  // Waves that issue the LDS-DMA of a stage.  An 8-wave workgroup puts two waves on every SIMD; if all eight issue
  // their share right after the barrier, the ~1.3 k cycles the 64 DMA instructions need to drain through the
  // texture path pass with the matrix pipe idle (s_memtime: 544 cycles for the older wave of a SIMD, 1,590 for the
  // younger, then both start their MFMAs).  With only the younger half (waves 4-7) issuing, the older wave of each
  // SIMD runs its MFMAs under the younger's DMA issue and the younger follows.
  static constexpr int DMA_WAVES = WAVES == 8 ? 4 : WAVES;
  static constexpr int DMA_FIRST = WAVES - DMA_WAVES;
  static constexpr int R_BYTES = TR * 128, L_BYTES = TL * 128;  // one K step of each operand
  static constexpr int STAGE_BYTES = R_BYTES + L_BYTES;
  static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
  static constexpr int MIN_WAVES_PER_SIMD = THREADS >= 512 ? 2 : 2;
};
using Tile128 = TileCfg<2, 2, 2, 2>;
using Tile256 = TileCfg<2, 4, 4, 2>;
// 256 x 128: the similarity scan with at most 128 queries (ip_topk.hip).  That regime is HBM-bound -- the passage block
// streams once, the query tile is re-read from L2 -- so what matters is passage bytes in flight per CU: on the R3 K step
// (3 R slots of 32 KB + 2 L slots of 16 KB = 128 KB) two 32 KB passage chunks are in flight per CU, against 2 x 16 KB with
// two 128 x 128 workgroups.
using TileTall = TileCfg<2, 4, 4, 1>;
// 256 x 128 for the encoder GEMMs (k_gemm: 256 features x 128 tokens): 4 x 2 waves of 64 x 64, i.e. 2 x 2 MFMA tiles per
// wave -- 4 fragment reads per 4 MFMAs where TileTall's 128 x 32 wave needs 5 -- on the same R3 K step.  Serves problems
// whose 256 x 256 tiling leaves the 256 CUs a bad last round (or none at all: N = 768 at ~9 k training rows is 108 tiles):
// half the tile, so twice the tiles, at 1.5 x the L2 -> LDS bytes per FLOP of 256 x 256 instead of the 2 x of the
// 128 x 128 engine, and one 8-wave workgroup per CU whose K step is long enough to hide its own barrier.
using TileWide = TileCfg<4, 2, 2, 2>;

template <class T>
struct GemmAcc {
  f32x16 c[T::MT][T::NT];
  // element r of c[mt][nt] for lane l:  R index = wr*MT*32 + mt*32 + (r&3) + 8*(r>>2) + 4*(l>>5)
  //                                      L index = wl*NT*32 + nt*32 + (l&31)
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
      for (int j = 0; j < T::NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
  }
};

template <class T>
struct WavePos {
  int lane, wave, wr, wl, hi, li;
  __device__ __forceinline__ WavePos() : WavePos((int)threadIdx.x) {}
  // from an explicit thread index: epilogues pass an opaque copy (asm volatile("" : "+v"(tid))) so that the address
  // arithmetic derived from it is NOT hoisted out of a persistent tile loop and kept in registers across the main loop
  __device__ __forceinline__ explicit WavePos(int tid) {
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wr = wave / T::WL;
    wl = wave - wr * T::WL;
    hi = lane >> 5;
    li = lane & 31;
  }
  // first of the 4 consecutive R indices held in registers 4g..4g+3 of tile mt
  __device__ __forceinline__ int r_base(int mt, int g) const { return (wr * T::MT + mt) * 32 + 8 * g + 4 * hi; }
  __device__ __forceinline__ int r_index(int mt, int r) const { return (wr * T::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }
  __device__ __forceinline__ int l_index(int nt) const { return (wl * T::NT + nt) * 32 + li; }
};

// Stage rows [row0, row0 + ROWS) x k-chunk kt of G ([nrows, ld] bf16) into an LDS tile of ROWS x 128 B.
// The source goes through a buffer descriptor whose window starts at the tile's first row and ends at the end of
// the operand: rows past nrows-1 are out of range for the hardware bounds check and arrive as zeros (their products
// are never stored), and the whole address is ONE per-lane 32-bit offset (row-in-round x pitch + swizzled 16-byte
// chunk) plus wave-uniform scalar offsets for the round and the K chunk -- no per-round 64-bit address registers in
// the main loop (the flat-address form held 16 of them and pushed the persistent kernels into scratch).
typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));

struct StageSrc {
  __amdgpu_buffer_rsrc_t rsrc;   // window [tile row 0, end of operand)
  uint32_t voff;                 // this lane's byte offset inside a round
  uint32_t round_pitch;          // bytes between rounds (8 * issuing waves rows)
};

template <int WAVES, int FIRST>   // WAVES issuing waves, the first of which is wave FIRST
__device__ __forceinline__ StageSrc gemm_stage_src(const bf16_t* __restrict__ G, int64_t ld, int64_t row0, int64_t nrows,
                                                    int wave, int lane) {
  wave = wave >= FIRST ? wave - FIRST : 0;
  StageSrc s;
  int64_t bytes = (nrows - row0) * ld * 2;
  bytes = bytes < 0 ? 0 : (bytes > 0xffffffffll ? 0xffffffffll : bytes);
  // the descriptor must be provably wave-uniform or hipcc wraps every load in a waterfall loop
  const uint64_t base = (uint64_t)(G + row0 * ld);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
  const int row = wave * 8 + (lane >> 3);                       // row inside a round; rounds are 8 * WAVES rows apart,
  const int gch = (lane & 7) ^ ((row >> 1) & 7);                // a multiple of 16, so the swizzle is round-independent
  s.voff = (uint32_t)(row * ld * 2) + gch * 16;
  s.round_pitch = __builtin_amdgcn_readfirstlane((uint32_t)(8 * WAVES * ld * 2));
  return s;
}

template <int ROWS, int WAVES, int FIRST, int AUX = 0>   // AUX: cache policy of the DMA (2 = nt)
__device__ __forceinline__ void gemm_stage(const StageSrc& s, int kt, char* lds_tile, int wave) {
  if (FIRST > 0 && wave < FIRST) return;   // wave-uniform
  wave -= FIRST;
  constexpr int ROUNDS = ROWS / (8 * WAVES);
  static_assert(ROUNDS * 8 * WAVES == ROWS, "tile rows must be a multiple of 8 * waves");
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lptr_t)(lds_tile + (i * WAVES + wave) * 8 * 128), 16, s.voff,
                                             i * s.round_pitch + kt * (GEMM_BK * 2), 0, AUX);
}

// LDS image: two stages of STAGE_BYTES = R_BYTES + L_BYTES, each [R tile | L tile] (a whole stage is one contiguous
// region, so the idle stage can serve as epilogue scratch while the other already receives the next tile).
template <class T>
struct TileSrc {
  StageSrc R, L;
  __device__ __forceinline__ TileSrc(const bf16_t* __restrict__ Rp, int64_t ldr, int64_t nR, const bf16_t* __restrict__ Lp,
                                     int64_t ldl, int64_t nL, int64_t r0, int64_t l0, const WavePos<T>& w)
      : R(gemm_stage_src<T::DMA_WAVES, T::DMA_FIRST>(Rp, ldr, r0, nR, w.wave, w.lane)),
        L(gemm_stage_src<T::DMA_WAVES, T::DMA_FIRST>(Lp, ldl, l0, nL, w.wave, w.lane)) {}
};

template <class T>
__device__ __forceinline__ void gemm_issue_stage(const TileSrc<T>& src, int kt, char* stage, const WavePos<T>& w) {
  gemm_stage<T::TR, T::DMA_WAVES, T::DMA_FIRST>(src.R, kt, stage, w.wave);
  gemm_stage<T::TL, T::DMA_WAVES, T::DMA_FIRST>(src.L, kt, stage + T::R_BYTES, w.wave);
}

// acc += R[r0:r0+TR, :] * L[l0:l0+TL, :]^T   (K must be a multiple of 64)
// first_buf: the stage that holds (or receives) K chunk 0; stage0_in_flight: the caller has already issued it with
// gemm_issue_stage (cross-tile prefetch of a persistent kernel); stage0_landed: ... and waited for it (vmcnt).  Returns the stage NOT read by the last K step: once
// a wave is back from this call it may issue DMA into that stage (every wave has passed the last barrier, so all
// reads of it are done); the other stage may be reused only after one more __syncthreads().
template <class T, bool F16 = false>
__device__ __forceinline__ int gemm_nt_mainloop(const TileSrc<T>& src, int K, char* smem, GemmAcc<T>& acc,
                                                const WavePos<T>& w, int first_buf = 0, bool stage0_in_flight = false,
                                                bool stage0_landed = false, unsigned long long* step_trace = nullptr) {
  const int nk = K / GEMM_BK;
  const int sw = (w.lane >> 1) & 7;
  const int offR = (w.wr * T::MT * 32 + w.li) * 128;
  const int offL = T::R_BYTES + (w.wl * T::NT * 32 + w.li) * 128;

  if (!stage0_in_flight) gemm_issue_stage<T>(src, 0, smem + first_buf * T::STAGE_BYTES, w);

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = (kt + first_buf) & 1;
    // Tile kt must have LANDED in LDS for every wave before anyone reads it.  LDS-DMA completion is tracked by
    // vmcnt; hipcc's automatic wait before the barrier is NOT reliable for LDS-DMA (observed missing in one of two
    // inlined copies of this loop -> rare stale operand rows), so drain explicitly.
    // (stage0_landed: the caller has already waited for chunk 0 -- before issuing its epilogue stores, so that this
    // wait, which is in issue order, does not sit behind their write acknowledgements)
#ifdef CONVDR_ENABLE_TRACE   // stamps of K step 6 for lane 0 of every wave: [wave][0..4]
#define CONVDR_STEP_TRACE(i) \
  if (step_trace && kt == 6 + (i) / 5 && w.lane == 0) step_trace[w.wave * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_STEP_TRACE(i)
#endif
    CONVDR_STEP_TRACE(5)   // (top of step 7)
    CONVDR_STEP_TRACE(0)
    if (kt > 0 || !stage0_landed) lds_dma_wait_all();
    CONVDR_STEP_TRACE(1)
    lds_barrier();  // ... and every wave is done reading buffer buf^1 (step kt-1)
    CONVDR_STEP_TRACE(2)
    if (kt + 1 < nk) gemm_issue_stage<T>(src, kt + 1, smem + (buf ^ 1) * T::STAGE_BYTES, w);
    CONVDR_STEP_TRACE(3)
    const char* tR = smem + buf * T::STAGE_BYTES + offR;
    const char* tL = smem + buf * T::STAGE_BYTES + offL;
    // Fragments of 16-wide K sub-step s + 1 are read while the MFMAs of sub-step s run (two register sets).  Left to
    // itself hipcc reuses one small set and parks an LDS round trip (s_waitcnt lgkmcnt(0..2)) in front of every
    // second MFMA pair -- ~240 idle pipe cycles per sub-step per wave in the s_memtime trace.
    bf16x8 fa[2][T::MT], fb[2][T::NT];
    auto load_frags = [&](int s, int set) {
      const int ch = ((2 * s + w.hi) ^ sw) * 16;
#pragma unroll
      for (int j = 0; j < T::NT; ++j) fb[set][j] = *(const bf16x8*)(tL + j * 32 * 128 + ch);
#pragma unroll
      for (int i = 0; i < T::MT; ++i) fa[set][i] = *(const bf16x8*)(tR + i * 32 * 128 + ch);
    };
    load_frags(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
#if defined(CONVDR_FRAG_BLOCK) || defined(CONVDR_FRAG_BLOCK_2STAGE)
      __builtin_amdgcn_sched_barrier(0);   // the prefetch stays ahead of this sub-step's MFMAs
#endif
#pragma unroll
      for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
          acc.c[i][j] = mfma_32x32x16<F16>(fa[s & 1][i], fb[s & 1][j], acc.c[i][j]);
#if !defined(CONVDR_FRAG_BLOCK) && !defined(CONVDR_FRAG_BLOCK_2STAGE)
      if (s + 1 < 4) {   // fragment reads of sub-step s + 1 threaded between this sub-step's MFMAs (see the R3 loop below)
#pragma unroll
        for (int r = 0; r < T::MT + T::NT && r < T::MT * T::NT; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, T::MT + T::NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, T::MT * T::NT, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    CONVDR_STEP_TRACE(4)
  }
  return (nk + first_buf) & 1;
}

// ------------------------------------------------------------------------------------------------------------------
// The "R3" K step: THREE R slots and TWO L slots (3 x 32 KB + 2 x 32 KB = the whole 160 KB for 256 x 256 tiles; tiles with a
// 128-row L operand leave 32 KB over).  The L chunk of step t + 1 is issued right after the first fragment reads (the issue
// stalls pass under the LDS round trip) and the R chunk of step t + 2 AFTER the MFMAs of step t; counted vmcnt as in
// gemm_ln.hpp.  No wave has a block of DMA issue in front of its MFMAs any more (in the two-stage loop the issuing wave of
// a SIMD is the critical path of the step: 1.15 k cycles of issue, then its MFMAs, while its partner idles), and the two
// waves of a SIMD run their MFMA phases together.  Measured on the scan (same box, A/B): 1.600 -> 1.502 ms emitting,
// 1.327 -> 1.266 ms with +inf thresholds.
// Who issues the LDS-DMA of the R3 K step (8-wave tiles): the OLDER wave of each SIMD (waves 0 .. WAVES / 2 - 1) issues
// the whole R chunk of step t + 2 after its MFMAs, the YOUNGER the whole L chunk of step t + 1 at the top of the step.
// The s_memtime anatomy of a step (tools/dbg/gemm_trace_blk.py) shows the older wave -- served first by the matrix pipe --
// done with its 32 MFMAs after ~1.25 k cycles and then ~1.15 k cycles in the barrier, while the younger needs ~2.0 k: with
// roles the R issue rides in the older wave's idle time and the older wave starts its MFMAs without L issue in front.
// (Variants that thread the DMA instructions between the MFMA rows, s_setprio patterns and every-wave-its-share issue
// were measured null or slower in rounds 2-3: DESIGN.md section 5.0; they live in the git history, not here.)
template <class T>
struct R3Issue {
  static constexpr bool ROLES = T::WAVES == 8;
  static constexpr int RW = ROLES ? T::WAVES / 2 : T::WAVES;    // waves issuing an R chunk (from wave 0)
  static constexpr int LW = ROLES ? T::WAVES / 2 : T::WAVES;    // waves issuing an L chunk ...
  static constexpr int LFIRST = ROLES ? T::WAVES / 2 : 0;       // ... from this wave on
  static constexpr int R_DPW = T::TR / (8 * RW);                // DMA instructions per issuing wave per R chunk
  static __device__ __forceinline__ bool issues_r(int wave) { return !ROLES || wave < RW; }
  template <int AUX = 0>
  static __device__ __forceinline__ void r(const StageSrc& s, int kt, char* dst, int wave) {
    if (issues_r(wave)) gemm_stage<T::TR, RW, 0, AUX>(s, kt, dst, wave);
  }
  static __device__ __forceinline__ void l(const StageSrc& s, int kt, char* dst, int wave) {
    gemm_stage<T::TL, LW, LFIRST>(s, kt, dst, wave);   // (returns at once for waves below LFIRST)
  }
};
template <class T>
struct TileSrcAll {   // staging sources of the R3 loop (issue by role, see R3Issue)
  StageSrc R, L;
  __device__ __forceinline__ TileSrcAll(const bf16_t* __restrict__ Rp, int64_t ldr, int64_t nR, const bf16_t* __restrict__ Lp,
                                        int64_t ldl, int64_t nL, int64_t r0, int64_t l0, const WavePos<T>& w)
      : R(gemm_stage_src<R3Issue<T>::RW, 0>(Rp, ldr, r0, nR, w.wave, w.lane)),
        L(gemm_stage_src<R3Issue<T>::LW, R3Issue<T>::LFIRST>(Lp, ldl, l0, nL, w.wave, w.lane)) {}
};
// Slot state of the R3 loop: R chunk 0 of a tile goes to R slot `rs`, chunk 1 to rs + 1 (mod 3), L chunk 0 to L slot `ls`.
struct R3Slots { int rs, ls; };
// the first three DMA groups of a tile (R chunk 0, L chunk 0, R chunk 1 -- in this order: the counted waits rely on it)
// with_r1 = false leaves R chunk 1 to step 0 of the main loop (r1_deferred): its slot then stays untouched until every
// wave has entered the loop -- k_gemm keeps the tile's bias slice there.
// R_AUX: cache policy of the R operand's DMA (2 = nt, for an R operand that is streamed exactly once: the passage block
// of a scan with a single query tile)
template <class T, int R_AUX = 0>
__device__ __forceinline__ void gemm_r3_prologue(const TileSrcAll<T>& src, int K, char* smem, const WavePos<T>& w, R3Slots s,
                                                 bool with_r1 = true) {
  char* sR = smem;
  char* sL = smem + 3 * T::R_BYTES;
  R3Issue<T>::template r<R_AUX>(src.R, 0, sR + s.rs * T::R_BYTES, w.wave);
  R3Issue<T>::l(src.L, 0, sL + s.ls * T::L_BYTES, w.wave);
  if (with_r1 && K > GEMM_BK) R3Issue<T>::template r<R_AUX>(src.R, 1, sR + (s.rs == 2 ? 0 : s.rs + 1) * T::R_BYTES, w.wave);
}
// Returns the slot state for the NEXT tile: once a wave is back from this call, the R slots `rs`, rs + 1 and the L slot
// `ls` of the returned state are free (the last step read the other ones), so the next tile's prologue may be issued
// at once -- under this tile's epilogue (prologue_in_flight on the next call).
template <class T, bool F16 = false, int R_AUX = 0>
__device__ __forceinline__ R3Slots gemm_nt_mainloop_r3(const TileSrcAll<T>& src, int K, char* smem, GemmAcc<T>& acc,
                                                       const WavePos<T>& w, R3Slots st = R3Slots{0, 0},
                                                       bool prologue_in_flight = false, bool r1_deferred = false,
                                                       bool stage0_landed = false, unsigned long long* step_trace = nullptr) {
  constexpr int R_DPW = R3Issue<T>::R_DPW;   // DMA instructions per issuing wave per R chunk
  const bool r_wave = R3Issue<T>::issues_r(w.wave);   // (wave-uniform) this wave has R chunks in flight
  const int nk = K / GEMM_BK;
  const int sw = (w.lane >> 1) & 7;
  const int offR = (w.wr * T::MT * 32 + w.li) * 128;
  const int offL = (w.wl * T::NT * 32 + w.li) * 128;
  char* sR = smem;
  char* sL = smem + 3 * T::R_BYTES;
  if (!prologue_in_flight) gemm_r3_prologue<T, R_AUX>(src, K, smem, w, st, !r1_deferred);
  int rs = st.rs, ls = st.ls;
#ifdef CONVDR_ENABLE_TRACE   // stamps of K step 6 (and the top of step 7) for lane 0 of every wave: [wave][0..6]
#define CONVDR_R3_STEP(i) \
  if (step_trace && kt == 6 + (i) / 6 && w.lane == 0) step_trace[w.wave * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_R3_STEP(i)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    CONVDR_R3_STEP(6)
    CONVDR_R3_STEP(0)
    // (r1_deferred: at step 0 only chunks 0 are in flight -- there is no newer R group to leave outstanding;
    //  stage0_landed: the caller has already waited for them, ahead of its epilogue's stores)
    if (kt == 0 && (r1_deferred || stage0_landed)) {
      if (!stage0_landed) lds_dma_wait_all();
    } else if (kt + 1 < nk && r_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R_DPW) : "memory");
    else lds_dma_wait_all();   // (a wave without R chunks has only the L chunk of this step outstanding)
    CONVDR_R3_STEP(1)
    lds_barrier();
    CONVDR_R3_STEP(2)
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
    const bool issue_l = kt + 1 < nk, issue_r = kt + 2 < nk;
    char* l_dst = sL + (ls ^ 1) * T::L_BYTES;
    const int rnext = rs == 0 ? 2 : rs - 1;   // (kt + 2) % 3
    char* r_dst = sR + rnext * T::R_BYTES;
    // the L chunk of step t + 1 as a block under the first fragments' LDS round trip, the R chunk of step t + 2 as a
    // block after the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    if (r1_deferred && kt == 0 && issue_l) R3Issue<T>::template r<R_AUX>(src.R, 1, sR + (rs == 2 ? 0 : rs + 1) * T::R_BYTES, w.wave);
    if (issue_l) R3Issue<T>::l(src.L, kt + 1, l_dst, w.wave);
    CONVDR_R3_STEP(3)
    // Round 5: the fragment reads of sub-step s + 1 are THREADED between the MFMAs of sub-step s -- one ds_read_b128 behind
    // each of the first MT + NT MFMAs (sched_group_barrier) -- instead of issued as a block in front of them.  As a block the
    // six reads of all eight waves hit the LDS together (48 KB per sub-step at once) and every wave's first MFMA of the
    // sub-step waited out that queue; threaded, the LDS sees one read per wave per ~32 cycles.  Main loop of the stand-alone
    // prototype (tools/proto/w16_proto.hip): 2,744 -> 2,569 ticks per K step (2,485 with no fragment reads at all).
    // CONVDR_FRAG_BLOCK=1 at compile time gives the round-2..4 form back (A/B builds).
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
#ifdef CONVDR_FRAG_BLOCK
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
          acc.c[i][j] = mfma_32x32x16<F16>(fa[s & 1][i], fb[s & 1][j], acc.c[i][j]);
#ifndef CONVDR_FRAG_BLOCK
      if (s + 1 < 4) {
#pragma unroll
        for (int r = 0; r < T::MT + T::NT && r < T::MT * T::NT; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x100, T::MT + T::NT, 0);   // (tiles with fewer MFMAs than reads: the rest of the reads)
        __builtin_amdgcn_sched_group_barrier(0x008, T::MT * T::NT, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    CONVDR_R3_STEP(4)
    if (issue_r) R3Issue<T>::template r<R_AUX>(src.R, kt + 2, r_dst, w.wave);
    CONVDR_R3_STEP(5)
    rs = rs == 2 ? 0 : rs + 1;
    ls ^= 1;
  }
  return R3Slots{rs, ls};
}

// convenience form: contraction range [k_begin, k_begin + K) of R[r0.., :] and L[l0.., :]
template <class T>
__device__ __forceinline__ int gemm_nt_mainloop(const bf16_t* __restrict__ R, int64_t ldr, int64_t nR,
                                                const bf16_t* __restrict__ L, int64_t ldl, int64_t nL, int K,
                                                int64_t r0, int64_t l0, char* smem, GemmAcc<T>& acc,
                                                const WavePos<T>& w, int k_begin = 0) {
  const TileSrc<T> src(R + k_begin, ldr, nR, L + k_begin, ldl, nL, r0, l0, w);
  return gemm_nt_mainloop<T>(src, K, smem, acc, w);
}

}  // namespace convdr
