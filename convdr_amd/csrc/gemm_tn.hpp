// bf16 MFMA "TN" tile engine for gfx950:  acc[TR x TL] = sum_t R[t, r0 + .] * L[t, l0 + .]
// with both operands row-major bf16 [rows, ld] and the CONTRACTION running over the rows (t) -- the shape of every
// weight gradient of the training step,
//     dW[n, k] = sum_t dY[t, n] X[t, k]          (/root/reference/drivers/run_convdr_train.py:178, loss.backward())
// straight from the token-major activations / activation gradients the forward and the dgrad chain leave in HBM.
// Round 1 materialised both operands transposed (k_transpose_bf16, 8 launches + 2 x the operand bytes per layer) to
// reuse the NT engine; here the transposition happens in the LDS fragment read:
//   * an operand K step is staged as it lies in memory, [64 t][TW columns] (TW * 2 bytes per row), by the same 16-byte
//     LDS-DMA through a buffer descriptor as gemm_nt.hpp (rows past the end of the operand and columns past the end of
//     a row arrive as zeros);
//   * a lane's MFMA fragment -- 8 consecutive contraction indices of ONE column -- is two ds_read_b64_tr_b16: within a
//     16-lane group lane i hands in the address of row (i >> 2), columns 4 (i & 3) .. + 3 of a 4 x 16 block and gets back
//     column i of that block (4 rows).  Groups 0 / 1 of a wave cover columns 0-15 / 16-31 of a 32-column MFMA block for
//     t = 0..3 | 4..7, groups 2 / 3 the same columns for t = 8..11 | 12..15: exactly the 32x32x16 A / B operand layout
//     (lane & 31 = column, lane >> 5 = which 8 of the 16 contraction slots).  Both operands use the same slot <-> t map,
//     which is all the contraction needs.
//   * bank conflicts: the two reads of a 32-lane half touch 4 rows x 64 B; rows are a multiple of 256 B apart, so the
//     64-byte segment index is XOR-ed with (t & 3) -- on the DMA source offset, LDS-DMA writes lane-linear -- and the 4
//     rows land in the 4 segments of a 256-byte bank row.
// Output: fp32 tile stored as out[l * ld + r] (r contiguous, 16 bytes per lane), optionally one slab per split of the
// contraction range (deterministic split-K: slabs are summed in a fixed order by k_reduce_partials).
#pragma once
#include "gemm_nt.hpp"

namespace convdr {

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_v;

template <class T>
struct TnCfg {
  static constexpr int R_ROWB = T::TR * 2, L_ROWB = T::TL * 2;          // bytes per staged row
  static constexpr int R_BYTES = 64 * R_ROWB, L_BYTES = 64 * L_ROWB;    // one K step (64 t) of each operand
  static constexpr int STAGE_BYTES = R_BYTES + L_BYTES;
  // 8-wave tiles run the R3 K step (three R slots + two L slots, gemm_tn_mainloop_r3: 160 KB for 256 x 256), 4-wave tiles the
  // two-stage loop (two workgroups per CU)
  static constexpr bool R3 = T::WAVES == 8;
  static constexpr int SMEM_BYTES = R3 ? 3 * R_BYTES + 2 * L_BYTES : 2 * STAGE_BYTES;
};

// Issue by role: in the 8-wave tile the YOUNGER wave of each SIMD (waves 4-7) issues the whole next K step's DMA at the top
// of the step -- the matrix pipe serves the older wave first, so the younger is not on the critical path there -- and the
// older wave goes straight to its fragments (see R3Issue in gemm_nt.hpp).
template <class T>
struct TnIssue {
  static constexpr bool ROLES = T::WAVES == 8;
  static constexpr int W = ROLES ? T::WAVES / 2 : T::WAVES;      // issuing waves ...
  static constexpr int FIRST = ROLES ? T::WAVES / 2 : 0;        // ... from this one on
};

// Stage source of one operand: window [first t of the split, end of operand), this lane's byte offset inside a DMA
// instruction's rows (with the segment swizzle), column guard folded into the offset.
template <int TW, int WAVES>
struct TnStageSrc {
  static constexpr int LPR = TW / 8;              // lanes (16-byte chunks) per row
  static constexpr int RPI = 64 / LPR;            // rows per wave instruction
  static constexpr int ROUNDS = 64 / (RPI * WAVES);
  static_assert(LPR <= 64 && RPI * LPR == 64 && ROUNDS * RPI * WAVES == 64, "tile width must be 64..512 columns");
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff, round_pitch, step_pitch;
  __device__ __forceinline__ TnStageSrc(const bf16_t* __restrict__ G, int64_t ld, int64_t ncols, int64_t col0, int64_t t_begin,
                                        int64_t nrows, int wave, int lane) {
    int64_t bytes = (nrows - t_begin) * ld * 2;
    bytes = bytes < 0 ? 0 : (bytes > 0x7fffffffll ? 0x7fffffffll : bytes);
    const uint64_t base = (uint64_t)(G + t_begin * ld + col0);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    const int row = wave * RPI + lane / LPR;                 // row inside a round (a multiple of 4 rows per instruction
    const int cpos = lane % LPR;                             //  or RPI < 4: then ROUNDS keeps row & 3 round-independent)
    const int gch = cpos ^ ((row & 3) << 2);                 // LDS chunk cpos of this row holds source chunk gch
    voff = (uint32_t)(row * ld * 2) + gch * 16;
    if (col0 + gch * 8 >= ncols) voff = 0x80000000u;         // column past the end of the row: out of range -> zeros
    round_pitch = __builtin_amdgcn_readfirstlane((uint32_t)(RPI * WAVES * ld * 2));
    step_pitch = __builtin_amdgcn_readfirstlane((uint32_t)(64 * ld * 2));
    static_assert((RPI * WAVES) % 4 == 0, "rounds must keep (row & 3)");
  }
  __device__ __forceinline__ void issue(int kt, char* lds_tile, int wave) const {   // wave: index among the issuing waves
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds_tile + (i * WAVES + wave) * 1024), 16, voff,
                                               i * round_pitch + kt * step_pitch, 0, 0);
  }
};

// One fragment = two transposing reads at byte offsets OFF and OFF + 4 rows from the lane's base address.  Inline asm:
// hipcc treats the ds_read_tr builtin as a read of unknown memory and parks s_waitcnt vmcnt(0) in front of it while an
// LDS-DMA is in flight -- the prefetch of the next K step would be drained before the first fragment read of this one.
// The caller counts lgkmcnt itself (tn_wait_frags).
union TnFrag { bf16x8 v; u32x2_t h[2]; };
template <int OFF, int ROWB>
__device__ __forceinline__ void tn_frag(uint32_t addr, TnFrag& f) {
  static_assert(OFF + 4 * ROWB < 65536, "ds offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(f.h[0]), "=&v"(f.h[1])
               : "v"(addr), "n"(OFF), "n"(OFF + 4 * ROWB)
               : "memory");
}
// All but the newest N LDS reads of this wave have returned: fragment set (fa[0..MT), fb[0..NT)) is valid.  The statement
// names the set's registers "+v": hipcc does not count an asm load, so every consumer of the set -- and any register copy
// the allocator wants to make of it -- must come after this wait (cdna guide 5.7 item 1, form ii); the sched_barrier keeps
// the register-only MFMAs below it (rule 18).
template <int N, int MT, int NT>
__device__ __forceinline__ void tn_wait_frags(TnFrag* fa, TnFrag* fb) {
  static_assert(NT == 2 && (MT == 2 || MT == 4), "fragment-set shapes of Tile128 / Tile256");
  if constexpr (MT == 2) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(fa[0].h[0]), "+v"(fa[0].h[1]), "+v"(fa[1].h[0]), "+v"(fa[1].h[1]), "+v"(fb[0].h[0]), "+v"(fb[0].h[1]),
                   "+v"(fb[1].h[0]), "+v"(fb[1].h[1])
                 : "n"(N)
                 : "memory");
  } else {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(fa[0].h[0]), "+v"(fa[0].h[1]), "+v"(fa[1].h[0]), "+v"(fa[1].h[1]), "+v"(fa[2].h[0]), "+v"(fa[2].h[1]),
                   "+v"(fa[3].h[0]), "+v"(fa[3].h[1]), "+v"(fb[0].h[0]), "+v"(fb[0].h[1]), "+v"(fb[1].h[0]), "+v"(fb[1].h[1])
                 : "n"(N)
                 : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
}

// acc += sum over K steps [0, nk) of the staged operands.  Two stages, one barrier per step; every wave issues its share
// of step kt + 1 right after the barrier of step kt.
template <class T>
__device__ __forceinline__ void gemm_tn_mainloop(const TnStageSrc<T::TR, TnIssue<T>::W>& srcR,
                                                 const TnStageSrc<T::TL, TnIssue<T>::W>& srcL,
                                                 int nk, char* smem, GemmAcc<T>& acc, const WavePos<T>& w) {
  const bool issuer = w.wave >= TnIssue<T>::FIRST;            // wave-uniform
  const int iw = w.wave - TnIssue<T>::FIRST;
  using C = TnCfg<T>;
  // this lane's byte offset inside an operand image for MFMA column block 0 of its wave, sub-step 0, first read:
  //   row = 8 (lane >> 5) + ((lane & 15) >> 2),  column = 16 ((lane >> 4) & 1) + 4 (lane & 3), segment swizzle by row & 3
  const int i16 = w.lane & 15, g = w.lane >> 4;
  const int rsw = i16 >> 2;
  const int trow = 8 * (g >> 1) + rsw;
  const int cb = 32 * (g & 1) + 8 * (i16 & 3);   // byte offset of the column inside its 64-byte segment
  uint32_t offR[T::MT], offL[T::NT];
  const uint32_t s0 = lds_off(smem);
#pragma unroll
  for (int i = 0; i < T::MT; ++i) {
    const int seg = w.wr * T::MT + i;            // 64-byte segment (= 32-column MFMA block) inside the row
    offR[i] = s0 + trow * C::R_ROWB + (((seg & ~3) | ((seg ^ rsw) & 3)) << 6) + cb;
  }
#pragma unroll
  for (int j = 0; j < T::NT; ++j) {
    const int seg = w.wl * T::NT + j;
    offL[j] = s0 + C::R_BYTES + trow * C::L_ROWB + (((seg & ~3) | ((seg ^ rsw) & 3)) << 6) + cb;
  }
  constexpr int NFRAG_READS = 2 * (T::MT + T::NT);   // LDS instructions per fragment set
  if (issuer) {
    srcR.issue(0, smem, iw);
    srcL.issue(0, smem + C::R_BYTES, iw);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    lds_dma_wait_all();
    lds_barrier();
    if (kt + 1 < nk && issuer) {
      srcR.issue(kt + 1, smem + (buf ^ 1) * C::STAGE_BYTES, iw);
      srcL.issue(kt + 1, smem + (buf ^ 1) * C::STAGE_BYTES + C::R_BYTES, iw);
    }
    const uint32_t sb = buf * C::STAGE_BYTES;
    TnFrag fa[2][T::MT], fb[2][T::NT];
#define CONVDR_TN_LOAD(S, SET)                                                                          \
  {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < T::NT; ++j) tn_frag<16 * (S) * C::L_ROWB, C::L_ROWB>(offL[j] + sb, fb[SET][j]); \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i) tn_frag<16 * (S) * C::R_ROWB, C::R_ROWB>(offR[i] + sb, fa[SET][i]); \
  }
#define CONVDR_TN_MMA(SET)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < T::NT; ++j)                                                 \
        acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i].v, fb[SET][j].v, acc.c[i][j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
    CONVDR_TN_LOAD(0, 0)
    CONVDR_TN_LOAD(1, 1)
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[0], fb[0]);
    CONVDR_TN_MMA(0)
    CONVDR_TN_LOAD(2, 0)
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[1], fb[1]);
    CONVDR_TN_MMA(1)
    CONVDR_TN_LOAD(3, 1)
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[0], fb[0]);
    CONVDR_TN_MMA(0)
    tn_wait_frags<0, T::MT, T::NT>(fa[1], fb[1]);
    CONVDR_TN_MMA(1)
#undef CONVDR_TN_LOAD
#undef CONVDR_TN_MMA
  }
}

// The R3 K step for the TN engine (round 5; gemm_nt.hpp has the NT form and the reasoning).  In the two-stage loop the younger
// wave of each SIMD issued the WHOLE next K step -- 16 LDS-DMA instructions, ~1.2 k cycles through the texture path -- before
// its own fragments, and a K step took ~3.4 k cycles for 2,048 of matrix work.  Here: three R slots + two L slots = the whole
// 160 KB; the YOUNGER waves (4-7) issue the L chunk of step t + 1 (8 instructions each) under their first fragment reads, the
// OLDER waves (0-3) the R chunk of step t + 2 after their MFMAs, in the time they used to spend in the barrier; counted vmcnt
// (an R wave leaves its newest group in flight).
template <class T>
__device__ __forceinline__ void gemm_tn_mainloop_r3(const TnStageSrc<T::TR, T::WAVES / 2>& srcR,
                                                    const TnStageSrc<T::TL, T::WAVES / 2>& srcL, int nk, char* smem,
                                                    GemmAcc<T>& acc, const WavePos<T>& w) {
  using C = TnCfg<T>;
  constexpr int HW = T::WAVES / 2;
  const bool r_wave = w.wave < HW;                             // wave-uniform
  const int iw = r_wave ? w.wave : w.wave - HW;
  constexpr int R_DPW = TnStageSrc<T::TR, HW>::ROUNDS;         // DMA instructions per R wave per chunk
  const int i16 = w.lane & 15, g = w.lane >> 4;
  const int rsw = i16 >> 2;
  const int trow = 8 * (g >> 1) + rsw;
  const int cb = 32 * (g & 1) + 8 * (i16 & 3);
  uint32_t offR[T::MT], offL[T::NT];
  const uint32_t s0 = lds_off(smem);
#pragma unroll
  for (int i = 0; i < T::MT; ++i) {
    const int seg = w.wr * T::MT + i;
    offR[i] = s0 + trow * C::R_ROWB + (((seg & ~3) | ((seg ^ rsw) & 3)) << 6) + cb;
  }
#pragma unroll
  for (int j = 0; j < T::NT; ++j) {
    const int seg = w.wl * T::NT + j;
    offL[j] = s0 + 3 * C::R_BYTES + trow * C::L_ROWB + (((seg & ~3) | ((seg ^ rsw) & 3)) << 6) + cb;
  }
  constexpr int NFRAG_READS = 2 * (T::MT + T::NT);
  char* sR = smem;
  char* sL = smem + 3 * C::R_BYTES;
  if (r_wave) {
    srcR.issue(0, sR, iw);
    if (nk > 1) srcR.issue(1, sR + C::R_BYTES, iw);
  } else {
    srcL.issue(0, sL, iw);
  }
  int rs = 0, ls = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (r_wave && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R_DPW) : "memory");
    else lds_dma_wait_all();
    lds_barrier();
    const uint32_t sbR = rs * C::R_BYTES, sbL = ls * C::L_BYTES;
    TnFrag fa[2][T::MT], fb[2][T::NT];
#define CONVDR_TN_LOAD(S, SET)                                                                          \
  {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < T::NT; ++j) tn_frag<16 * (S) * C::L_ROWB, C::L_ROWB>(offL[j] + sbL, fb[SET][j]); \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i) tn_frag<16 * (S) * C::R_ROWB, C::R_ROWB>(offR[i] + sbR, fa[SET][i]); \
  }
#define CONVDR_TN_MMA(SET)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < T::NT; ++j)                                                 \
        acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i].v, fb[SET][j].v, acc.c[i][j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
// MMA(SET) with the reads of sub-step S (into the SAME set) threaded through it: the R fragment of row i is re-read as soon as
// the row's MFMAs have issued, the L fragments after the last row.  As one block in front of the next sub-step's MFMAs the
// twelve transposing reads of all eight waves hit the LDS together and held up every wave's first MFMA (gemm_nt.hpp, round 5).
// The waits stay as they were: LDS reads return in order, so "all but the newest 12" still names the other set.
#define CONVDR_TN_MMA_LOAD(SET, S)                                                                      \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < T::MT; ++i) {                                                 \
      _Pragma("unroll") for (int j = 0; j < T::NT; ++j)                                                 \
        acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i].v, fb[SET][j].v, acc.c[i][j], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      tn_frag<16 * (S) * C::R_ROWB, C::R_ROWB>(offR[i] + sbR, fa[SET][i]);                              \
    }                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < T::NT; ++j) tn_frag<16 * (S) * C::L_ROWB, C::L_ROWB>(offL[j] + sbL, fb[SET][j]); \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
    CONVDR_TN_LOAD(0, 0)
    CONVDR_TN_LOAD(1, 1)
    // the L chunk of step t + 1 under the first fragments' LDS round trip (its slot was read in step t - 1: free since the barrier)
    if (!r_wave && kt + 1 < nk) srcL.issue(kt + 1, sL + (ls ^ 1) * C::L_BYTES, iw);
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[0], fb[0]);
#ifdef CONVDR_TN_FRAG_BLOCK
    CONVDR_TN_MMA(0)
    CONVDR_TN_LOAD(2, 0)
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[1], fb[1]);
    CONVDR_TN_MMA(1)
    CONVDR_TN_LOAD(3, 1)
#else
    CONVDR_TN_MMA_LOAD(0, 2)
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[1], fb[1]);
    CONVDR_TN_MMA_LOAD(1, 3)
#endif
    tn_wait_frags<NFRAG_READS, T::MT, T::NT>(fa[0], fb[0]);
    CONVDR_TN_MMA(0)
    tn_wait_frags<0, T::MT, T::NT>(fa[1], fb[1]);
    CONVDR_TN_MMA(1)
#undef CONVDR_TN_LOAD
#undef CONVDR_TN_MMA
#undef CONVDR_TN_MMA_LOAD
    // the R chunk of step t + 2 once this wave's MFMAs are in the pipe (slot (t + 2) % 3 was read in step t - 1)
    if (r_wave && kt + 2 < nk) srcR.issue(kt + 2, sR + (rs == 0 ? 2 : rs - 1) * C::R_BYTES, iw);
    rs = rs == 2 ? 0 : rs + 1;
    ls ^= 1;
  }
}

// One launch serves up to TN_MAX_PROBLEMS independent products (the four weight gradients of an encoder layer have the
// same contraction length and become available within one layer of the backward chain): blockIdx.x walks the
// concatenated tile lists, blockIdx.y the slices of the contraction range.
constexpr int TN_MAX_PROBLEMS = 8;
struct GemmTnProblem {
  const bf16_t* Rm;    // [rows, ldr]: its columns land on accumulator registers -> contiguous output index (k of dW[n, k])
  const bf16_t* Lm;    // [rows, ldl]: its columns land on lanes -> output row (n of dW[n, k])
  float* out;          // nsplit == 1: dW [NL][NR], accumulated into (+=; GemmTnArgs::overwrite: =);  nsplit > 1: slabs [nsplit][NL][NR], overwritten
  int64_t ldr, ldl;
  int NR, NL;          // columns of Rm / Lm that take part
  int tilesR;          // tiles along NR
  int tile_end;        // one past this problem's last tile in the concatenated order
};
struct GemmTnArgs {
  GemmTnProblem p[TN_MAX_PROBLEMS];
  int count;
  int64_t rows;        // contraction length (common to the batch)
  int steps_per_split; // K steps (of 64 rows) per blockIdx.y
  int nsplit;
  // Ordered in-place accumulation (nsplit > 1 and flags != nullptr): slice y of a tile adds its partial product into dW
  // itself, AFTER slice y - 1 has (flags[tile] counts the slices that are done; zeroed before the launch).  The sum is
  // evaluated in slice order whatever the timing -- deterministic like the slab form -- without the slab round trip
  // (write + re-read of nsplit x the gradient) and its reduction launches.  It is what lets a contraction of ~140 K steps
  // (configs[2]: 9 k token rows) be cut in two: 108 tiles of 256^2 fill 42 % of the CUs, 216 fill 84 %.  No deadlock:
  // workgroups are dispatched in (y, x) order, so a waiting slice-y workgroup implies every slice-(y - 1) workgroup is
  // already resident or done, and those never wait for anything younger.
  int* flags;
  // nsplit == 1 / ordered slices: the first slice STORES its product instead of adding it to what dW holds (the caller promises a
  // gradient buffer nobody has written yet: convdr_encoder_backward_fresh) -- no read of dW, and no fill of it before the backward
  int overwrite;
};

template <class T>
static __global__ void __launch_bounds__(T::THREADS, 2) k_gemm_tn(const GemmTnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WavePos<T> w;
  const uint32_t tile_g = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 0; i < TN_MAX_PROBLEMS - 1; ++i)
    if (i + 1 < a.count && (int)tile_g >= a.p[i].tile_end) pi = i + 1;
  // (the selected problem's fields are read through a wave-uniform index: scalar loads from the kernel arguments)
  const GemmTnProblem& q = a.p[pi];
  const int tile = (int)tile_g - (pi ? a.p[pi - 1].tile_end : 0);
  const int tl = tile / q.tilesR, tr = tile - tl * q.tilesR;
  const int r0 = tr * T::TR, l0 = tl * T::TL;
  const int64_t t_begin = (int64_t)blockIdx.y * a.steps_per_split * 64;
  const int64_t left = a.rows - t_begin;
  int nk = (int)((left + 63) / 64);
  nk = nk < a.steps_per_split ? nk : a.steps_per_split;
  GemmAcc<T> acc;
  acc.zero();
  if (nk > 0) {
#ifdef CONVDR_TN_TWO_STAGE
    constexpr bool USE_R3 = false;
#else
    constexpr bool USE_R3 = TnCfg<T>::R3;
#endif
    if constexpr (USE_R3) {
      const int iw = w.wave < T::WAVES / 2 ? w.wave : w.wave - T::WAVES / 2;         // index among the waves of its role
      const TnStageSrc<T::TR, T::WAVES / 2> srcR(q.Rm, q.ldr, q.NR, r0, t_begin, a.rows, iw, w.lane);
      const TnStageSrc<T::TL, T::WAVES / 2> srcL(q.Lm, q.ldl, q.NL, l0, t_begin, a.rows, iw, w.lane);
      gemm_tn_mainloop_r3<T>(srcR, srcL, nk, smem, acc, w);
    } else {
      const int iw = w.wave >= TnIssue<T>::FIRST ? w.wave - TnIssue<T>::FIRST : 0;   // index among the issuing waves
      const TnStageSrc<T::TR, TnIssue<T>::W> srcR(q.Rm, q.ldr, q.NR, r0, t_begin, a.rows, iw, w.lane);
      const TnStageSrc<T::TL, TnIssue<T>::W> srcL(q.Lm, q.ldl, q.NL, l0, t_begin, a.rows, iw, w.lane);
      gemm_tn_mainloop<T>(srcR, srcL, nk, smem, acc, w);
    }
  }
  const bool ordered = a.nsplit > 1 && a.flags != nullptr;
  const bool slab = a.nsplit > 1 && !ordered;
  float* out = q.out + (slab ? (size_t)blockIdx.y * q.NL * q.NR : (size_t)0);
  if (ordered && blockIdx.y > 0) {
    // wait for the previous slice of this tile (cdna guide, Guideline 16: poll relaxed, fence once, then plain loads)
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(&a.flags[tile_g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)blockIdx.y)
        __builtin_amdgcn_s_sleep(4);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const int n = l0 + w.l_index(nt);
    if (n >= q.NL) continue;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int k = r0 + w.r_base(mt, gq);
        if (k < q.NR) {
          const f32x16& v = acc.c[mt][nt];
          float4 o = make_float4(v[4 * gq + 0], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]);
          float4* dst = (float4*)(out + (size_t)n * q.NR + k);
          if (!slab && !(a.overwrite && blockIdx.y == 0)) {
            const float4 c = *dst;
            o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
          }
          *dst = o;
        }
      }
  }
  if (ordered && blockIdx.y + 1 < gridDim.y) {   // publish: this slice's sums are in dW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.flags[tile_g], (int)blockIdx.y + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace convdr
