// Self-attention of the TRAINING step on gfx950: forward that keeps the log-sum-exp, and the backward (dQ / dK, dV).
// Replaces what torch autograd runs for BertSelfAttention (HF transformers 2.3.0 modeling_bert.py, reached from
// /root/reference/model/models.py:141-142 under /root/reference/drivers/run_convdr_train.py:109,178).
//
// Everything reads ONE token-major buffer QKV [rows, 3H] (Q | K | V, the fused projection's tile output) and the
// token-major dO [rows, H]: no transposed copies exist.  Round 1 kept V^T, Q^T, K^T and dO^T in HBM (two transposes per
// layer, a third of the bytes every backward workgroup staged); here an operand that MFMA wants with the contraction
// index on the tile's ROW axis is read from the same LDS tile with the transposing LDS read:
//   tile image  [64 rows][64 d] bf16, 128-byte rows, 16-byte chunk index XOR (row >> 1) & 7 (the engine's swizzle, so the
//               row-fragment ds_read_b128 stay conflict-free; the transposing reads see 2-way conflicts, irrelevant here:
//               these kernels are bound by latency and VALU, not by LDS bandwidth)
//   A^T frag    rows d = 32 dt + (lane & 31), contraction slots j = 0..7 <-> tile rows 16 s + 8 (lane >> 5) + j:
//               two ds_read_b64_tr_b16 (tile rows +0..3 | +4..7); within a 16-lane group lane i hands in the address of
//               tile row (i >> 2), columns 4 (i & 3) .. + 3 of the group's 16-column block (gemm_tn.hpp).
// Tiles arrive by LDS-DMA through a buffer descriptor whose window ends with the last packed row: rows past it read as
// zeros (0 x finite, never 0 x junk, in the MFMAs whose other operand is masked to zero).  Two tile sets in flight.
#pragma once
#include <type_traits>

#include "encoder_kernels.hpp"

namespace convdr {

// ---- staging: 64 rows x 128 bytes from a [nrows, ld] bf16 matrix, window [row 0, nrows) ------------------------------
struct AttnTileSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff;     // this lane's offset inside a round (row-in-round x row pitch + swizzled chunk)
  uint32_t pitch8;   // bytes of 8 rows
  uint32_t rowb;     // bytes of one row
};
__device__ __forceinline__ AttnTileSrc attn_tile_src(const bf16_t* base, int64_t ld, int64_t nrows, int lane) {
  AttnTileSrc s;
  int64_t bytes = nrows * ld * 2;
  bytes = bytes > 0xffffffffll ? 0xffffffffll : bytes;
  const uint64_t b = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
  const int r = lane >> 3;                       // row inside an 8-row wave instruction
  s.rowb = __builtin_amdgcn_readfirstlane((uint32_t)(ld * 2));
  s.pitch8 = 8 * s.rowb;
  // rows of a round are r0 = 8 * (i * 4 + wave) + r: (row >> 1) & 7 = (4 (i * 4 + wave) + (r >> 1)) & 7 -> needs wave, see stage()
  s.voff = (uint32_t)r * s.rowb;
  return s;
}
// first_row: the tile's first row in the matrix, col_bytes: byte offset of the head's 64 columns inside a row
__device__ __forceinline__ void attn_stage_tile(const AttnTileSrc& s, int64_t first_row, uint32_t col_bytes, char* lds, int wave,
                                                int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r0 = (i * 4 + wave) * 8;
    const int row = r0 + (lane >> 3);
    const uint32_t gch = (uint32_t)((lane & 7) ^ ((row >> 1) & 7));
    const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)((first_row + r0) * s.rowb) + col_bytes);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lptr_t)(lds + r0 * 128), 16, s.voff + gch * 16, soff, 0, 0);
  }
}

// ---- transposing fragment reads from a 64 x 128 B tile ---------------------------------------------------------------
// per-lane byte offsets for (dt, rd): d block dt (32 columns), rd = second read of the fragment (tile rows + 4)
struct TrLane {
  uint32_t a[2][2];
};
__device__ __forceinline__ TrLane tr_lane(int lane) {
  TrLane t;
  const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = 8 * (g >> 1) + (i16 >> 2) + 4 * rd;          // inside a 16-row block (blocks keep (row >> 1) & 7)
      const int c = 4 * dt + 2 * (g & 1) + ((i16 & 3) >> 1);      // source chunk of this lane's 4 columns
      t.a[dt][rd] = (uint32_t)(row * 128 + ((c ^ ((row >> 1) & 7)) << 4) + 8 * (i16 & 1));
    }
  return t;
}
union TrFrag {
  bf16x8 v;
  u32x2_t h[2];
};
// fragment of d block dt for the 16 tile rows starting at ROW0 (a multiple of 16); tile = LDS byte address of the tile
template <int ROW0>
__device__ __forceinline__ void tr_frag(uint32_t tile, const TrLane& t, int dt, TrFrag& f) {
  const uint32_t a0 = tile + t.a[dt][0], a1 = tile + t.a[dt][1];
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
               : "=&v"(f.h[0]), "=&v"(f.h[1])
               : "v"(a0), "v"(a1), "n"(ROW0 * 128)
               : "memory");
}
// Wait until all but the newest N LDS reads of this wave have returned, naming the fragments that are now valid: hipcc does
// not count an asm load, so every consumer (and every register copy the allocator wants to make) of those fragments must
// sit behind this statement -- "+v" makes it their definition (cdna guide 5.7 item 1, form ii); the sched_barrier keeps
// register-only MFMAs from being hoisted above the wait (rule 18).
template <int N>
__device__ __forceinline__ void tr_wait2(TrFrag& x, TrFrag& y) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x.h[0]), "+v"(x.h[1]), "+v"(y.h[0]), "+v"(y.h[1]) : "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void tr_wait4(TrFrag& x, TrFrag& y, TrFrag& z, TrFrag& w) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(x.h[0]), "+v"(x.h[1]), "+v"(y.h[0]), "+v"(y.h[1]), "+v"(z.h[0]), "+v"(z.h[1]), "+v"(w.h[0]), "+v"(w.h[1])
               : "n"(N)
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int ATT_TILE = 64 * 128;   // one 64-row x 128-byte LDS tile

// ---------------------------------------------------------------------------------------------------------------------
// forward (training): same lane-local scheme as k_attention_fwd (S^T = K Q^T, O^T = V^T P^T, lane = query), V read
// from its token-major tile through transposing reads, LSE (natural log) saved for the backward.
// ---------------------------------------------------------------------------------------------------------------------
struct AttnTrainArgs {
  const bf16_t* QKV;   // [rows, 3H]
  int64_t rows;
  const int32_t *cu, *lens;
  int H;
  bf16_t* ctx;         // [rows, H]
  DropSite drop;       // attention-probability dropout (thresh 0 = off); heads = gridDim.y
  float* lse;          // [heads, ldt]: log2 of the softmax denominator, log2(sum_k exp(s_k * scale)) (base 2: the backward's exp2 argument)
  int64_t ldt;
  float scale;
  const int32_t* order;   // [B] or null: workgroup z works on sequence order[z] (k_len_order: longest first)
  uint32_t* mbits;        // [heads][ATTM_PIECES][ldt] or null: the dropout keep bits of this forward, for k_attention_bwd_fused
  float* cls32;           // [B, H] or null: the context row of every sequence's FIRST query in fp32 (last layer, CLS pooling): the
                          // backward's D = dO . O of the only queries that carry gradient there -- see AttnBwdArgs::cls32
};

// Dropout keep bits, forward -> backward (round 6).  The mask stays DEFINED by csrc/dropout.hpp (oracle/dropout.py); the
// forward, which has to evaluate the hash anyway, additionally leaves one bit per (query, key) behind, and the one-workgroup
// backward reads bits instead of hashing again: the hash was 36 % of that kernel's instruction stream (227 of 625 VALU
// instructions per 32 queries x 32 keys per wave, ISA count) for 2 instructions per element here.
//   word (head, piece p = 2 * (key >> 6) + ((key >> 3) & 1), packed query row):
//   bit  e = 16 * ((key >> 5) & 1) + 8 * ((key >> 4) & 1) + (key & 7)   -- a forward lane's 32 registers of one 64-key tile
// i.e. exactly the 32 (kt, r) elements a forward lane (query, hi = (key >> 3) & 1) holds, in register order: the forward
// assembles a word with no data movement, the backward lane (= key) extracts ITS bit of the words of the step's queries.
// Only for sequences of at most ATTF_MAX_LEN = 256 tokens (four key tiles: ATTM_PIECES = 8), the one-workgroup backward.
constexpr int ATTM_PIECES = 8;

// Dispatch order of the attention kernels of a training step: sequences in descending length (ties: ascending index).
// A ragged batch (configs[2]: 32 .. 256 tokens) gives these kernels workgroups of 1 .. 4 tiles x 1 .. 8 waves of work, three
// rounds of them per CU; in batch order the last round is whatever the collate function put last, and the kernel ends
// with a few CUs finishing 256-token sequences while the rest idle: heaviest first measured -18 % on the backward and
// -8 % on the forward kernel (timing experiment with a sorted batch, round 4).  One thread per sequence, O(B) each.
static __global__ void __launch_bounds__(256) k_len_order(const int32_t* __restrict__ lens, int B, int32_t* __restrict__ order) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int mine = lens[b];
  int rank = 0;
  for (int j = 0; j < B; ++j) {
    const int o = lens[j];
    rank += (o > mine) | ((o == mine) & (j < b));
  }
  order[rank] = b;
}

template <bool DROP>
static __global__ void __launch_bounds__(256, 3) k_attention_train_fwd(const AttnTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (K tile | V tile)
  const int b = a.order ? a.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
  const int len = a.lens[b];
  const int q0 = blockIdx.x * 128;
  if (q0 >= len) return;
  const int64_t base = a.cu[b];
  const int plen = a.cu[b + 1] - (int)base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int q = q0 + wave * 32 + li;
  const int qc = q < len ? q : len - 1;
  const int H = a.H, H3 = 3 * a.H;
  bf16x8 qf[4];
  {
    const bf16_t* qp = a.QKV + (base + qc) * H3 + h * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
  }
  const float c = a.scale * 1.44269504088896341f;
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  const int krow = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);   // bits 2 <-> 3 (see k_attention_fwd)
  const int ksw = (krow >> 1) & 7;
  const AttnTileSrc src = attn_tile_src(a.QKV, H3, a.rows, lane);
  const TrLane trl = tr_lane(lane);
  const uint32_t s0 = lds_off(smem);
  auto stage = [&](int kv0, int buf) {
    attn_stage_tile(src, base + kv0, (uint32_t)((H + h * 64) * 2), smem + buf * 2 * ATT_TILE, wave, lane);
    attn_stage_tile(src, base + kv0, (uint32_t)((2 * H + h * 64) * 2), smem + buf * 2 * ATT_TILE + ATT_TILE, wave, lane);
  };
  stage(0, 0);
  if (len > 64) stage(64, 1);
  const bool active = q0 + wave * 32 < plen;   // waves whose 32 queries all lie past the sequence only help with the staging
  for (int kv0 = 0, it = 0; kv0 < len; kv0 += 64, ++it) {
    const int buf = it & 1;
    lds_dma_wait_all();
    __syncthreads();
    if (it >= 1 && kv0 + 64 < len) stage(kv0 + 64, buf ^ 1);
    if (!active) continue;
    const char* sK = smem + buf * 2 * ATT_TILE;
    const uint32_t tV = s0 + buf * 2 * ATT_TILE + ATT_TILE;
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
      const char* kp = sK + (kt * 32 + krow) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *(const bf16x8*)(kp + (((2 * s + hi) ^ ksw) * 16));
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
      }
    }
    // V^T fragments, two register sets: the first is issued now and returns under the softmax arithmetic, set s + 1 under
    // the MFMAs of key block s
    TrFrag vf[2][2];
    tr_frag<0>(tV, trl, 0, vf[0][0]);  tr_frag<0>(tV, trl, 1, vf[0][1]);
    float mx = -INFINITY;
    if (kv0 + 64 > len) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= len) st[kt][r] = -INFINITY;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float mnc = mn * c;
    const float alpha = __builtin_amdgcn_exp2f(m * c - mnc);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], c, -mnc));
        st[kt][r] = p;
        ps += p;
      }
    l = l * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    if constexpr (DROP) {   // dropout on the probabilities (the softmax denominator above is that of the undropped row)
      const uint32_t db = drop_att_base(base + qc, gridDim.y, h);
      uint32_t word = 0u;   // keep bits of this lane's 32 elements, register order (see ATTM_PIECES)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int key = kv0 + 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7);
          const uint32_t hh = drop_mix32((db + (uint32_t)(key >> 1)) ^ a.drop.key);
          const bool k0 = (hh & 0xffffu) >= a.drop.thresh, k1 = (hh >> 16) >= a.drop.thresh;
          st[kt][r] *= k0 ? a.drop.scale : 0.f;
          st[kt][r + 1] *= k1 ? a.drop.scale : 0.f;
          word |= (k0 ? 1u << (16 * kt + r) : 0u) | (k1 ? 2u << (16 * kt + r) : 0u);
        }
      if (a.mbits && q < len) a.mbits[((int64_t)(h * ATTM_PIECES + 2 * it + hi)) * a.ldt + base + q] = word;
    }
#define CONVDR_PV_STEP(S4, NEXT_ROW0, WAITN)                                                                    \
    {                                                                                                            \
      constexpr int kt = (S4) >> 1, r0 = ((S4) & 1) * 8, cur = (S4) & 1;                                         \
      if ((S4) < 3) { tr_frag<NEXT_ROW0>(tV, trl, 0, vf[cur ^ 1][0]); tr_frag<NEXT_ROW0>(tV, trl, 1, vf[cur ^ 1][1]); } \
      union { bf16x8 v; uint32_t u[4]; } pb;                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) pb.u[j] = pack_bf16x2(st[kt][r0 + 2 * j], st[kt][r0 + 2 * j + 1]); \
      tr_wait2<WAITN>(vf[cur][0], vf[cur][1]);                                                                   \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                           \
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[cur][dt].v, pb.v, o[dt], 0, 0, 0);                    \
    }
    CONVDR_PV_STEP(0, 16, 4)
    CONVDR_PV_STEP(1, 32, 4)
    CONVDR_PV_STEP(2, 48, 4)
    CONVDR_PV_STEP(3, 48, 0)
#undef CONVDR_PV_STEP
  }
  l += __shfl_xor(l, 32, 64);
  __syncthreads();   // the K / V tiles are dead: their LDS takes the output tiles (whole-line stores, attn_park_store)
  {
    const float inv = q < len ? 1.f / l : 0.f;   // alignment rows [len, plen): zeros
    const int r0 = q - (lane & 31);
    attn_park_store(smem + wave * 4096, o, inv, lane, a.ctx + (base + r0) * H + h * 64, H, plen - r0);
    if (q < plen && hi == 0) a.lse[(int64_t)h * a.ldt + base + q] = q < len ? m * c + __log2f(l) : 0.f;
    if (a.cls32 && q == 0) {   // (accumulator layout: lane (query, hi) holds dims 32 dt + 8 g + 4 hi + 0..3 in o[dt][4 g + 0..3])
      float* c32 = a.cls32 + (int64_t)b * H + h * 64 + 4 * hi;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(c32 + 32 * dt + 8 * g) = make_float4(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const bf16_t* QKV;   // [rows, 3H] token-major (Q | K | V)
  const bf16_t* dO;    // [rows, H]
  int64_t rows;
  const float* LSE;    // [heads, ldt], base-2 (see AttnTrainArgs::lse)
  int heads;
  const float* Dr;     // [heads, ldt]  D[h, t] = sum_d dO[t, d] O[t, d]  (read by the dK / dV kernel)
  const bf16_t* O;     // [rows, H] forward output (ctx).  Non-null: the dQ kernel computes D for its own queries from its dO
  float* Dw;           //   fragment + the matching O fragment and WRITES it to Dw (= Dr) for the dK / dV kernel that follows on
                       //   the stream -- the separate row-dot pass (12 launches per step) is gone
  int64_t ldt;
  const int32_t *cu, *lens;
  int H;
  bf16_t* dQKV;        // [rows, 3H]
  float scale;
  DropSite drop;       // attention-probability dropout of the forward (thresh 0 = off)
  int q_limit;         // dK / dV kernel: > 0 = only the first q_limit queries of a sequence carry gradient (last layer: the
                       // CLS query; a multiple of 64): the query loop stops there
  const int32_t* order;   // [B] or null: workgroup z works on sequence order[z] (k_len_order: longest first)
  const uint32_t* mbits;  // k_attention_bwd_fused<true>: the forward's dropout keep bits (AttnTrainArgs::mbits)
  // Round 6.  dS = P (dP - D) with D = dO . O is a small difference of large terms in a saturated head (P ~ one-hot: D ~ dP of
  // the hot key), and O stored in bf16 puts a 2^-9 relative error into D that lands on the hot key's dS undiminished: on
  // trained-model statistics the last layer's query-projection gradient sat 7.6e-2 (1 - cos) from fp32 for that reason alone
  // (tests/test_train_gpu.py, cfg2_trained).  In that layer only the first query of a sequence carries gradient, so the
  // forward keeps those B context rows in fp32 as well (AttnTrainArgs::cls32) and D of query 0 is taken from them: with the
  // hot key's unnormalised probability exactly 1.0 in the forward, the D that results agrees with the backward's own P dP
  // to fp32 rounding.  Null in every other layer (there the bf16 form is at the bf16-emulating oracle's own distance).
  const float* cls32;     // [B, H] or null
  unsigned long long* trace;   // TRACE builds (tools/dbg/attn_bwd_trace.py): [workgroup][2][16] s_memtime stamps, or null
};
#ifdef CONVDR_ENABLE_TRACE
#define CONVDR_ATTB_STAMP(i)                                                                               \
  if (a.trace && (threadIdx.x == 0 || threadIdx.x == 256))                                                  \
    a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + (threadIdx.x >> 8) * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define CONVDR_ATTB_STAMP(i)
#endif

constexpr int ATTB_DQ_SMEM = 2 * 2 * ATT_TILE;            // two sets of (K tile | V tile)
constexpr int ATTB_DKV_SMEM = 2 * (2 * ATT_TILE + 512);   // two sets of (Q tile | dO tile | 64 LSE | 64 D)

// dQ: workgroup = 128 queries of one (sequence, head); loop over 64-key tiles.  S^T = K Q^T and dP^T = V dO^T have lane =
// query, registers = keys; dQ^T = K^T dS^T with dS^T fed from registers and K^T read transposed from the K tile.
// Two workgroups per CU, not three: at the 168-register cap of three the kernel spilled 73 VGPRs into its key loop
// (scratch traffic per tile; 83 us per launch at the configs[2] size) -- found with the PMC pass of round 3
// (SQ_INSTS_VMEM_WR 34 per wave for a kernel with 5 stores).  At 256 registers it needs 215, no spills: 58 us, the KD step
// 11.63 -> 11.14 ms.  The kernel is latency-bound either way; the third resident workgroup bought less than the
// spills cost.
static __global__ void __launch_bounds__(256, 2) k_attention_bwd_dq(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = a.order ? a.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
  const int len = a.lens[b];
  const int q0 = blockIdx.x * 128;
  if (q0 >= len) return;
  const int64_t base = a.cu[b];
  const int plen = a.cu[b + 1] - (int)base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int q = q0 + wave * 32 + li;
  const int qc = q < len ? q : len - 1;
  const int H = a.H, H3 = 3 * a.H;
  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qp = a.QKV + (base + qc) * H3 + h * 64 + 8 * hi;
    const bf16_t* dp = a.dO + (base + qc) * H + h * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) { qf[s] = *(const bf16x8*)(qp + 16 * s); dof[s] = *(const bf16x8*)(dp + 16 * s); }
  }
  const float c = a.scale * 1.44269504088896341f;
  const float lse2 = a.LSE[(int64_t)h * a.ldt + base + qc];
  float Di;
  if (a.O) {   // D of this lane's query: its half of the 64 head dimensions here, the other half in lane ^ 32
    const bf16_t* op = a.O + (base + qc) * H + h * 64 + 8 * hi;
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      union { bf16x8 v; uint32_t u[4]; } x, y;
      x.v = dof[s];
      y.v = *(const bf16x8*)(op + 16 * s);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc += __uint_as_float(x.u[j] << 16) * __uint_as_float(y.u[j] << 16) +
               __uint_as_float(x.u[j] & 0xffff0000u) * __uint_as_float(y.u[j] & 0xffff0000u);
    }
    if (a.cls32 && q == 0) {   // query 0 from the fp32 context row (AttnBwdArgs::cls32)
      const float* o32 = a.cls32 + (int64_t)b * H + h * 64 + 8 * hi;
      acc = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        union { bf16x8 v; uint32_t u[4]; } x;
        x.v = dof[s];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc += __uint_as_float(x.u[j] << 16) * o32[16 * s + 2 * j] + __uint_as_float(x.u[j] & 0xffff0000u) * o32[16 * s + 2 * j + 1];
      }
    }
    acc += __shfl_xor(acc, 32, 64);
    Di = acc;
    if (hi == 0 && q < len) a.Dw[(int64_t)h * a.ldt + base + q] = Di;
  } else {
    Di = a.Dr[(int64_t)h * a.ldt + base + qc];
  }
  f32x16 dq[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
  const int krow = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);
  const int ksw = (krow >> 1) & 7;
  const AttnTileSrc src = attn_tile_src(a.QKV, H3, a.rows, lane);
  const TrLane trl = tr_lane(lane);
  const uint32_t s0 = lds_off(smem);
  auto stage = [&](int kv0, int buf) {
    attn_stage_tile(src, base + kv0, (uint32_t)((H + h * 64) * 2), smem + buf * 2 * ATT_TILE, wave, lane);
    attn_stage_tile(src, base + kv0, (uint32_t)((2 * H + h * 64) * 2), smem + buf * 2 * ATT_TILE + ATT_TILE, wave, lane);
  };
  stage(0, 0);
  if (len > 64) stage(64, 1);
  const bool active = q0 + wave * 32 < len;
  for (int kv0 = 0, it = 0; kv0 < len; kv0 += 64, ++it) {
    const int buf = it & 1;
    lds_dma_wait_all();
    __syncthreads();
    if (it >= 1 && kv0 + 64 < len) stage(kv0 + 64, buf ^ 1);
    if (!active) continue;
    const char* sK = smem + buf * 2 * ATT_TILE;
    const char* sV = sK + ATT_TILE;
    const uint32_t tK = s0 + buf * 2 * ATT_TILE;
    f32x16 st[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[kt][r] = 0.f; dp[kt][r] = 0.f; }
      const char* kp = sK + (kt * 32 + krow) * 128;
      const char* vp = sV + (kt * 32 + krow) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ch = ((2 * s + hi) ^ ksw) * 16;
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + ch), qf[s], st[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(vp + ch), dof[s], dp[kt], 0, 0, 0);
      }
    }
    TrFrag kt_f[2][2];   // K^T fragments, two register sets (first one returns under the elementwise pass)
    tr_frag<0>(tK, trl, 0, kt_f[0][0]);  tr_frag<0>(tK, trl, 1, kt_f[0][1]);
    const bool ragged = kv0 + 64 > len;   // workgroup-uniform
    if (a.drop.thresh) {   // dP = dP_dropped * mask: one hash per pair of consecutive keys (registers r, r + 1)
      const uint32_t db = drop_att_base(base + qc, gridDim.y, h);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int key = kv0 + 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7);
          float m0, m1;
          drop_pair(a.drop, db + (uint32_t)(key >> 1), m0, m1);
          dp[kt][r] *= m0; dp[kt][r + 1] *= m1;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], c, -lse2));
        float ds = p * (dp[kt][r] - Di) * a.scale;
        if (ragged) {
          const int key = kv0 + 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7);
          ds = key < len ? ds : 0.f;   // select, never 0 * junk
        }
        st[kt][r] = ds;
      }
#define CONVDR_DQ_STEP(S4, NEXT_ROW0, WAITN)                                                                    \
    {                                                                                                            \
      constexpr int kt = (S4) >> 1, r0 = ((S4) & 1) * 8, cur = (S4) & 1;                                         \
      if ((S4) < 3) { tr_frag<NEXT_ROW0>(tK, trl, 0, kt_f[cur ^ 1][0]); tr_frag<NEXT_ROW0>(tK, trl, 1, kt_f[cur ^ 1][1]); } \
      union { bf16x8 v; uint32_t u[4]; } pb;                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) pb.u[j] = pack_bf16x2(st[kt][r0 + 2 * j], st[kt][r0 + 2 * j + 1]); \
      tr_wait2<WAITN>(kt_f[cur][0], kt_f[cur][1]);                                                               \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt)                                                           \
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_f[cur][dt].v, pb.v, dq[dt], 0, 0, 0);                \
    }
    CONVDR_DQ_STEP(0, 16, 4)
    CONVDR_DQ_STEP(1, 32, 4)
    CONVDR_DQ_STEP(2, 48, 4)
    CONVDR_DQ_STEP(3, 48, 0)
#undef CONVDR_DQ_STEP
  }
  __syncthreads();   // the K / V tiles are dead
  {
    const float keep = q < len ? 1.f : 0.f;
    const int r0 = q - (lane & 31);
    attn_park_store(smem + wave * 4096, dq, keep, lane, a.dQKV + (base + r0) * H3 + h * 64, H3, plen - r0);
  }
}

// dK, dV: workgroup = 128 keys of one (sequence, head), lane = key; loop over 64-query tiles.
//   S = Q K^T and dP = dO V^T with A = query rows (registers), B = this lane's key;
//   dV^T = dO^T P and dK^T = Q^T dS with P / dS fed from registers, dO^T / Q^T read transposed from the dO / Q tiles.
static __global__ void __launch_bounds__(256, 2) k_attention_bwd_dkv(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SET = 2 * ATT_TILE + 512;
  const int b = a.order ? a.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
  const int len = a.lens[b];
  const int k0 = blockIdx.x * 128;
  if (k0 >= len) return;
  const int64_t base = a.cu[b];
  const int plen = a.cu[b + 1] - (int)base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int key = k0 + wave * 32 + li;
  const int kc = key < len ? key : len - 1;
  const int H = a.H, H3 = 3 * a.H;
  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = a.QKV + (base + kc) * H3 + H + h * 64 + 8 * hi;
    const bf16_t* vp = a.QKV + (base + kc) * H3 + 2 * H + h * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = *(const bf16x8*)(kp + 16 * s); vf[s] = *(const bf16x8*)(vp + 16 * s); }
  }
  const float c = a.scale * 1.44269504088896341f;
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
  const int qrow = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);
  const int qsw = (qrow >> 1) & 7;
  const AttnTileSrc srcQ = attn_tile_src(a.QKV, H3, a.rows, lane);
  const AttnTileSrc srcO = attn_tile_src(a.dO, H, a.rows, lane);
  const TrLane trl = tr_lane(lane);
  const uint32_t s0 = lds_off(smem);
  // LSE / D of the tile's 64 queries ride the same DMA queue (4 bytes per lane; an ordinary load here would make hipcc
  // drain the tile DMA it follows).  Columns past the last row lie inside the [heads, ldt] slack; past the array: zeros.
  const AttnTileSrc srcL = attn_tile_src((const bf16_t*)a.LSE, 2 * a.ldt, a.heads, lane);
  const AttnTileSrc srcD = attn_tile_src((const bf16_t*)a.Dr, 2 * a.ldt, a.heads, lane);
  auto stage = [&](int q0, int buf) {
    char* set = smem + buf * SET;
    attn_stage_tile(srcQ, base + q0, (uint32_t)(h * 64 * 2), set, wave, lane);
    attn_stage_tile(srcO, base + q0, (uint32_t)(h * 64 * 2), set + ATT_TILE, wave, lane);
    if (wave < 2) {   // wave-uniform
      const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(((int64_t)h * a.ldt + base + q0) * 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? srcL.rsrc : srcD.rsrc, (lptr_t)(set + 2 * ATT_TILE + wave * 256), 4,
                                               (uint32_t)lane * 4, soff, 0, 0);
    }
  };
  const int qlen = (a.q_limit > 0 && a.q_limit < len) ? a.q_limit : len;   // queries that carry gradient
  stage(0, 0);
  if (qlen > 64) stage(64, 1);
  const bool active = k0 + wave * 32 < len;
  for (int q0 = 0, it = 0; q0 < qlen; q0 += 64, ++it) {
    const int buf = it & 1;
    lds_dma_wait_all();
    __syncthreads();
    if (it >= 1 && q0 + 64 < qlen) stage(q0 + 64, buf ^ 1);
    if (!active) continue;
    const char* sQ = smem + buf * SET;
    const char* sdO = sQ + ATT_TILE;
    const float* sLse = (const float*)(sQ + 2 * ATT_TILE);
    const float* sD = sLse + 64;
    const uint32_t tQ = s0 + buf * SET, tO = tQ + ATT_TILE;
    const bool ragged = q0 + 64 > len;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {  // 32 queries at a time (keeps the register footprint at one S / dP tile)
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const char* qp = sQ + (qt * 32 + qrow) * 128;
      const char* op = sdO + (qt * 32 + qrow) * 128;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int ch = ((2 * s4 + hi) ^ qsw) * 16;
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(qp + ch), kf[s4], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(op + ch), vf[s4], dp, 0, 0, 0);
      }
      // transposed fragments [set][dt] of the 16-query halves, dO^T and Q^T: half 0 is issued now and returns under the
      // elementwise pass, half 1 under the MFMAs of half 0
      TrFrag of[2][2], qf[2][2];
      if (qt == 0) {
        tr_frag<0>(tO, trl, 0, of[0][0]); tr_frag<0>(tO, trl, 1, of[0][1]);
        tr_frag<0>(tQ, trl, 0, qf[0][0]); tr_frag<0>(tQ, trl, 1, qf[0][1]);
      } else {
        tr_frag<32>(tO, trl, 0, of[0][0]); tr_frag<32>(tO, trl, 1, of[0][1]);
        tr_frag<32>(tQ, trl, 0, qf[0][0]); tr_frag<32>(tQ, trl, 1, qf[0][1]);
      }
      // register r <-> query q0 + 32 qt + 16 (r >> 3) + 8 hi + (r & 7)
      // Dropout masks (this lane's key, the register's query): the mask hash covers a PAIR of keys, and the two lanes of
      // a pair would each compute it for every query.  Instead the even-key lane hashes query r, the odd-key lane query
      // r + 1, and they swap (one DPP move): one hash per two elements, as in the kernels whose lanes are queries -- the
      // hash was 2/3 of this kernel's VALU work.  (Same mask: drop_pair of csrc/dropout.hpp.)
      float dmask[16];
      if (a.drop.thresh) {
        // parity of the UNCLAMPED key (= the lane parity: k0 + 32 wave is even).  The clamped key would give the partner
        // lane of an odd-length sequence's last key (key == len, clamped to the even len - 1) the wrong role in the swap.
        const int odd = key & 1;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int qi_mine = 32 * qt + 16 * (r >> 3) + 8 * hi + (r & 7) + odd;
          const uint32_t h_mine = drop_mix32((drop_att_base(base + q0 + qi_mine, gridDim.y, h) + (uint32_t)(kc >> 1)) ^ a.drop.key);
          const uint32_t h_other = (uint32_t)__builtin_amdgcn_mov_dpp((int)h_mine, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
          const uint32_t h0 = odd ? h_other : h_mine, h1 = odd ? h_mine : h_other;   // hashes of queries r, r + 1
          const uint32_t t0 = odd ? (h0 >> 16) : (h0 & 0xffffu), t1 = odd ? (h1 >> 16) : (h1 & 0xffffu);
          dmask[r] = t0 >= a.drop.thresh ? a.drop.scale : 0.f;
          dmask[r + 1] = t1 >= a.drop.thresh ? a.drop.scale : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = 32 * qt + 16 * (r >> 3) + 8 * hi + (r & 7);
        float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -sLse[qi]));
        float dpr = dp[r], pd = p;
        if (a.drop.thresh) {   // P_dropped = P * m feeds dV, dP = dP_dropped * m feeds dS
          const float m = dmask[r];
          dpr *= m;
          pd *= m;
        }
        float ds = p * (dpr - sD[qi]) * a.scale;
        if (ragged) {
          const bool ok = q0 + qi < len;
          pd = ok ? pd : 0.f;
          ds = ok ? ds : 0.f;
        }
        s[r] = pd;
        dp[r] = ds;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // 16 queries per MFMA k-step
        const int r0 = half * 8;
        if (half == 0) {
          if (qt == 0) {
            tr_frag<16>(tO, trl, 0, of[1][0]); tr_frag<16>(tO, trl, 1, of[1][1]);
            tr_frag<16>(tQ, trl, 0, qf[1][0]); tr_frag<16>(tQ, trl, 1, qf[1][1]);
          } else {
            tr_frag<48>(tO, trl, 0, of[1][0]); tr_frag<48>(tO, trl, 1, of[1][1]);
            tr_frag<48>(tQ, trl, 0, qf[1][0]); tr_frag<48>(tQ, trl, 1, qf[1][1]);
          }
        }
        union { bf16x8 v; uint32_t u[4]; } pb, sb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pb.u[j] = pack_bf16x2(s[r0 + 2 * j], s[r0 + 2 * j + 1]);
          sb.u[j] = pack_bf16x2(dp[r0 + 2 * j], dp[r0 + 2 * j + 1]);
        }
        if (half == 0) tr_wait4<8>(of[0][0], of[0][1], qf[0][0], qf[0][1]);
        else tr_wait4<0>(of[1][0], of[1][1], qf[1][0], qf[1][1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[half][dt].v, pb.v, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[half][dt].v, sb.v, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();   // the Q / dO tiles are dead
  {
    const float keep = key < len ? 1.f : 0.f;
    const int r0 = key - (lane & 31);
    attn_park_store(smem + wave * 4096, dk, keep, lane, a.dQKV + (base + r0) * H3 + H + h * 64, H3, plen - r0);
    attn_park_store(smem + wave * 4096, dv, keep, lane, a.dQKV + (base + r0) * H3 + 2 * H + h * 64, H3, plen - r0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ, dK and dV of one (sequence, head) in ONE workgroup, for sequences of at most 256 tokens (round 4).
// The two kernels above each recompute S, P, the dropout mask and dS for every (query, key) pair -- exp2 + hash + the
// dS arithmetic are what these latency- and VALU-bound kernels spend their time on -- and each pays its own prologue
// (operand fragments, first tiles) and launch.  Here eight waves own 32 keys each (lane = key, exactly the dK / dV kernel's
// scheme and arithmetic), loop over 64-query tiles, and additionally park dS (bf16, [key][query], the engine's tile
// image) in LDS; one step later -- from the other of two dS sets, beside the next step of the waves that own keys -- waves
// 4 .. 7 contract it (a 32-query x 32-dimension block each) with K^T read from the sequence's K tiles:
//   dQ^T[d, q] = sum_key K^T[d, key] dS^T[key, q]      (both operands through ds_read_b64_tr_b16)
// so dQ needs neither a second pass over the scores nor a cross-workgroup reduction (one workgroup sees every key), and a
// step has one barrier.  D[q] = dO[q] . O[q] is computed in the prologue for the whole sequence (two threads per query).
// Measured (configs[2], 11 layers): dQ kernel + dK / dV kernel 98 us per layer; this kernel 96.6 us in its first form (dQ
// phase behind a second barrier, batch order), 77.7 us with the late dQ phase on two waves and the longest sequences
// dispatched first, ~72 us with it on four.
// The attention backward is ON the step's critical path: a timing-only cut of its work by 0.49 ms shortened the configs[2]
// step by 0.8 ms (10.30 -> 9.49).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ATTF_MAX_LEN = 256;
constexpr int ATTF_SET = 2 * ATT_TILE + 512;          // Q tile | dO tile | 64 LSE (+ pad)
constexpr int ATTF_K = 2 * ATTF_SET;                  // four K tiles (the sequence's keys)
constexpr int ATTF_DS = ATTF_K + 4 * ATT_TILE;        // two sets of four dS^T tiles: rows = keys, columns = the 64 queries of a step
constexpr int ATTF_D = ATTF_DS + 8 * ATT_TILE;        // D of the sequence's queries (256 floats)
constexpr int ATTF_PARK = ATTF_D + 1024;              // park regions of the four dQ waves (2 KB each)
constexpr int ATTF_M = ATTF_PARK + 2 * 4096;          // two sets of ATTM_PIECES x 64 dropout keep words (one per query of a step)
constexpr int ATTF_M_SET = ATTM_PIECES * 256;
constexpr int ATTB_FUSED_SMEM = ATTF_M + 2 * ATTF_M_SET;

// one 8-row round of a 64-row tile per wave (eight waves: the whole tile)
__device__ __forceinline__ void attn_stage_rows8(const AttnTileSrc& s, int64_t first_row, uint32_t col_bytes, char* lds, int wave,
                                                 int lane) {
  const int r0 = wave * 8;
  const int row = r0 + (lane >> 3);
  const uint32_t gch = (uint32_t)((lane & 7) ^ ((row >> 1) & 7));
  const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)((first_row + r0) * s.rowb) + col_bytes);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lptr_t)(lds + r0 * 128), 16, s.voff + gch * 16, soff, 0, 0);
}

// One 32-row x 32-column bf16 block (half the head dimensions of 32 queries) from an accumulator tile: parked in the wave's
// 2 KB ([32 rows][64 bytes]), written out 16 bytes per lane, four lanes per row.  Same register layout as attn_park_store.
__device__ __forceinline__ void attn_park_store_half(char* so, const f32x16& o, float scale, int lane, bf16_t* dst, int64_t pitch,
                                                     int nvalid) {
  const int li = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint2 ov;
    ov.x = pack_bf16x2(o[4 * g + 0] * scale, o[4 * g + 1] * scale);
    ov.y = pack_bf16x2(o[4 * g + 2] * scale, o[4 * g + 3] * scale);
    *(uint2*)(so + li * 64 + ((g ^ (li & 3)) << 4) + hi * 8) = ov;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int c4 = lane & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (lane >> 2) + 16 * i;
    const uint4 v = *(const uint4*)(so + row * 64 + ((c4 ^ (row & 3)) << 4));
    if (row < nvalid) *(uint4*)(dst + row * pitch + c4 * 8) = v;
  }
  __builtin_amdgcn_wave_barrier();
}

// Round 6 -- the instruction diet.  The kernel is bound by its waves' own instruction streams (MfmaUtil 7.8 %, 25 % of its HBM
// floor): per 32 queries x 32 keys a wave issued 625 instructions for 16 MFMAs.  (i) DROP is a template parameter and the
// keep bits come from the forward (ATTM_PIECES): 2 instructions per element instead of 14; (ii) the ragged last query step
// is its own instantiation of the step body instead of two selects + a compare per element in every step; (iii) the softmax
// scale is applied once to dK / dQ at the end (0.125: a power of two, the bf16 rounding of dS is unchanged bit for bit).
template <bool DROP>
static __global__ void __launch_bounds__(512, 1) k_attention_bwd_fused(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CONVDR_ATTB_STAMP(0)
  const int b = a.order ? a.order[blockIdx.y] : (int)blockIdx.y, h = blockIdx.x;
  const int len = a.lens[b];
#ifdef CONVDR_ENABLE_TRACE
  if (a.trace && threadIdx.x == 0) {
    a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + 14] = (unsigned long long)len;
    a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + 15] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
  }
#endif
  const int64_t base = a.cu[b];
  const int plen = a.cu[b + 1] - (int)base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int key = wave * 32 + li;
  const int kc = key < len ? key : len - 1;
  const bool key_ok = key < len;
  const int H = a.H, H3 = 3 * a.H;
  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = a.QKV + (base + kc) * H3 + H + h * 64 + 8 * hi;
    const bf16_t* vp = a.QKV + (base + kc) * H3 + 2 * H + h * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = *(const bf16x8*)(kp + 16 * s); vf[s] = *(const bf16x8*)(vp + 16 * s); }
  }
  // D: thread (query = t >> 1, half of the head dimensions = t & 1); the loads are issued here, ahead of the tile DMA, and
  // consumed behind it
  bf16x8 dox[4], ox[4];
  const int qd = threadIdx.x >> 1, dhalf = threadIdx.x & 1;
  {
    const int qdc = qd < len ? qd : len - 1;
    const bf16_t* dp_ = a.dO + (base + qdc) * H + h * 64 + 32 * dhalf;
    const bf16_t* op_ = a.O + (base + qdc) * H + h * 64 + 32 * dhalf;
#pragma unroll
    for (int s = 0; s < 4; ++s) { dox[s] = *(const bf16x8*)(dp_ + 8 * s); ox[s] = *(const bf16x8*)(op_ + 8 * s); }
  }
  const float c = a.scale * 1.44269504088896341f;
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
  const int qrow = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);
  const int qsw = (qrow >> 1) & 7;
  const AttnTileSrc srcQ = attn_tile_src(a.QKV, H3, a.rows, lane);
  const AttnTileSrc srcO = attn_tile_src(a.dO, H, a.rows, lane);
  const AttnTileSrc srcL = attn_tile_src((const bf16_t*)a.LSE, 2 * a.ldt, a.heads, lane);
  const AttnTileSrc srcM = attn_tile_src((const bf16_t*)a.mbits, 2 * a.ldt, DROP ? a.heads * ATTM_PIECES : 0, lane);
  const TrLane trl = tr_lane(lane);
  const uint32_t s0 = lds_off(smem);
  auto stage = [&](int q0, int buf) {
    char* set = smem + buf * ATTF_SET;
    attn_stage_rows8(srcQ, base + q0, (uint32_t)(h * 64 * 2), set, wave, lane);
    attn_stage_rows8(srcO, base + q0, (uint32_t)(h * 64 * 2), set + ATT_TILE, wave, lane);
    if (wave == 0) {
      const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(((int64_t)h * a.ldt + base + q0) * 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcL.rsrc, (lptr_t)(set + 2 * ATT_TILE), 4, (uint32_t)lane * 4, soff, 0, 0);
    }
    if constexpr (DROP) {   // wave w: the keep words of piece w (key tile w >> 1, key half-groups w & 1) for the step's 64 queries
      if ((wave >> 1) * 64 < len) {
        const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(((int64_t)(h * ATTM_PIECES + wave) * a.ldt + base + q0) * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcM.rsrc, (lptr_t)(smem + ATTF_M + buf * ATTF_M_SET + wave * 256), 4,
                                                 (uint32_t)lane * 4, soff, 0, 0);
      }
    }
  };
  const int qlen = (a.q_limit > 0 && a.q_limit < len) ? a.q_limit : len;   // queries that carry gradient (last layer: the CLS tile)
  stage(0, 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t * 64 < len) attn_stage_rows8(srcQ, base + t * 64, (uint32_t)((H + h * 64) * 2), smem + ATTF_K + t * ATT_TILE, wave, lane);
  if (qlen > 64) stage(64, 1);
  const bool active = wave * 32 < len;
  float* sDall = (float*)(smem + ATTF_D);
  {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      union { bf16x8 v; uint32_t u[4]; } x, y;
      x.v = dox[s];
      y.v = ox[s];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc += __uint_as_float(x.u[j] << 16) * __uint_as_float(y.u[j] << 16) +
               __uint_as_float(x.u[j] & 0xffff0000u) * __uint_as_float(y.u[j] & 0xffff0000u);
    }
    if (a.cls32 && qd == 0) {   // query 0 from the fp32 context row (AttnBwdArgs::cls32)
      const float* o32 = a.cls32 + (int64_t)b * H + h * 64 + 32 * dhalf;
      acc = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        union { bf16x8 v; uint32_t u[4]; } x;
        x.v = dox[s];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc += __uint_as_float(x.u[j] << 16) * o32[8 * s + 2 * j] + __uint_as_float(x.u[j] & 0xffff0000u) * o32[8 * s + 2 * j + 1];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    if (dhalf == 0) sDall[qd] = acc;
  }
  // this wave's 32 rows of the dS^T tiles: written every step by a wave that owns keys, zero for good otherwise (the dQ
  // contraction runs over whole 64-key tiles)
  const int ds_row = (wave & 1) * 32 + li;
  char* ds_rowp = smem + ATTF_DS + (wave >> 1) * ATT_TILE + ds_row * 128;
  const int ds_sw = (ds_row >> 1) & 7;
  if (!active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(ds_rowp + (hi * 4 + i) * 16) = make_uint4(0u, 0u, 0u, 0u);
      *(uint4*)(ds_rowp + 4 * ATT_TILE + (hi * 4 + i) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  // The dQ contraction of step i runs one step late, from the other dS^T set, on waves 4 .. 7 -- one 32-query x 32-dimension
  // block each -- beside step i + 1 of the waves that own keys (for sequences of at most 128 tokens waves 4 .. 7 own none;
  // past 192 tokens every wave owns keys and the contraction is spread over four of them): one barrier per step, not two.
  const bool dq_wave = wave >= 4;
  const int qb = (wave - 4) & 1, dtq = (wave - 4) >> 1;
  TrLane trq = trl, trk = trl;   // fragment addresses selected here: a run-time table index would put the table in scratch
  trq.a[0][0] = qb ? trl.a[1][0] : trl.a[0][0];
  trq.a[0][1] = qb ? trl.a[1][1] : trl.a[0][1];
  trk.a[0][0] = dtq ? trl.a[1][0] : trl.a[0][0];
  trk.a[0][1] = dtq ? trl.a[1][1] : trl.a[0][1];
  auto dq_phase = [&](int qs, int set) {
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    for (int t = 0; t * 64 < len; ++t) {
      const uint32_t tK = s0 + ATTF_K + t * ATT_TILE, tS = s0 + ATTF_DS + (set * 4 + t) * ATT_TILE;
      TrFrag ka[4], sf[4];
      tr_frag<0>(tK, trk, 0, ka[0]);  tr_frag<0>(tS, trq, 0, sf[0]);
      tr_frag<16>(tK, trk, 0, ka[1]); tr_frag<16>(tS, trq, 0, sf[1]);
      tr_frag<32>(tK, trk, 0, ka[2]); tr_frag<32>(tS, trq, 0, sf[2]);
      tr_frag<48>(tK, trk, 0, ka[3]); tr_frag<48>(tS, trq, 0, sf[3]);
      tr_wait4<0>(ka[0], ka[1], ka[2], ka[3]);
      tr_wait4<0>(sf[0], sf[1], sf[2], sf[3]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s4].v, sf[s4].v, dq, 0, 0, 0);
    }
    const int r0 = qs + 32 * qb;
    const float keep = r0 + li < len ? a.scale : 0.f;   // (dS was parked without the softmax scale)
    attn_park_store_half(smem + ATTF_PARK + (wave - 4) * 2048, dq, keep, lane, a.dQKV + (base + r0) * H3 + h * 64 + 32 * dtq, H3,
                         plen - r0);
  };
  // this lane's dropout keep bit inside a word, and the piece its words come from (ATTM_PIECES)
  const int e_lane = 16 * (wave & 1) + 8 * ((li >> 4) & 1) + (li & 7);
  const int p_lane = 2 * (wave >> 1) + ((li >> 3) & 1);
  const uint32_t drop_scale_bits = __float_as_uint(a.drop.scale);
  // one query step of a wave that owns keys
  auto step_body = [&](const int q0, const int buf, char* const ds_cur) __attribute__((always_inline)) {
    const char* sQ = smem + buf * ATTF_SET;
    const char* sdO = sQ + ATT_TILE;
    const float* sLse = (const float*)(sQ + 2 * ATT_TILE);
    const float* sD = sDall + q0;
    const uint32_t* sM = (const uint32_t*)(smem + ATTF_M + buf * ATTF_M_SET + p_lane * 256);
    const uint32_t tQ = s0 + buf * ATTF_SET, tO = tQ + ATT_TILE;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const char* qp = sQ + (qt * 32 + qrow) * 128;
      const char* op = sdO + (qt * 32 + qrow) * 128;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int ch = ((2 * s4 + hi) ^ qsw) * 16;
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(qp + ch), kf[s4], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(op + ch), vf[s4], dp, 0, 0, 0);
      }
      TrFrag of[2][2], qf[2][2];
      if (qt == 0) {
        tr_frag<0>(tO, trl, 0, of[0][0]); tr_frag<0>(tO, trl, 1, of[0][1]);
        tr_frag<0>(tQ, trl, 0, qf[0][0]); tr_frag<0>(tQ, trl, 1, qf[0][1]);
      } else {
        tr_frag<32>(tO, trl, 0, of[0][0]); tr_frag<32>(tO, trl, 1, of[0][1]);
        tr_frag<32>(tQ, trl, 0, qf[0][0]); tr_frag<32>(tQ, trl, 1, qf[0][1]);
      }
      // register r <-> query q0 + 32 qt + 16 (r >> 3) + 8 hi + (r & 7)
      uint32_t kw[16];
      if constexpr (DROP) {   // the keep words of those 16 queries for this lane's piece: 2 x 8 consecutive words
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint4 w0 = *(const uint4*)(sM + 32 * qt + 16 * g + 8 * hi), w1 = *(const uint4*)(sM + 32 * qt + 16 * g + 8 * hi + 4);
          kw[8 * g + 0] = w0.x; kw[8 * g + 1] = w0.y; kw[8 * g + 2] = w0.z; kw[8 * g + 3] = w0.w;
          kw[8 * g + 4] = w1.x; kw[8 * g + 5] = w1.y; kw[8 * g + 6] = w1.z; kw[8 * g + 7] = w1.w;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = 32 * qt + 16 * (r >> 3) + 8 * hi + (r & 7);
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -sLse[qi]));
        float pd = p, dpr = dp[r];
        if constexpr (DROP) {   // P_dropped = P * m feeds dV, dP = dP_dropped * m feeds dS; m = keep ? 1 / (1 - p) : 0
          const float m = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)kw[r], (uint32_t)e_lane, 1u) & drop_scale_bits);
          dpr *= m;
          pd *= m;
        }
        s[r] = pd;
        dp[r] = p * (dpr - sD[qi]);   // dS (x softmax scale: applied to dK / dQ at the end)
      }
      // The step that reaches past the sequence's last query (only the last one, and only when the length is not a multiple
      // of 64) zeroes those queries' P and dS -- selects, never 0 * junk -- behind a REAL wave-uniform branch: as a condition
      // inside the loop above hipcc if-converted it into two selects + a compare per element of every step.
      if (__builtin_expect(q0 + 64 > len, 0)) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = q0 + 32 * qt + 16 * (r >> 3) + 8 * hi + (r & 7) < len;
          s[r] = ok ? s[r] : 0.f;
          dp[r] = ok ? dp[r] : 0.f;
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int r0 = half * 8;
        if (half == 0) {
          if (qt == 0) {
            tr_frag<16>(tO, trl, 0, of[1][0]); tr_frag<16>(tO, trl, 1, of[1][1]);
            tr_frag<16>(tQ, trl, 0, qf[1][0]); tr_frag<16>(tQ, trl, 1, qf[1][1]);
          } else {
            tr_frag<48>(tO, trl, 0, of[1][0]); tr_frag<48>(tO, trl, 1, of[1][1]);
            tr_frag<48>(tQ, trl, 0, qf[1][0]); tr_frag<48>(tQ, trl, 1, qf[1][1]);
          }
        }
        union { bf16x8 v; uint32_t u[4]; } pb, sb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pb.u[j] = pack_bf16x2(s[r0 + 2 * j], s[r0 + 2 * j + 1]);
          sb.u[j] = pack_bf16x2(dp[r0 + 2 * j], dp[r0 + 2 * j + 1]);
        }
        if (half == 0) tr_wait4<8>(of[0][0], of[0][1], qf[0][0], qf[0][1]);
        else tr_wait4<0>(of[1][0], of[1][1], qf[1][0], qf[1][1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[half][dt].v, pb.v, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[half][dt].v, sb.v, dk[dt], 0, 0, 0);
        }
        // dS of this lane's key for queries 32 qt + 16 half + 8 hi + 0..7: chunk 4 qt + 2 half + hi of the key's row
        // (a lane whose key lies past the sequence holds a copy of the last key's values: zero)
        const int chunk = 4 * qt + 2 * half + hi;
        const uint4 z = key_ok ? make_uint4(sb.u[0], sb.u[1], sb.u[2], sb.u[3]) : make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(ds_cur + ((chunk ^ ds_sw) << 4)) = z;
      }
    }
  };
  int it = 0;
  CONVDR_ATTB_STAMP(1)
  for (int q0 = 0; q0 < qlen; q0 += 64, ++it) {
    const int buf = it & 1;
    lds_dma_wait_all();
    __syncthreads();   // tile `it` has landed; dS^T of step it - 1 is complete; the set of step it - 2 is free
    CONVDR_ATTB_STAMP(2 + 2 * it)
    if (it >= 1 && q0 + 64 < qlen) stage(q0 + 64, buf ^ 1);
    if (dq_wave && it >= 1) dq_phase(q0 - 64, buf ^ 1);
    CONVDR_ATTB_STAMP(3 + 2 * it)
    char* const ds_cur = ds_rowp + buf * 4 * ATT_TILE;
    if (active) step_body(q0, buf, ds_cur);
  }
  CONVDR_ATTB_STAMP(10)
  __syncthreads();   // the last step's dS^T is complete; the Q / dO tiles are dead
  CONVDR_ATTB_STAMP(11)
  // dK / dV leave FIRST (their park is the dead Q / dO region; the last dQ contraction reads the K tiles and the dS set, its own
  // park is elsewhere): the stores drain under the dQ waves' last contraction instead of behind it (trace: 2.7 k + 2.6 k cycles in
  // a row on waves 4..7 of a 256-token workgroup)
  if (active) {
    const float keep = key_ok ? 1.f : 0.f;
    const int r0 = wave * 32;
    attn_park_store(smem + wave * 4096, dk, keep * a.scale, lane, a.dQKV + (base + r0) * H3 + H + h * 64, H3, plen - r0);
    attn_park_store(smem + wave * 4096, dv, keep, lane, a.dQKV + (base + r0) * H3 + 2 * H + h * 64, H3, plen - r0);
  }
  CONVDR_ATTB_STAMP(12)
  if (dq_wave) dq_phase((it - 1) * 64, (it - 1) & 1);
  CONVDR_ATTB_STAMP(13)
}

}  // namespace convdr
