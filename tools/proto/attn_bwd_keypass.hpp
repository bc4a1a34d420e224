// Prototype (NOT product, not compiled by the Makefile): the one-workgroup attention backward as a KEY-PASS kernel -- 256 threads per
// (sequence, head), 128 keys per pass, 76 KB of LDS so that TWO workgroups are resident per CU.  Built into the product for one
// A/B in round 6 (it slots in after k_attention_bwd_fused in csrc/attention_train.hpp, needs `float* dq32` ([rows, H] fp32 scratch) in
// AttnBwdArgs and a launch of dim3(heads, B) x 256 threads with ATTP_SMEM bytes of dynamic LDS), bit-identical to the dQ + dK/dV
// kernel pair on every ragged shape with and without dropout (tools/dbg/attn_fused_vs_split.py, FUSED=2) -- and SLOWER than the
// 8-wave kernel: attention_bwd 0.75-0.77 vs 0.68 ms per configs[2] step, step 9.03 vs 8.92 ms (profiles/r06_ab_attention_experiments.txt).
// The second resident workgroup does hide the other's prologue and tail, but a sequence of more than 128 tokens now takes eight
// half-steps with two barriers each instead of four steps with one, and restages its Q / dO tiles per pass.  Kept as the record
// of the experiment.
// ---------------------------------------------------------------------------------------------------------------------
// Round 6 -- the same backward as a KEY-PASS kernel: 256 threads per (sequence, head), TWO workgroups per CU.
// The 8-wave kernel above needs 145 KB of LDS and 250 registers, so a CU holds one workgroup, and that workgroup's memory phases
// (18 k cycles before the first arithmetic of a 256-token item, 8 k of stores and last contraction behind it) overlap nobody's
// arithmetic: its eight waves move in lock step.  Here four waves own 32 keys each -- 128 keys per PASS -- and a sequence of more
// than 128 tokens is walked in two passes; every pass runs the whole query loop for its keys.  One dS^T set (two tiles) instead of
// two sets of four, two K tiles instead of four: 76 KB, so two workgroups are resident per CU (still eight waves, 256 registers
// each) and one's loads, barriers and stores run under the other's MFMAs.
//   dK, dV: complete per pass (a pass owns its keys).
//   dQ:     every wave is also the dQ wave of one 32-query x 32-dimension block; a step's block is contracted over the PASS's keys
//           one step late.  Pass 0 of a two-pass sequence leaves it in fp32 in `dq32` (same thread, same addresses in pass 1:
//           program order is all the ordering needed -- no atomics, one fixed summation order), the last pass adds and stores bf16.
// Costs: Q / dO tiles and the keep words are staged once per pass (twice for > 128 tokens), a second barrier per step (the single
// dS^T set is read by the dQ contraction before the step's body rewrites it).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ATTP_SET = 2 * ATT_TILE + 512;          // Q tile | dO tile | 64 LSE (+ pad)
constexpr int ATTP_K = 2 * ATTP_SET;                  // two K tiles (the pass's keys)
constexpr int ATTP_DS = ATTP_K + 2 * ATT_TILE;        // ONE set of two dS^T tiles
constexpr int ATTP_D = ATTP_DS + 2 * ATT_TILE;        // D of the sequence's queries (256 floats)
constexpr int ATTP_PARK = ATTP_D + 1024;              // park regions of the four dQ blocks (2 KB each)
constexpr int ATTP_M = ATTP_PARK + 4 * 2048;          // two sets of 4 x 64 dropout keep words (the pass's four pieces)
constexpr int ATTP_M_SET = 4 * 256;
constexpr int ATTP_SMEM = ATTP_M + 2 * ATTP_M_SET;    // 77,824 bytes: two per CU
static_assert(ATTP_SMEM <= 80 * 1024, "two key-pass workgroups must fit one CU's LDS");

template <bool DROP>
static __global__ void __launch_bounds__(256, 2) k_attention_bwd_kp(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = a.order ? a.order[blockIdx.y] : (int)blockIdx.y, h = blockIdx.x;
  const int len = a.lens[b];
  const int64_t base = a.cu[b];
  const int plen = a.cu[b + 1] - (int)base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 .. 3
  const int hi = lane >> 5, li = lane & 31;
  const int H = a.H, H3 = 3 * a.H;
  const float c = a.scale * 1.44269504088896341f;
  const int qrow = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);
  const int qsw = (qrow >> 1) & 7;
  const AttnTileSrc srcQ = attn_tile_src(a.QKV, H3, a.rows, lane);
  const AttnTileSrc srcO = attn_tile_src(a.dO, H, a.rows, lane);
  const AttnTileSrc srcL = attn_tile_src((const bf16_t*)a.LSE, 2 * a.ldt, a.heads, lane);
  const AttnTileSrc srcM = attn_tile_src((const bf16_t*)a.mbits, 2 * a.ldt, DROP ? a.heads * ATTM_PIECES : 0, lane);
  const TrLane trl = tr_lane(lane);
  const uint32_t s0 = lds_off(smem);
  const int qlen = (a.q_limit > 0 && a.q_limit < len) ? a.q_limit : len;   // queries that carry gradient (last layer: the CLS tile)
  const int npass = (len + 127) >> 7;
  float* sDall = (float*)(smem + ATTP_D);
  // ---- D[q] = dO[q] . O[q] for the whole sequence, once: thread (query = t >> 1 of a 128-query round, half of the dimensions) ----
  {
    const int qd = threadIdx.x >> 1, dhalf = threadIdx.x & 1;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      if (rd * 128 >= len) break;
      const int q = rd * 128 + qd, qc = q < len ? q : len - 1;
      const bf16_t* dp_ = a.dO + (base + qc) * H + h * 64 + 32 * dhalf;
      const bf16_t* op_ = a.O + (base + qc) * H + h * 64 + 32 * dhalf;
      float acc = 0.f;
      const bool fp32_row = a.cls32 && q == 0;   // query 0 from the fp32 context row (AttnBwdArgs::cls32)
      const float* o32 = a.cls32 + (int64_t)b * H + h * 64 + 32 * dhalf;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        union { bf16x8 v; uint32_t u[4]; } x, y;
        x.v = *(const bf16x8*)(dp_ + 8 * s);
        y.v = *(const bf16x8*)(op_ + 8 * s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o0 = fp32_row ? o32[8 * s + 2 * j] : __uint_as_float(y.u[j] << 16);
          const float o1 = fp32_row ? o32[8 * s + 2 * j + 1] : __uint_as_float(y.u[j] & 0xffff0000u);
          acc += __uint_as_float(x.u[j] << 16) * o0 + __uint_as_float(x.u[j] & 0xffff0000u) * o1;
        }
      }
      acc += __shfl_xor(acc, 1, 64);
      if (dhalf == 0) sDall[q] = acc;
    }
  }
  // this wave's 32 rows of the (single) dS^T set, and its dQ block
  const int ds_row = (wave & 1) * 32 + li;
  char* const ds_cur = smem + ATTP_DS + (wave >> 1) * ATT_TILE + ds_row * 128;
  const int ds_sw = (ds_row >> 1) & 7;
  const int qb = wave & 1, dtq = wave >> 1;
  TrLane trq = trl, trk = trl;
  trq.a[0][0] = qb ? trl.a[1][0] : trl.a[0][0];
  trq.a[0][1] = qb ? trl.a[1][1] : trl.a[0][1];
  trk.a[0][0] = dtq ? trl.a[1][0] : trl.a[0][0];
  trk.a[0][1] = dtq ? trl.a[1][1] : trl.a[0][1];
  const int e_lane = 16 * (wave & 1) + 8 * ((li >> 4) & 1) + (li & 7);
  const int p_lane = 2 * (wave >> 1) + ((li >> 3) & 1);          // piece inside the pass (0 .. 3)
  const uint32_t drop_scale_bits = __float_as_uint(a.drop.scale);

  for (int kp = 0; kp < npass; ++kp) {
    const int k0 = kp * 128;
    const int key = k0 + wave * 32 + li;
    const int kc = key < len ? key : len - 1;
    const bool key_ok = key < len;
    const bool active = k0 + wave * 32 < len;
    bf16x8 kf[4], vf[4];
    {
      const bf16_t* kptr = a.QKV + (base + kc) * H3 + H + h * 64 + 8 * hi;
      const bf16_t* vptr = a.QKV + (base + kc) * H3 + 2 * H + h * 64 + 8 * hi;
#pragma unroll
      for (int s = 0; s < 4; ++s) { kf[s] = *(const bf16x8*)(kptr + 16 * s); vf[s] = *(const bf16x8*)(vptr + 16 * s); }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    // four waves stage a 64-row tile as two 8-row rounds each
    auto stage_tile = [&](const AttnTileSrc& src, int64_t first_row, uint32_t col_bytes, char* lds) {
      attn_stage_rows8(src, first_row, col_bytes, lds, wave, lane);
      attn_stage_rows8(src, first_row, col_bytes, lds, wave + 4, lane);
    };
    auto stage = [&](int q0, int buf) {
      char* set = smem + buf * ATTP_SET;
      stage_tile(srcQ, base + q0, (uint32_t)(h * 64 * 2), set);
      stage_tile(srcO, base + q0, (uint32_t)(h * 64 * 2), set + ATT_TILE);
      if (wave == 0) {
        const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(((int64_t)h * a.ldt + base + q0) * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcL.rsrc, (lptr_t)(set + 2 * ATT_TILE), 4, (uint32_t)lane * 4, soff, 0, 0);
      }
      if constexpr (DROP) {   // wave w: the keep words of the pass's piece w (key tile 2 kp + (w >> 1), half-groups w & 1)
        if (k0 + (wave >> 1) * 64 < len) {
          const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(((int64_t)(h * ATTM_PIECES + 4 * kp + wave) * a.ldt + base + q0) * 4));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srcM.rsrc, (lptr_t)(smem + ATTP_M + buf * ATTP_M_SET + wave * 256), 4,
                                                   (uint32_t)lane * 4, soff, 0, 0);
        }
      }
    };
    stage(0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (k0 + t * 64 < len) stage_tile(srcQ, base + k0 + t * 64, (uint32_t)((H + h * 64) * 2), smem + ATTP_K + t * ATT_TILE);
    if (qlen > 64) stage(64, 1);
    if (!active) {   // rows of a covered tile whose wave owns no key of this sequence: zero for the whole pass
#pragma unroll
      for (int i = 0; i < 4; ++i) *(uint4*)(ds_cur + (hi * 4 + i) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // dQ block of the 64-query step starting at qs, contracted over this pass's keys
    auto dq_phase = [&](int qs) __attribute__((always_inline)) {
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
      for (int t = 0; t < 2 && k0 + t * 64 < len; ++t) {
        const uint32_t tK = s0 + ATTP_K + t * ATT_TILE, tS = s0 + ATTP_DS + t * ATT_TILE;
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
      // lane (query r0 + li, hi) holds dimensions 32 dtq + 8 g + 4 hi + 0..3 in dq[4 g + 0..3]
      float* p32 = a.dq32 + (base + r0 + li) * H + h * 64 + 32 * dtq + 4 * hi;
      const bool row_ok = r0 + li < plen;
      if (kp > 0 && row_ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *(const float4*)(p32 + 8 * g);
          dq[4 * g] += v.x; dq[4 * g + 1] += v.y; dq[4 * g + 2] += v.z; dq[4 * g + 3] += v.w;
        }
      }
      if (kp + 1 < npass) {
        if (row_ok) {
#pragma unroll
          for (int g = 0; g < 4; ++g) *(float4*)(p32 + 8 * g) = make_float4(dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]);
        }
      } else {
        const float keep = r0 + li < len ? a.scale : 0.f;   // (dS is parked without the softmax scale)
        attn_park_store_half(smem + ATTP_PARK + wave * 2048, dq, keep, lane, a.dQKV + (base + r0) * H3 + h * 64 + 32 * dtq, H3,
                             plen - r0);
      }
    };
    auto step_body = [&](const int q0, const int buf) __attribute__((always_inline)) {
      const char* sQ = smem + buf * ATTP_SET;
      const char* sdO = sQ + ATT_TILE;
      const float* sLse = (const float*)(sQ + 2 * ATT_TILE);
      const float* sD = sDall + q0;
      const uint32_t* sM = (const uint32_t*)(smem + ATTP_M + buf * ATTP_M_SET + p_lane * 256);
      const uint32_t tQ = s0 + buf * ATTP_SET, tO = tQ + ATT_TILE;
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
        uint32_t kw[16];
        if constexpr (DROP) {
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
          if constexpr (DROP) {
            const float m = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)kw[r], (uint32_t)e_lane, 1u) & drop_scale_bits);
            dpr *= m;
            pd *= m;
          }
          s[r] = pd;
          dp[r] = p * (dpr - sD[qi]);
        }
        if (__builtin_expect(q0 + 64 > len, 0)) {   // the step reaches past the last query: selects, never 0 * junk
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
          const int chunk = 4 * qt + 2 * half + hi;
          const uint4 z = key_ok ? make_uint4(sb.u[0], sb.u[1], sb.u[2], sb.u[3]) : make_uint4(0u, 0u, 0u, 0u);
          *(uint4*)(ds_cur + ((chunk ^ ds_sw) << 4)) = z;
        }
      }
    };
    int it = 0;
    for (int q0 = 0; q0 < qlen; q0 += 64, ++it) {
      const int buf = it & 1;
      lds_dma_wait_all();
      __syncthreads();   // tile `it` has landed; dS^T of step it - 1 is complete
      if (it >= 1) {
        if (q0 + 64 < qlen) stage(q0 + 64, buf ^ 1);
        dq_phase(q0 - 64);
        __syncthreads();   // the dS^T set has been read: the step's body may rewrite it
      }
      if (active) step_body(q0, buf);
    }
    __syncthreads();   // the last step's dS^T is complete; the Q / dO tiles are dead
    if (active) {
      const float keep = key_ok ? 1.f : 0.f;
      const int r0 = k0 + wave * 32;
      attn_park_store(smem + wave * 4096, dk, keep * a.scale, lane, a.dQKV + (base + r0) * H3 + H + h * 64, H3, plen - r0);
      attn_park_store(smem + wave * 4096, dv, keep, lane, a.dQKV + (base + r0) * H3 + 2 * H + h * 64, H3, plen - r0);
    }
    dq_phase((it - 1) * 64);
    __syncthreads();   // end of the pass: every LDS region may be restaged
  }
}

