// Counter-based dropout for the training kernels (gfx950).
//
// The reference trains the student with hidden / attention-probability dropout 0.1
// (/root/reference/drivers/run_convdr_train.py:107 `model.train()` on HF BertModel / RobertaModel: dropout after the
// embedding LayerNorm, on the attention probabilities, and after the attention-output and FFN-output dense layers).
// torch's RNG stream cannot be matched, so the mask is DEFINED here, as a pure function of (seed, site, layer, element):
// the forward, the backward (which regenerates it instead of storing it) and the CPU oracle
// (oracle/dropout.py, same arithmetic in numpy uint32) agree bit for bit.
//
//   key      = mix32(seed ^ (site * 0x9E3779B9 + layer * 0x85EBCA6B))                  (host)
//   h        = mix32(pair_index ^ key)                                                 one hash per TWO elements
//   keep(e)  = 16-bit half of h (low half: even element, high half: odd) >= thresh16,  thresh16 = round(p * 65536)
//   value    = keep ? x / (1 - thresh16 / 65536) : 0                                   (inverted dropout, exact rate)
//   pair_index: hidden sites   row * (H / 2) + (col >> 1)          (row = packed token row, col = feature)
//               attention      ((query_row * heads + head) << 9) + (key >> 1)          (key index inside the sequence)
// mix32 is Bob Jenkins' 6-line integer hash: shifts, adds and xors only -- full-rate VALU (v_lshl_add_u32, v_xor3, ...);
// 32-bit integer multiplies are quarter rate on this part, which is what rules out Philox here (4 of them per round).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CONVDR_HD __host__ __device__ __forceinline__
#else
#define CONVDR_HD inline
#endif

namespace convdr {

enum { DROP_SITE_EMB = 0, DROP_SITE_ATTN_OUT = 1, DROP_SITE_FFN_OUT = 2, DROP_SITE_ATT_PROBS = 3 };

CONVDR_HD uint32_t drop_mix32(uint32_t a) {
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}

struct DropSite {
  uint32_t key;       // per (seed, site, layer)
  uint32_t thresh;    // 16-bit threshold; 0 = dropout off
  float scale;        // 1 / keep probability
};

static inline DropSite drop_site(uint32_t seed, int site, int layer, float p) {
  DropSite d;
  uint32_t t = (uint32_t)(p * 65536.0f + 0.5f);
  if (t > 65535u) t = 65535u;
  d.thresh = p > 0.f ? t : 0u;
  d.key = drop_mix32(seed ^ ((uint32_t)site * 0x9E3779B9u + (uint32_t)layer * 0x85EBCA6Bu));
  d.scale = d.thresh ? 65536.0f / (float)(65536u - d.thresh) : 1.f;
  return d;
}

#if defined(__HIPCC__)
// multipliers (0 or scale) of the two elements of pair `pair_index`
__device__ __forceinline__ void drop_pair(const DropSite& d, uint32_t pair_index, float& m_even, float& m_odd) {
  const uint32_t h = drop_mix32(pair_index ^ d.key);
  m_even = (h & 0xffffu) >= d.thresh ? d.scale : 0.f;
  m_odd = (h >> 16) >= d.thresh ? d.scale : 0.f;
}
// four consecutive features col .. col + 3 (col % 4 == 0) of packed row `row` of an [rows, H] matrix
__device__ __forceinline__ void drop_hidden4(const DropSite& d, int64_t row, int col, int H, float& m0, float& m1, float& m2,
                                             float& m3) {
  const uint32_t base = (uint32_t)row * (uint32_t)(H >> 1) + (uint32_t)(col >> 1);
  drop_pair(d, base, m0, m1);
  drop_pair(d, base + 1u, m2, m3);
}
__device__ __forceinline__ uint32_t drop_att_base(int64_t query_row, int heads, int head) {
  return (((uint32_t)query_row * (uint32_t)heads + (uint32_t)head) << 9);
}
#endif

}  // namespace convdr
