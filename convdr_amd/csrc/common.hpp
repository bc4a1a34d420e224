// Shared device/host helpers for libconvdr_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace convdr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t bf16_t;  // storage type for bf16 in global memory

// ---- error reporting across the C ABI ------------------------------------
void set_error(const char* fmt, ...);
extern int64_t g_ip_fused_finish;   // ip_topk.hip; convdr_set_option("ip_fused_finish") lives in encoder.hip
int hip_fail(hipError_t e, const char* what);
// compute units of the current device, rounded down to a multiple of the 8 XCDs (256 on MI355X); persistent kernels
// launch this many workgroups (x resident workgroups per CU)
int device_cu_count();

#define CONVDR_CHECK_HIP(expr)                                   \
  do {                                                           \
    hipError_t _e = (expr);                                      \
    if (_e != hipSuccess) return ::convdr::hip_fail(_e, #expr);  \
  } while (0)

#define CONVDR_CHECK_LAUNCH(name)                                        \
  do {                                                                   \
    hipError_t _e = hipGetLastError();                                   \
    if (_e != hipSuccess) return ::convdr::hip_fail(_e, "launch " name); \
  } while (0)

#define CONVDR_REQUIRE(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      ::convdr::set_error(__VA_ARGS__);  \
      return -1;                         \
    }                                    \
  } while (0)

// "first call on this device?" for per-device one-time setup (hipFuncSetAttribute is a property of the function ON a
// device: a process that drives several GPUs must repeat it for each).  Not thread-safe by design: one caller thread per
// device, and repeating the setup is harmless.
struct DeviceOnce {
  bool done[64] = {};
  bool first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};

// ---- optional per-kernel hipEvent timing (convdr_prof_enable / convdr_prof_collect) ----------
int prof_begin(const char* name, hipStream_t st);
void prof_end(int idx, hipStream_t st);
struct ProfScope {
  int idx; hipStream_t st;
  ProfScope(const char* name, hipStream_t s) : idx(prof_begin(name, s)), st(s) {}
  ~ProfScope() { prof_end(idx, st); }
};

// ---- bf16 <-> f32 -----------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32 on gfx950 (the compiler selects it
// from the vector conversion; the bit-twiddling form above costs ~6 VALU per element)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *(const uint32_t*)&r;
}

// same result from integer ops only; used where the hardware-convert form upsets register allocation (gemm_ln.hpp)
__device__ __forceinline__ uint32_t pack_bf16x2_sw(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ---- wave (64 lanes) reductions -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): give every XCD a
// contiguous chunk of the logical tile order so neighbouring tiles share its L2.
// Bijective for any grid size (cdna guide §5 "XCD swizzle must be bijective").
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
  const uint32_t xcd = bid & 7u, idx = bid >> 3;
  const uint32_t q = nwg >> 3, r = nwg & 7u;
  const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace convdr
