// Prototype (NOT product): what would a fused FFN1 -> FFN2 kernel of the inference encoder have to stream through LDS, and how
// fast does one CU take it?  (VERDICT r05 "Next 2": H_j = gelu(X W1_j^T + b1_j) parked in LDS and contracted into the
// row-complete [tile, 768] accumulators, `Hm` never reaching HBM.)
//
// The fused kernel's row tile owns ALL of W1 and W2 (2 x 768 x 3072 bf16 = 9.44 MB) per tile of T rows, where the two separate
// kernels of today amortise a weight tile over 256 rows (FFN1, 256 x 256 tiles) resp. stream W2 once per 128 rows (FFN2 + LayerNorm,
// 128 x 768 tiles).  And X [T, 768] must be the A operand of all 48 column slices of FFN1: at T = 128 it is 196 KB -- more
// than the CU's 160 KB of LDS, and as register fragments (96 VGPRs per lane of 512 threads) it does not fit beside the 192
// accumulators of the [128, 768] output -- so it is re-read from L2 per slice (another 9.44 MB per tile); at T = 64 it fits
// (98 KB) but every weight byte is then amortised over half the rows.
//
// This program measures the one thing that prices all of it: the rate at which ONE workgroup per CU (512 threads, 160 KB LDS, the
// product kernels' 16-byte LDS-DMA through a buffer descriptor, three 48 KB slots in flight) can stream L2-resident operands,
// all 256 CUs at once -- with no arithmetic at all (mode 0) and with the fused kernel's MFMA count issued beside the stream
// (mode 1: 2 x T x 768 x 3072 x 2 FLOP per tile on register operands, no LDS fragment reads: an upper bound on the overlap).
//   mlp_stream_proto [tiles_per_cu = 8]
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/mlp_stream_proto.hip -o gpurun_out/mlp_stream_proto
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define CK(x)                                                                                          \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } \
  } while (0)

constexpr int SLOT = 48 * 1024;          // one K slice of all 768 output rows: 768 x 32 bf16 (the product's K-slice-major weight chunk)
constexpr int NSLOT = 3;

// bytes: what one tile streams; mfma_per_chunk: MFMAs (32x32x16 bf16) each wave issues per 48 KB chunk
template <bool MFMA>
__global__ void __launch_bounds__(512, 1) k_stream(const char* __restrict__ src, size_t src_bytes, size_t bytes_per_tile, int tiles,
                                                   int mfma_per_chunk, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t b = (uint64_t)src;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (uint32_t)(src_bytes > 0xffffffffull ? 0xffffffffu : src_bytes), 0x00020000);
  const size_t chunks = bytes_per_tile / SLOT;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fa, fb;
#pragma unroll
  for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(0.001f * (lane + j)); fb[j] = (__bf16)(0.002f * (lane - j)); }
  // every workgroup walks the same 9.44 MB window (L2 / Infinity-Cache resident after the first pass), offset by its index
  size_t off = ((size_t)blockIdx.x * 37 % 64) * SLOT;
  auto issue = [&](int slot) {
    // 48 KB = 48 wave instructions of 1 KB; 8 waves -> 6 each
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const uint32_t piece = (uint32_t)(wave * 6 + i) * 1024u;
      const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)((off + piece) % (src_bytes - SLOT)));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + slot * SLOT + piece), 16, (uint32_t)lane * 16, soff, 0, 0);
    }
    off += SLOT;
  };
  for (int t = 0; t < tiles; ++t) {
    issue(0);
    issue(1);
    for (size_t c = 0; c < chunks; ++c) {
      if (c + 2 < chunks) issue((int)((c + 2) % NSLOT));
      if (MFMA) {
        for (int m = 0; m < mfma_per_chunk; m += 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
        }
      }
      // chunk c has landed when at most the two younger chunks' 12 instructions are outstanding
      if (c + 2 < chunks) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (c + 1 < chunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  s += (float)smem[(threadIdx.x * 16) % (NSLOT * SLOT)];
  if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? std::atoi(argv[1]) : 8;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t W = (size_t)2 * 768 * 3072 * 2;           // W1 + W2, bf16
  const size_t X128 = (size_t)128 * 768 * 2 * 48;        // the X tile re-read for each of the 48 column slices of FFN1 (T = 128)
  char* d;
  float* sink;
  const size_t src_bytes = W + SLOT * 66;
  CK(hipMalloc(&d, src_bytes));
  CK(hipMemset(d, 1, src_bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipFuncSetAttribute((const void*)k_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT));
  CK(hipFuncSetAttribute((const void*)k_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Case { const char* name; size_t bytes; int T; bool mfma; };
  const Case cases[] = {
      {"T=128, weights only (X resident: impossible, 196 KB)      ", W, 128, false},
      {"T=128, weights + X re-read per slice                      ", W + X128, 128, false},
      {"T= 64, weights only (X resident in 98 KB of LDS)          ", W, 64, false},
      {"T=128, weights + X re-read, with the tile's MFMAs beside  ", W + X128, 128, true},
      {"T= 64, weights only, with the tile's MFMAs beside         ", W, 64, true},
  };
  std::printf("device: %s, %d CUs; one 512-thread workgroup per CU, %d tiles each, 3 x 48 KB LDS-DMA slots\n", prop.name, cus, tiles);
  std::printf("today (BENCH_r05, 262,144 rows): FFN1 1.20 ms + FFN2+LayerNorm 1.16 ms = 2.36 ms per layer = 9.0 us per 1,000 rows\n");
  for (const Case& c : cases) {
    const size_t bytes = c.bytes / SLOT * SLOT;
    const size_t chunks = bytes / SLOT;
    // MFMAs of one tile: 2 GEMMs x T x 768 x 3072 x 2 FLOP / (32 x 32 x 16 x 2 FLOP) per MFMA, spread over 8 waves and the chunks
    const double mfma_tile = 2.0 * c.T * 768.0 * 3072.0 * 2.0 / (32.0 * 32.0 * 16.0 * 2.0);
    int per_chunk = (int)(mfma_tile / 8.0 / (double)chunks + 0.5);
    per_chunk = (per_chunk + 3) / 4 * 4;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (c.mfma) hipLaunchKernelGGL(k_stream<true>, dim3(cus), dim3(512), NSLOT * SLOT, 0, d, src_bytes, bytes, tiles, per_chunk, sink);
      else hipLaunchKernelGGL(k_stream<false>, dim3(cus), dim3(512), NSLOT * SLOT, 0, d, src_bytes, bytes, tiles, per_chunk, sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_tile = ms * 1e3 / tiles;
    const double tbs = (double)bytes * tiles * cus / (ms * 1e-3) / 1e12;
    const double layer_ms = us_tile * (262144.0 / c.T / cus) / 1e3;
    std::printf("%s %7.1f us per tile  %5.2f TB/s L2->LDS (all CUs)  => %5.2f ms per layer of 262,144 rows%s\n", c.name, us_tile, tbs, layer_ms,
                c.mfma ? "  [MFMAs on register operands: no LDS fragment reads, no epilogues]" : "  [stream only: a floor]");
  }
  return 0;
}
