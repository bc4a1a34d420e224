// A caller of libconvdr_hip.so with no Python and no torch anywhere: plain HIP runtime allocations, the C ABI of
// include/convdr_hip.h, and a brute-force fp64 check on the host.  What a C / C++ (or cgo / JNI) host of the reference's
// FAISS call sites (run_convdr_inference.py:180-182) would do.  Built and run by tests/test_capi_host_gpu.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/convdr_hip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
  } while (0)
#define CV(x)                                                                   \
  do {                                                                          \
    if ((x) != 0) { std::printf("convdr error: %s (%s:%d)\n", convdr_last_error(), __FILE__, __LINE__); return 3; } \
  } while (0)

static int run(bool f16);
int main() {
  const int a = run(false);
  if (a) return a;
  return run(true);
}

static int run(bool f16) {
  const int64_t n = 20000;
  const int d = 128, nq = 37, k = 50, cap = 4096;
  std::vector<float> P((size_t)n * d), Q((size_t)nq * d);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : P) v = rnd();
  for (auto& v : Q) v = rnd();
  for (int j = 0; j < d; ++j) P[(size_t)777 * d + j] = P[(size_t)42 * d + j];   // an exact tie: lower index first

  float *dP, *dQ, *dD, *dTau, *dMax, *dScratch, *dCentre;
  void *dPb, *dWs;
  int64_t* dI;
  int32_t* dStatus;
  CK(hipMalloc(&dP, P.size() * 4)); CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMalloc(&dPb, P.size() * 2));
  CK(hipMalloc(&dD, (size_t)nq * k * 4)); CK(hipMalloc(&dI, (size_t)nq * k * 8)); CK(hipMalloc(&dStatus, nq * 4));
  CK(hipMalloc(&dTau, nq * 4)); CK(hipMalloc(&dMax, 4)); CK(hipMalloc(&dScratch, 1024 * d * 4)); CK(hipMalloc(&dCentre, d * 4));
  CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dMax, 0, 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  CV(convdr_ip_column_mean(dP, n, d, dScratch, dCentre, st));
  const size_t ws_bytes = convdr_ip_workspace_bytes(nq, n, d, k, cap);
  CK(hipMalloc(&dWs, ws_bytes));
  if (!f16) {   // the bf16 rung
    CV(convdr_ip_prepare_block(dP, n, d, dCentre, dPb, nullptr, dMax, st));
    CV(convdr_ip_search(dQ, nq, dP, dPb, nullptr, n, d, k, dMax, nullptr, cap, 0, dWs, ws_bytes, dD, dI, dStatus, dTau, st));
  } else {      // the fp16 rung: a norm-only pass finds the power-of-two scale of the scan copy
    CV(convdr_ip_prepare_block_f16(dP, n, d, dCentre, 1.f, nullptr, nullptr, dMax, st));
    float mx = 0.f;
    CK(hipMemcpyAsync(&mx, dMax, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const float scale = convdr_ip_f16_scale(mx);
    CV(convdr_ip_prepare_block_f16(dP, n, d, dCentre, scale, dPb, nullptr, dMax, st));
    CV(convdr_ip_search_f16(dQ, nq, dP, dPb, nullptr, scale, n, d, k, dMax, nullptr, cap, 0, dWs, ws_bytes, dD, dI, dStatus, dTau, st));
  }
  CK(hipStreamSynchronize(st));
  std::vector<float> D((size_t)nq * k);
  std::vector<int64_t> I((size_t)nq * k);
  std::vector<int32_t> status(nq);
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(I.data(), dI, I.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(status.data(), dStatus, nq * 4, hipMemcpyDeviceToHost));

  int bad = 0;
  std::vector<std::pair<double, int64_t>> sc(n);
  for (int q = 0; q < nq; ++q) {
    if (status[q] != CONVDR_IP_OK) { std::printf("query %d: status %d\n", q, status[q]); ++bad; continue; }
    for (int64_t i = 0; i < n; ++i) {
      double a = 0;
      for (int j = 0; j < d; ++j) a += (double)P[(size_t)i * d + j] * (double)Q[(size_t)q * d + j];
      sc[i] = {a, i};
    }
    std::partial_sort(sc.begin(), sc.begin() + k, sc.end(), [](const auto& x, const auto& y) {
      return x.first > y.first || (x.first == y.first && x.second < y.second);
    });
    for (int j = 0; j < k; ++j) {
      // ids must match wherever the exact scores are distinguishable in fp64 (the library's canonical summation order
      // is its own: compare scores to 1e-5 relative, ids exactly unless two neighbours are that close)
      const bool close_pair = (j + 1 < k && std::fabs(sc[j].first - sc[j + 1].first) < 1e-9 * std::fabs(sc[j].first)) ||
                              (j > 0 && std::fabs(sc[j].first - sc[j - 1].first) < 1e-9 * std::fabs(sc[j].first));
      if (I[(size_t)q * k + j] != sc[j].second && !close_pair) {
        std::printf("query %d rank %d: id %lld, brute force %lld\n", q, j, (long long)I[(size_t)q * k + j], (long long)sc[j].second);
        ++bad;
      }
      if (std::fabs(D[(size_t)q * k + j] - sc[j].first) > 1e-5 * std::fabs(sc[j].first) + 1e-6) {
        std::printf("query %d rank %d: score %g vs %g\n", q, j, D[(size_t)q * k + j], sc[j].first);
        ++bad;
      }
    }
  }
  std::printf(bad ? "MISMATCH %d\n" : "capi host ok (%s scan): %d queries x %lld passages, top-%d exact (ABI version %d)\n",
              bad ? (const char*)"" : (f16 ? "fp16" : "bf16"), bad ? bad : nq, (long long)n, k, convdr_version());
  return bad ? 1 : 0;
}
