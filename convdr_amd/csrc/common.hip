// Error plumbing shared by every translation unit of libconvdr_hip.so.
#include "common.hpp"

#include <stdarg.h>

#include <vector>

#include "../../include/convdr_hip.h"

namespace convdr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return -2;
}

int device_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;
    n = cus & ~7;
  }
  return n;
}

// ---- optional per-kernel timing (hipEvents on the launch stream), used by bench.py ------------
struct ProfSpan { const char* name; hipEvent_t a, b; };
static std::vector<ProfSpan> g_spans;
static bool g_prof = false;

int prof_begin(const char* name, hipStream_t st) {
  if (!g_prof) return -1;
  ProfSpan s{name, nullptr, nullptr};
  if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return -1;
  (void)hipEventRecord(s.a, st);
  g_spans.push_back(s);
  return (int)g_spans.size() - 1;
}
void prof_end(int idx, hipStream_t st) {
  if (idx >= 0) (void)hipEventRecord(g_spans[idx].b, st);
}
static void prof_clear() {
  for (auto& s : g_spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  g_spans.clear();
}

}  // namespace convdr

extern "C" int convdr_prof_enable(int on) {
  convdr::prof_clear();
  convdr::g_prof = on != 0;
  return 0;
}

extern "C" int convdr_prof_collect(const char* name, float* total_ms, int* launches) {
  float tot = 0.f;
  int cnt = 0;
  for (auto& s : convdr::g_spans) {
    if (strcmp(s.name, name) != 0) continue;
    if (hipEventSynchronize(s.b) != hipSuccess) return convdr::hip_fail(hipGetLastError(), "hipEventSynchronize");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) return convdr::hip_fail(hipGetLastError(), "hipEventElapsedTime");
    tot += ms;
    ++cnt;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return 0;
}

extern "C" int convdr_version(void) { return 100; }

extern "C" int convdr_device_pci_bus_id(int device, char* out, int len) {
  CONVDR_REQUIRE(out != nullptr && len >= 16, "convdr_device_pci_bus_id: buffer of %d bytes", len);
  CONVDR_CHECK_HIP(hipDeviceGetPCIBusId(out, len, device));
  return 0;
}
extern "C" const char* convdr_last_error(void) { return convdr::g_err; }
