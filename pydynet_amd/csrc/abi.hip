// Library-level entry points of the pdnhip C ABI: error string, version, device queries.
#include "common.h"
#include <stdarg.h>
#include <atomic>

static thread_local char g_err[512] = "";

void pdn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pdn_last_error(void) { return g_err; }

extern "C" int pdn_abi_version(void) { return 1; }

// Number of visible HIP devices (0 when none); never fails.
extern "C" int pdn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Fills name (<= cap bytes) with the gcnArchName of `device`, returns the CU count.
extern "C" int pdn_device_info(int device, char* name, int cap, int* compute_units,
                               int64_t* total_mem) {
  hipDeviceProp_t prop;
  PDN_HIP(hipGetDeviceProperties(&prop, device));
  if (name && cap > 0) {
    strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (total_mem) *total_mem = (int64_t)prop.totalGlobalMem;
  return PDN_OK;
}

extern "C" int pdn_stream_synchronize(void* stream) {
  PDN_HIP(hipStreamSynchronize((hipStream_t)stream));
  return PDN_OK;
}

// Launch counters per kernel (slots: PDN_CNT_* in common.h, listed in include/pdn_hip.h).  Copies min(n, 16) counters
// to `out` (may be null) and clears them when `reset` is non-zero.
static std::atomic<int64_t> g_counters[PDN_CNT_SLOTS];
void pdn_count(int slot) {
  if (slot >= 0 && slot < PDN_CNT_SLOTS) g_counters[slot].fetch_add(1, std::memory_order_relaxed);
}
extern "C" int pdn_kernel_counters(int64_t* out, int n, int reset) {
  for (int i = 0; i < PDN_CNT_SLOTS; ++i) {
    const int64_t v = reset ? g_counters[i].exchange(0) : g_counters[i].load();
    if (out && i < n) out[i] = v;
  }
  return PDN_OK;
}
