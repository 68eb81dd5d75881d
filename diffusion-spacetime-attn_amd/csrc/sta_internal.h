// Internal to libsta_xattn.so: the thread-local error text behind sta_last_error(), the per-device
// "dynamic LDS size raised" bookkeeping, and the tuning hook.
#ifndef STA_INTERNAL_H
#define STA_INTERNAL_H
#include <hip/hip_runtime.h>
extern thread_local char g_sta_err[256];
int sta_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: one flag per (kernel
// instantiation, device), so a process that drives several GPUs raises the limit on each of them.
// (Benign race: the call is idempotent.)
constexpr int STA_MAX_DEVICES = 64;
struct StaLdsAttr {
  bool done[STA_MAX_DEVICES] = {};
  bool ensure(const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 0 && dev < STA_MAX_DEVICES && done[dev]) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (dev >= 0 && dev < STA_MAX_DEVICES) done[dev] = true;
    return true;
  }
};

// kernel-selection overrides set through sta_set_option (include/sta_xattn.h); 0 = automatic. Relaxed atomics: a test or tool
// thread may flip one while another thread launches (each launch reads every key it needs exactly once per decision).
#include <atomic>
#include "sta_xattn.h"
struct StaOpt {
  std::atomic<int> v{0};
  operator int() const { return v.load(std::memory_order_relaxed); }
  void operator=(int x) { v.store(x, std::memory_order_relaxed); }
};
extern StaOpt g_sta_opt[STA_OPT_COUNT];
#endif
