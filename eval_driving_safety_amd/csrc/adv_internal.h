// shared by the translation units of libadvengine.so (not installed, not part of the ABI)
#ifndef ADV_INTERNAL_H
#define ADV_INTERNAL_H
#include <hip/hip_runtime.h>

#include "advengine.h"

// records the hipError_t behind ADV_ELAUNCH for adv_last_hip_error() (thread-local, defined in advengine.hip)
void adv_internal_set_last_hip_error(int e);

static inline int adv_internal_finish_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    adv_internal_set_last_hip_error(static_cast<int>(e));
    return ADV_ELAUNCH;
  }
  return ADV_OK;
}

// Dynamic LDS beyond 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  Raised once per (kernel, device) and
// remembered in one atomic word per kernel - a write-once cache of a constant, so that no launch inside a stream capture makes a
// non-stream runtime call (the first call of a kernel should therefore happen outside a capture, as a warm-up does anyway).
#include <atomic>
#include <cstdint>
template <auto Kernel>
static inline bool adv_internal_lds_limit(size_t bytes) {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) != hipSuccess)
    return false;
  done.fetch_or(bit, std::memory_order_release);
  return true;
}

// A/B and parity-cross-check routes ("alternative code paths").  The SHIPPED library compiles them out: no entry point
// reads the environment, kernel selection depends on the arguments alone.  `make hooks` builds libadvengine_hooks.so with
// -DADV_TEST_HOOKS, in which the ADV_* environment variables are read at each launch; only tests/ and tools/ open that build
// (eval_driving_safety_amd._lib.using).  adv_build_has_test_hooks() tells the two apart.
#ifdef ADV_TEST_HOOKS
#include <cstdlib>
static inline bool adv_hook(const char* name) { return std::getenv(name) != nullptr; }
static inline const char* adv_hook_value(const char* name) { return std::getenv(name); }
#else
static inline constexpr bool adv_hook(const char*) { return false; }
static inline constexpr const char* adv_hook_value(const char*) { return nullptr; }
#endif
#endif
