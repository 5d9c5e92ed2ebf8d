// glb_common.hpp — what the translation units of libglb_hip.so share besides include/glb.h: the error slot.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

// launch grids: blocks of t threads for n items
inline unsigned blocks_for(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

namespace glb {
// set the calling thread's error message (glb_last_error) and return `code`
int api_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int api_hip_fail(hipError_t e, const char *what);

// More than 64 KiB of dynamic LDS has to be allowed per kernel AND per device (the attribute belongs to the device's copy
// of the function); `done`: one bit per device, owned by the call site.
inline hipError_t allow_dynamic_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
  return e;
}
}  // namespace glb
