// glb_common.hpp — what the translation units of libglb_hip.so share besides include/glb.h: the error slot.
#pragma once
#include <hip/hip_runtime.h>

namespace glb {
// set the calling thread's error message (glb_last_error) and return `code`
int api_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int api_hip_fail(hipError_t e, const char *what);
}  // namespace glb
