// glb_diag.hpp - the DIAGNOSTIC build's hooks (make dbg: -DGLB_STAMPS -> tools/dbg/libglb_hip_dbg.so; tools/dbg/stamps*.py,
// tools/ab.sh).  In the product build GLB_DIAG(...) is nothing and none of this exists: no field of a kernel's
// parameters, no instruction, no getenv.  The experiments these hooks served are written up in DESIGN.md §5.
#pragma once
#include <stdint.h>

#ifdef GLB_STAMPS
#define GLB_DIAG(...) __VA_ARGS__
#else
#define GLB_DIAG(...)
#endif
#define GLB_NOW() __builtin_amdgcn_s_memrealtime()

namespace glb {
#ifdef GLB_STAMPS
// placements and timing experiments that were measured and not kept (both placements slower): il_lag >= 0 deals the
// finishing blocks INSIDE the grid, il_lag units behind their rows' stats blocks; short_last deals every row's short last
// chunk after all full chunks; dbg_mode 1: one vector's worth of exponentials, 2: the rows come out of the caches
struct DiagParams {
  int32_t il_lag = -1, short_last = 0, dbg_mode = 0;
};
#endif
}  // namespace glb
