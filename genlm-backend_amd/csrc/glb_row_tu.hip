// glb_row_tu.hip — one translation unit per (element type, draw mode); compiled with
// -DGLB_DT=<0|1|2> -DGLB_MODE=<0|1|2>.  Instantiates the row kernel for every mask kind and launch
// geometry and exports one launcher that glb_api.hip dispatches to.
#include "glb_row_kernel.hpp"

#ifndef GLB_DT
#error "GLB_DT not defined"
#endif
#ifndef GLB_MODE
#error "GLB_MODE not defined"
#endif

namespace glb {

// geometry table: id -> (threads, vectors per lane).  Capacity = threads * NVL 16-byte vectors.
//   1:(256,4)  2:(1024,4)  3:(1024,8)  4:(1024,13)  5:(1024,16)
template <int MASK, int NVL, int T>
static hipError_t launch1(const RowParams &p, hipStream_t s) {
  hipLaunchKernelGGL((row_kernel<GLB_DT, MASK, GLB_MODE, NVL, T>), dim3(p.n_particles), dim3(T), 0, s, p);
  return hipGetLastError();
}

template <int MASK>
static hipError_t launch_geom(const RowParams &p, int geom, hipStream_t s) {
  switch (geom) {
    case 1: return launch1<MASK, 4, 256>(p, s);
    case 2: return launch1<MASK, 4, 1024>(p, s);
    case 3: return launch1<MASK, 8, 1024>(p, s);
    case 4: return launch1<MASK, 13, 1024>(p, s);
    case 5: return launch1<MASK, 16, 1024>(p, s);
    default: return hipErrorInvalidValue;
  }
}

#define GLB_CAT_(a, b, c) a##b##_##c
#define GLB_CAT(a, b, c) GLB_CAT_(a, b, c)

hipError_t GLB_CAT(launch_row_, GLB_DT, GLB_MODE)(const RowParams &p, int mask_kind, int geom,
                                                  hipStream_t s) {
  switch (mask_kind) {
    case kMaskNone: return launch_geom<kMaskNone>(p, geom, s);
    case kMaskBits: return launch_geom<kMaskBits>(p, geom, s);
    case kMaskF32: return launch_geom<kMaskF32>(p, geom, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace glb
