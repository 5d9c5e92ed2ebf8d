// glb_row_tu.hip — one translation unit per (element type, draw mode); compiled with
// -DGLB_DT=<0|1|2> -DGLB_MODE=<0|1|2>.  Instantiates the row kernel for every mask kind and launch
// geometry and exports one launcher that glb_api.hip dispatches to.
#include "glb_row_persist.hpp"
#include "glb_row_stream.hpp"

#ifndef GLB_DT
#error "GLB_DT not defined"
#endif
#ifndef GLB_MODE
#error "GLB_MODE not defined"
#endif

namespace glb {

#define GLB_CAT3_(a, b, c) a##b##_##c
#define GLB_CAT3(a, b, c) GLB_CAT3_(a, b, c)

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

// persistent kernel (glb_row_persist.hpp), 512 threads; ids -> NVL (capacity 512*NVL 16-byte vectors per row):
//   24: 8   22: 16   21: 25 (fp32 gpt2-sized rows)   23: 32 (16-bit 128k rows)   25: 40 (16-bit 152k rows, fp32 up to 81 912)
#if GLB_MODE != 2
template <int MASK, int NVL>
static hipError_t launch2(const RowParams &p0, int grid, hipStream_t s) {
  const RowParams &p = p0;
  if (p.use_scale)
    hipLaunchKernelGGL((row_kernel_persist<GLB_DT, MASK, GLB_MODE, NVL, 512, true>), dim3(grid), dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL((row_kernel_persist<GLB_DT, MASK, GLB_MODE, NVL, 512, false>), dim3(grid), dim3(512), 0, s, p);
  return hipGetLastError();
}

template <int MASK>
static hipError_t launch_geom2(const RowParams &p, int geom, int grid, hipStream_t s) {
  switch (geom) {
    case 24: return launch2<MASK, 8>(p, grid, s);
    case 22: return launch2<MASK, 16>(p, grid, s);
    case 21: return launch2<MASK, 25>(p, grid, s);
    case 23: return launch2<MASK, 32>(p, grid, s);
    case 25: return launch2<MASK, 40>(p, grid, s);
    default: return hipErrorInvalidValue;
  }
}
#endif


#define GLB_CAT_(a, b, c) a##b##_##c
#define GLB_CAT(a, b, c) GLB_CAT_(a, b, c)

hipError_t GLB_CAT(launch_row_, GLB_DT, GLB_MODE)(const RowParams &p, int mask_kind, int geom,
                                                  hipStream_t s) {
#if GLB_MODE != 2
  if (geom == 99) {  // streaming fallback for rows beyond the register-resident capacity
    if (mask_kind == kMaskNone)
      hipLaunchKernelGGL((row_stream_kernel<GLB_DT, kMaskNone, GLB_MODE>), dim3(p.n_particles), dim3(1024), 0, s, p);
    else if (mask_kind == kMaskBits)
      hipLaunchKernelGGL((row_stream_kernel<GLB_DT, kMaskBits, GLB_MODE>), dim3(p.n_particles), dim3(1024), 0, s, p);
    else
      return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if ((geom & 0xff) >= 21) {
    const int grid = geom >> 8;  // persistent grid size rides in the upper bits
    const int g = geom & 0xff;
    switch (mask_kind) {
      case kMaskNone: return launch_geom2<kMaskNone>(p, g, grid, s);
      case kMaskBits: return launch_geom2<kMaskBits>(p, g, grid, s);
      default: return hipErrorInvalidValue;
    }
  }
#endif
  switch (mask_kind) {
    case kMaskNone: return launch_geom<kMaskNone>(p, geom, s);
    case kMaskBits: return launch_geom<kMaskBits>(p, geom, s);
    case kMaskF32: return launch_geom<kMaskF32>(p, geom, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace glb
