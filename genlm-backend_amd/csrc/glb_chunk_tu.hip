// glb_chunk_tu.hip — one translation unit per element type (-DGLB_DT=<0|1|2>): instantiates the chunked step
// kernels for every mask kind / draw mode and exports the launchers glb_api.hip dispatches to.
#include <hip/hip_ext.h>

#include "glb_chunk.hpp"

#ifndef GLB_DT
#error "GLB_DT not defined"
#endif

namespace glb {

#define GLB_CAT_(a, b) a##b
#define GLB_CAT(a, b) GLB_CAT_(a, b)

// glb_logprob_mask_sample_timed (glb_api.hip): HIP events that the step's first / last launch carries as its own start /
// stop time stamps (hipExtLaunchKernel) - the launch duration as the profiler sees it, without marker packets
extern thread_local hipEvent_t g_step_ev_start, g_step_ev_stop;
extern int g_lds_pad[3];

template <class K>
static void launch_k(K kernel, dim3 grid, dim3 block, hipStream_t s, const StepParams &p, bool first, bool last,
                     unsigned lds = 0) {
  hipEvent_t e0 = first ? g_step_ev_start : nullptr, e1 = last ? g_step_ev_stop : nullptr;
  if (e0 || e1) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, e0, e1, 0, p);
  else hipLaunchKernelGGL(kernel, grid, block, lds, s, p);
}

// launches with fewer items than this use four waves per chunk (the chip has 1024 SIMDs; below about one wave per
// two SIMDs the per-wave latency is the launch time)
constexpr int64_t kSmallLaunchItems = 512;

// the term's arithmetic contract (glb_math.hpp): both exist for every element type (round 6: float32 rows gain less from
// the hardware exponential than 16-bit rows - they are bound by memory - but a wave that is done sooner frees its loads'
// slots sooner: 38.0 -> 36.6 us at 1024 x 50257, profiles/r06/ab_contract_f32_v1.log)
constexpr bool kHasHwExp = true;

template <int MASK, int EXPC>
static hipError_t stats1(const StepParams &p, bool scaled, hipStream_t s) {
  const int64_t waves = (int64_t)p.n_pairs * p.nch;
  if constexpr (MASK != kMaskF32) {
    if (waves <= kSmallLaunchItems) {
      const dim3 grid((unsigned)waves), block(256);
      if (scaled) launch_k(chunk_stats_small_kernel<GLB_DT, MASK, true, EXPC>, grid, block, s, p, true, false);
      else launch_k(chunk_stats_small_kernel<GLB_DT, MASK, false, EXPC>, grid, block, s, p, true, false);
      return hipGetLastError();
    }
  }
  const dim3 grid((unsigned)waves), block(64);  // one-wave workgroups: every wave slot refills on its own
  if (scaled) launch_k(chunk_stats_kernel<GLB_DT, MASK, true, EXPC>, grid, block, s, p, true, false);
  else launch_k(chunk_stats_kernel<GLB_DT, MASK, false, EXPC>, grid, block, s, p, true, false);
  return hipGetLastError();
}

template <int EXPC>
static hipError_t stats0(const StepParams &p, int mask_kind, bool scaled, hipStream_t s) {
  switch (mask_kind) {
    case kMaskNone: return stats1<kMaskNone, EXPC>(p, scaled, s);
    case kMaskBits: return stats1<kMaskBits, EXPC>(p, scaled, s);
    case kMaskF32: return stats1<kMaskF32, EXPC>(p, scaled, s);
  }
  return hipErrorInvalidValue;
}

hipError_t GLB_CAT(launch_stats_, GLB_DT)(const StepParams &p, int mask_kind, bool scaled, int expc, hipStream_t s) {
  if constexpr (kHasHwExp) {
    if (expc == kExpHw) return stats0<kExpHw>(p, mask_kind, scaled, s);
  }
  return expc == kExpPoly ? stats0<kExpPoly>(p, mask_kind, scaled, s) : hipErrorInvalidValue;
}

// the step in one launch (statistics / Philox modes): the caller has set the grid's two parts in p
template <int MASK, int MODE, int EXPC>
static hipError_t fused2(const StepParams &p, bool scaled, hipStream_t s) {
  const dim3 grid((unsigned)(p.stats_blocks + p.fin_blocks)), block(64);
  // (dynamic LDS nobody touches: a cap on the workgroups a CU holds, for occupancy experiments; 0 in the product)
  const unsigned lds = (unsigned)g_lds_pad[GLB_DT];
  if (scaled) launch_k(fused_step_kernel<GLB_DT, MASK, true, MODE, EXPC>, grid, block, s, p, true, true, lds);
  else launch_k(fused_step_kernel<GLB_DT, MASK, false, MODE, EXPC>, grid, block, s, p, true, true, lds);
  return hipGetLastError();
}

template <int MASK, int EXPC>
static hipError_t fused1(const StepParams &p, int mode, bool scaled, hipStream_t s) {
  if (mode == kModeStats) return fused2<MASK, kModeStats, EXPC>(p, scaled, s);
  if (mode == kModePhilox) return fused2<MASK, kModePhilox, EXPC>(p, scaled, s);
  return hipErrorInvalidValue;
}

template <int EXPC>
static hipError_t fused0(const StepParams &p, int mask_kind, int mode, bool scaled, hipStream_t s) {
  switch (mask_kind) {
    case kMaskNone: return fused1<kMaskNone, EXPC>(p, mode, scaled, s);
    case kMaskBits: return fused1<kMaskBits, EXPC>(p, mode, scaled, s);
    case kMaskF32: return fused1<kMaskF32, EXPC>(p, mode, scaled, s);
    case kMaskRaw: return fused1<kMaskRaw, EXPC>(p, mode, scaled, s);  // (the one-launch step only: glb_chunk.hpp)
  }
  return hipErrorInvalidValue;
}

hipError_t GLB_CAT(launch_fused_step_, GLB_DT)(const StepParams &p, int mask_kind, int mode, bool scaled, int expc, hipStream_t s) {
  if constexpr (kHasHwExp) {
    if (expc == kExpHw) return fused0<kExpHw>(p, mask_kind, mode, scaled, s);
  }
  return expc == kExpPoly ? fused0<kExpPoly>(p, mask_kind, mode, scaled, s) : hipErrorInvalidValue;
}

template <int MASK, int EXPC>
static hipError_t finish1(const StepParams &p, int mode, hipStream_t s) {
  // one workgroup per particle: one wave, four for the parity-mode race over the whole row
  const dim3 grid((unsigned)p.n_particles), block(mode == kModeNoise ? 256 : 64);
  switch (mode) {
    case kModeStats: launch_k(finish_kernel<GLB_DT, MASK, kModeStats, kExpPoly>, grid, block, s, p, false, true); break;
    case kModePhilox: launch_k(finish_kernel<GLB_DT, MASK, kModePhilox, EXPC>, grid, block, s, p, false, true); break;
    case kModeNoise: launch_k(finish_kernel<GLB_DT, MASK, kModeNoise, EXPC>, grid, block, s, p, false, true); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int EXPC>
static hipError_t finish0(const StepParams &p, int mask_kind, int mode, hipStream_t s) {
  switch (mask_kind) {
    case kMaskNone: return finish1<kMaskNone, EXPC>(p, mode, s);
    case kMaskBits: return finish1<kMaskBits, EXPC>(p, mode, s);
    case kMaskF32: return finish1<kMaskF32, EXPC>(p, mode, s);
  }
  return hipErrorInvalidValue;
}

hipError_t GLB_CAT(launch_finish_, GLB_DT)(const StepParams &p, int mask_kind, int mode, int expc, hipStream_t s) {
  if constexpr (kHasHwExp) {
    if (expc == kExpHw) return finish0<kExpHw>(p, mask_kind, mode, s);
  }
  return expc == kExpPoly ? finish0<kExpPoly>(p, mask_kind, mode, s) : hipErrorInvalidValue;
}

hipError_t GLB_CAT(launch_logprob_rows_, GLB_DT)(const void *logits, int64_t ld, int V, float scale,
                                                  const float *lse, void *out, bool out16, int64_t out_ld, int n_rows,
                                                  hipStream_t s) {
  constexpr int EPV = ElemTraits<GLB_DT>::EPV;
  const int64_t total = (int64_t)n_rows * ((V + EPV - 1) / EPV);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if constexpr (GLB_DT != kDtF32) {
    if (out16) {
      hipLaunchKernelGGL((logprob_rows_kernel<GLB_DT, true>), grid, block, 0, s, logits, ld, V, scale, lse, out, out_ld, n_rows);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((logprob_rows_kernel<GLB_DT, false>), grid, block, 0, s, logits, ld, V, scale, lse, out, out_ld, n_rows);
  return hipGetLastError();
}

template <int WPS, int STORE, bool OUT16 = false, int CPW = 1>
static void lsm_waves1(const void *logits, int64_t ld, int V, int nch, float scale, void *out, int64_t out_ld,
                       float *out_lse, int n_rows, uint64_t *recs, uint32_t epoch, uint32_t *err, uint64_t spin,
                       hipStream_t s) {
  const int wpr = (nch + CPW - 1) / CPW;
  const dim3 grid((unsigned)((int64_t)n_rows * wpr)), block(64);
  if (scale != 1.0f)
    hipLaunchKernelGGL((logprob_rows_waves_kernel<GLB_DT, true, WPS, STORE, OUT16, CPW>), grid, block, 0, s, logits, ld, V, nch,
                       wpr, scale, out, out_ld, out_lse, recs, epoch, err, spin);
  else
    hipLaunchKernelGGL((logprob_rows_waves_kernel<GLB_DT, false, WPS, STORE, OUT16, CPW>), grid, block, 0, s, logits, ld, V, nch,
                       wpr, scale, out, out_ld, out_lse, recs, epoch, err, spin);
}

// variant: 0 = the product's choice; the diagnostic build passes others (waves per SIMD * 10 + store form, + 100 * (chunks
// per wave - 1))
hipError_t GLB_CAT(launch_logprob_waves_, GLB_DT)(const void *logits, int64_t ld, int V, int nch, float scale, void *out,
                                                   int64_t out_ld, float *out_lse, int n_rows, uint64_t *recs,
                                                   uint32_t epoch, uint32_t *err, uint64_t spin, bool out16, int variant,
                                                   hipStream_t s) {
#define GLB_LSM_ARGS logits, ld, V, nch, scale, out, out_ld, out_lse, n_rows, recs, epoch, err, spin, s
  if constexpr (GLB_DT == kDtF32) {
#ifdef GLB_STAMPS
    if (variant == 50) lsm_waves1<5, 0>(GLB_LSM_ARGS);
    else if (variant == 30) lsm_waves1<3, 0>(GLB_LSM_ARGS);
    else
#endif
    lsm_waves1<4, 0>(GLB_LSM_ARGS);
  } else if (out16) {
#ifdef GLB_STAMPS
    if (variant == 60) lsm_waves1<6, 0, true>(GLB_LSM_ARGS);
    else if (variant == 40) lsm_waves1<4, 0, true>(GLB_LSM_ARGS);
    else if (variant == 50) lsm_waves1<5, 0, true>(GLB_LSM_ARGS);
    else if (variant == 80) lsm_waves1<8, 0, true>(GLB_LSM_ARGS);
    else if (variant == 140) lsm_waves1<4, 0, true, 2>(GLB_LSM_ARGS);
    else if (variant == 150) lsm_waves1<5, 0, true, 2>(GLB_LSM_ARGS);
    else if (variant == 240) lsm_waves1<4, 0, true, 3>(GLB_LSM_ARGS);
    else if (variant == 330) lsm_waves1<3, 0, true, 4>(GLB_LSM_ARGS);
    else
#endif
    // chunks per wave by the row's length (same-box, bf16 -> bf16, us at 1024 x 50257 / 512 x 128256: one chunk per wave
    // 43 / 69, two 44 / 55, three 47 / 52, four 63 / 60 - profiles/r04/ab_lsm_cpw_v4.log)
    if (nch > 16) lsm_waves1<4, 0, true, 3>(GLB_LSM_ARGS);
    else lsm_waves1<8, 0, true, 1>(GLB_LSM_ARGS);
  } else {
#ifdef GLB_STAMPS
    if (variant == 50) lsm_waves1<5, 0>(GLB_LSM_ARGS);       // 107 / 141 us at 1024 x 50257 / 512 x 128256 bf16
    else if (variant == 51) lsm_waves1<5, 1>(GLB_LSM_ARGS);  //  75 / 102 (plain stores: L2 merges the halves)
    else if (variant == 42) lsm_waves1<4, 2>(GLB_LSM_ARGS);
    else if (variant == 52) lsm_waves1<5, 2>(GLB_LSM_ARGS);  //  65 /  94
    else if (variant == 142) lsm_waves1<4, 2, false, 2>(GLB_LSM_ARGS);
    else if (variant == 152) lsm_waves1<5, 2, false, 2>(GLB_LSM_ARGS);
    else if (variant == 242) lsm_waves1<4, 2, false, 3>(GLB_LSM_ARGS);
    else if (variant == 332) lsm_waves1<3, 2, false, 4>(GLB_LSM_ARGS);
    else
#endif
    // (bf16 -> float32: one chunk per wave 61 / 94, two 63 / 76, three 68 / 76, four 77 / 82)
    if (nch > 16) lsm_waves1<5, 2, false, 2>(GLB_LSM_ARGS);
    else lsm_waves1<5, 2, false, 1>(GLB_LSM_ARGS);
  }
#undef GLB_LSM_ARGS
  return hipGetLastError();
}

}  // namespace glb
