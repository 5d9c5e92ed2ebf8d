// glb_weights.hip - the population's weights (include/glb.h: glb_normalize_weights, glb_resample_systematic): README.md:108-110's
// normalisation of the gathered log-weight vector, and the replicated deterministic resampling step the reference leaves to
// its users.  One workgroup each, GLB math (round 1's row-scale fixed-point terms); moved out
// of glb_api.hip in round 5.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/glb.h"
#include "glb_common.hpp"
#include "glb_math.hpp"

namespace {

// README.md:108-110 on the gathered log-weight vector, one workgroup, GLB math
__global__ __launch_bounds__(1024) void normalize_weights_kernel(const float *lw, int64_t n,
                                                                 float *probs, float *stats) {
  using namespace glb;
  const int T = 1024, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float s_m[16];
  __shared__ uint64_t s_s[2][16];
  __shared__ float s_lse;
  float m = kNegInf;
  for (int64_t i = tid; i < n; i += T) m = fmaxf(m, lw[i]);
  m = wave_max(m);
  if (lane == 0) s_m[wave] = m;
  __syncthreads();
  m = s_m[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, s_m[w]);
  const float N = exp_n(m), N2 = exp_n(m + m);
  const float Nb = N + (float)kFixShift, Nb2 = N2 + (float)kFixShift;
  uint64_t S = 0, S2 = 0;
  for (int64_t i = tid; i < n; i += T) {
    const float v = lw[i];
    S += fix_term(v, Nb);
    S2 += fix_term(v + v, Nb2);
  }
  S = wave_sum_u64(S);
  S2 = wave_sum_u64(S2);
  if (lane == 0) {
    s_s[0][wave] = S;
    s_s[1][wave] = S2;
  }
  __syncthreads();
  if (tid == 0) {
    S = 0;
    S2 = 0;
    for (int w = 0; w < 16; ++w) {
      S += s_s[0][w];
      S2 += s_s[1][w];
    }
    const double lse = S ? log_fix(S, (int32_t)N - kFixFrac) : (double)kNegInf;
    const double lse2 = S2 ? log_fix(S2, (int32_t)N2 - kFixFrac) : (double)kNegInf;
    s_lse = (float)lse;
    if (stats) {
      stats[0] = (float)lse;
      const float d = (float)(2.0 * lse - lse2);
      if (!(d > kNegInf) || !(d < -kNegInf)) {
        stats[1] = 0.0f;
      } else {
        float nf, P;
        exp_parts(d, nf, P);
        stats[1] = __builtin_ldexpf(P, (int)nf - 30);
      }
    }
  }
  __syncthreads();
  if (probs) {
    const float lsef = s_lse;
    for (int64_t i = tid; i < n; i += T) {
      const float d = lw[i] - lsef;
      float o = 0.0f;
      if (d > kNegInf) {
        float nf, P;
        exp_parts(d, nf, P);
        o = (nf < -120.0f) ? 0.0f : __builtin_ldexpf(P, (int)nf - 30);
      }
      probs[i] = o;
    }
  }
}

// Systematic resampling on the gathered log-weight vector, one workgroup, integers only after the terms:
//   q_i = fix_term(lw_i) on the vector's scale (as glb_normalize_weights), C_i = q_0 + .. + q_i, S = C_{n-1}
//   U0 = mulhi64(R, S) with R the Philox block of (seed, offset);  S = n*a + b
//   T_k = k*a + floor((U0 + k*b) / n)          (= floor((U0 + k*S) / n): the k-th point of the comb, exactly)
//   ancestor_k = the smallest i with C_i > T_k
// Same inputs give the same ancestors on every rank (the gathered weights are bit-identical).
__global__ __launch_bounds__(1024) void resample_systematic_kernel(const float *lw, int64_t n, uint64_t seed,
                                                                    uint64_t offset, int32_t *anc, uint64_t *cum,
                                                                    float *stats) {
  using namespace glb;
  const int T = 1024, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float s_m[16];
  __shared__ uint64_t s_w[16];
  __shared__ uint64_t s_carry;
  float m = kNegInf;
  for (int64_t i = tid; i < n; i += T) m = fmaxf(m, lw[i]);
  m = wave_max(m);
  if (lane == 0) s_m[wave] = m;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  m = s_m[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, s_m[w]);
  const float N = exp_n(m), Nb = N + (float)kFixShift;
  // inclusive scan of the terms in chunks of T
  for (int64_t base = 0; base < n; base += T) {
    const int64_t i = base + tid;
    const uint64_t q = i < n ? fix_term(lw[i], Nb) : 0ull;
    const uint64_t incl = wave_scan_u64(q);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint64_t wb = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint64_t v = s_w[w];
      if (w < wave) wb += v;
      tot += v;
    }
    const uint64_t carry = s_carry;
    if (i < n) cum[i] = carry + wb + incl;
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  const uint64_t S = s_carry;
  if (stats && tid == 0) stats[0] = S ? (float)log_fix(S, (int32_t)N - kFixFrac) : kNegInf;
  if (S == 0) {  // no mass at all: identity
    for (int64_t k = tid; k < n; k += T) anc[k] = (int32_t)k;
    return;
  }
  const uint32_t ctr[4] = {0xa5c3u, 0x5e5au, (uint32_t)offset, (uint32_t)(offset >> 32)};
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t rnd[4];
  philox4x32_10(ctr, key, rnd);
  const uint64_t R = ((uint64_t)rnd[1] << 32) | rnd[0];
  const uint64_t U0 = __umul64hi(R, S);
  const uint64_t un = (uint64_t)n, a = S / un, b = S % un;
  for (int64_t k = tid; k < n; k += T) {
    const uint64_t Tk = (uint64_t)k * a + (U0 + (uint64_t)k * b) / un;
    int64_t lo = 0, hi = n - 1;  // smallest i with cum[i] > Tk (cum[n-1] = S > Tk)
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (cum[mid] > Tk) hi = mid;
      else lo = mid + 1;
    }
    anc[k] = (int32_t)lo;
  }
}

}  // namespace

extern "C" {

int glb_normalize_weights(const float *log_weights, int64_t n, float *out_probs, float *out_stats,
                          void *stream) {
  if (!log_weights || (!out_probs && !out_stats)) return glb::api_fail(GLB_EINVAL, "null pointer");
  if (n <= 0) return glb::api_fail(GLB_EINVAL, "n must be positive");
  hipLaunchKernelGGL(normalize_weights_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                     log_weights, n, out_probs, out_stats);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "normalize_weights launch");
  return GLB_OK;
}

size_t glb_resample_workspace(int64_t n) { return n > 0 ? (size_t)n * sizeof(uint64_t) : 0; }

int glb_resample_systematic(const float *log_weights, int64_t n, uint64_t seed, uint64_t offset,
                            int32_t *out_ancestors, float *out_stats, void *workspace, size_t workspace_bytes,
                            void *stream) {
  if (!log_weights || !out_ancestors || !workspace) return glb::api_fail(GLB_EINVAL, "null pointer");
  if (n <= 0 || n > (1 << 18)) return glb::api_fail(GLB_EINVAL, "n must be in [1, 262144]");
  if (workspace_bytes < glb_resample_workspace(n) || ((uintptr_t)workspace) % 8)
    return glb::api_fail(GLB_ENOSPC, "workspace too small or misaligned");
  hipLaunchKernelGGL(resample_systematic_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, log_weights, n, seed,
                     offset, out_ancestors, (uint64_t *)workspace, out_stats);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "resample launch");
  return GLB_OK;
}

}  // extern "C"
