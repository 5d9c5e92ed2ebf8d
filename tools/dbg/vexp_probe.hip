// Is v_exp_f32 correctly rounded?  Every float32 y in [-126, 1]: v_exp_f32(y) against RN(2^y) taken from a double-precision
// exp2 (ocml, < 1 ulp of double: the float rounding of it is RN(2^y) except within 2^-52 of a tie, which are counted apart).
// If the hardware function were correctly rounded over the domain the fused step needs, a CPU oracle could restate it and
// the 16-bit rows' exponential (25 of 43 cycles per element) could be one instruction.  Prints the mismatch counts.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>

__global__ void probe(uint32_t lo, uint32_t n, unsigned long long *cnt) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0, bad2 = 0, tie = 0, total = 0;
  for (uint64_t k = gid; k < n; k += stride) {
    const uint32_t bits = lo + (uint32_t)k;
    const float y = __uint_as_float(bits);
    if (!(y >= -126.0f && y <= 1.0f)) continue;
    const float hw = __builtin_amdgcn_exp2f(y);
    const double ex = exp2((double)y);
    const float rn = (float)ex;
    // distance of ex from the rounding boundary, in units of float ulp
    const double ulp = (double)rn * 5.9604644775390625e-08;  // 2^-24 relative
    const double up = (double)__uint_as_float(__float_as_uint(rn) + 1), dn = (double)__uint_as_float(__float_as_uint(rn) - 1);
    const double mid_up = 0.5 * ((double)rn + up), mid_dn = 0.5 * ((double)rn + dn);
    const bool near_tie = fabs(ex - mid_up) < ulp * 1e-6 || fabs(ex - mid_dn) < ulp * 1e-6;
    ++total;
    if (hw != rn) {
      if (near_tie) ++tie;
      else {
        ++bad;
        const int d = (int)__float_as_uint(hw) - (int)__float_as_uint(rn);
        if (d > 1 || d < -1) ++bad2;
      }
    }
  }
  atomicAdd(&cnt[0], total);
  atomicAdd(&cnt[1], bad);
  atomicAdd(&cnt[2], bad2);
  atomicAdd(&cnt[3], tie);
}

int main() {
  unsigned long long *d, h[4] = {0, 0, 0, 0};
  hipMalloc(&d, sizeof h);
  // negative floats from -126 up to -0 (bit patterns 0x80000000 .. 0xC2FC0000) and positive 0 .. 1 (0 .. 0x3F800000)
  struct { uint32_t lo, n; const char *name; } ranges[] = {{0x80000000u, 0xC2FC0000u - 0x80000000u + 1, "[-126, -0]"},
                                                           {0x00000000u, 0x3F800000u + 1, "[0, 1]"},
                                                           {0xBF000000u - (1u << 23), (1u << 24), "around -0.5 (|f| <= 1/2 zone)"}};
  for (auto &r : ranges) {
    hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, r.lo, r.n, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-32s inputs %llu  mismatches %llu (more than 1 ulp: %llu)  at ties %llu  -> %.4f %%\n", r.name, h[0], h[1], h[2], h[3],
           100.0 * (double)h[1] / (double)(h[0] ? h[0] : 1));
  }
  return 0;
}
