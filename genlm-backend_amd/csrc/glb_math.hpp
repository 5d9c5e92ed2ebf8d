// glb_math.hpp — "GLB math": the deterministic arithmetic contract of the hot path.
//
// Device restatement of the contract that oracle/glb_oracle.c states for the CPU.  Every floating
// point step is a single IEEE-754 operation (mul, add, fma, rint, exact power-of-two scaling), the
// file is compiled with -ffp-contract=off, and all sums are 64-bit integer sums, so results do not
// depend on how a row is split over lanes, waves, workgroups or GPUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glb {

constexpr float kLog2e = __builtin_bit_cast(float, 0x3FB8AA3Bu);  // 1.44269502
constexpr float kMagic = 12582912.0f;  // 1.5 * 2^23 (0x4B400000): x + kMagic has rint(x) in its low mantissa bits
constexpr uint32_t kMagicBits = 0x4B400000u;
constexpr int kFixShift = 18;  // S = sum floor(P * 2^32 >> (18 + N - n)), 44 fractional bits
constexpr int kFixFrac = 44;
// masked sums stay on the row's scale unless that leaves them fewer than 37 significant bits
constexpr uint64_t kLowMass = 1ull << 37;
constexpr double kLn2D = 0.693147180559945309417232121458;
constexpr float kNegInf = -__builtin_huge_valf();

// degree-5 minimax polynomial for 2^f on |f| <= 1/2 (max rel. error 1.6e-7 as evaluated in fp32 Horner
// form), coefficients times 2^30; the constant term is exactly 2^30
constexpr float kC0 = __builtin_bit_cast(float, 0x4e800000u);
constexpr float kC1 = __builtin_bit_cast(float, 0x4e317216u);
constexpr float kC2 = __builtin_bit_cast(float, 0x4d75fcd9u);
constexpr float kC3 = __builtin_bit_cast(float, 0x4c635b16u);
constexpr float kC4 = __builtin_bit_cast(float, 0x4b1e7722u);
constexpr float kC5 = __builtin_bit_cast(float, 0x49adfe07u);

// binary exponent of e^x: n = x * log2(e) rounded to the nearest integer in ONE rounding (the fma adds the
// magic constant before rounding), returned as an integer-valued float.  Valid for |x * log2 e| < 2^22.
__device__ __forceinline__ float exp_n(float x) { return __builtin_fmaf(x, kLog2e, kMagic) - kMagic; }

// e^x = 2^n * P / 2^30 ; returns n (integer valued float) and P (float in [0.7071, 1.4143] * 2^30):
// f = x * log2(e) - n in one fma (|f| <= 1/2), P = 2^30 * 2^f by Horner.  Accuracy: the product uses the fp32
// value of log2(e), i.e. e^(x (1 + 1.4e-8)) - as if the logit carried a quarter-ulp error.
__device__ __forceinline__ void exp_parts(float x, float &nf, float &P) {
  const float tm = __builtin_fmaf(x, kLog2e, kMagic);
  const float neg_n = kMagic - tm;
  nf = tm - kMagic;
  const float f = __builtin_fmaf(x, kLog2e, neg_n);
  float p = kC5;
  p = __builtin_fmaf(p, f, kC4);
  p = __builtin_fmaf(p, f, kC3);
  p = __builtin_fmaf(p, f, kC2);
  p = __builtin_fmaf(p, f, kC1);
  p = __builtin_fmaf(p, f, kC0);
  P = p;
}

// float -> uint32 as the hardware does it (v_cvt_u32_f32: NaN and negatives -> 0, >= 2^32 -> 0xffffffff); spelled
// as an instruction because a C cast of NaN is undefined and gets folded
__device__ __forceinline__ uint32_t cvt_u32_sat(float v) {
  uint32_t r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// fixed-point term of x relative to the row exponent N (N + 18 passed pre-added as Nb).
// x == -inf contributes 0 (f is NaN -> P is NaN -> pfix = 0).
__device__ __forceinline__ uint64_t fix_term_from_parts(float nf, float P, float Nb) {
  // s = min(Nb - n, 63) (NaN -> 63); P < 2^31 so a shift by 63 always yields 0
  const uint32_t s = cvt_u32_sat(fminf(Nb - nf, 63.0f));
  const uint32_t pfix = cvt_u32_sat(P);
  return ((uint64_t)pfix << 32) >> s;
}

__device__ __forceinline__ uint64_t fix_term(float x, float Nb) {
  float nf, P;
  exp_parts(x, nf, P);
  return fix_term_from_parts(nf, P, Nb);
}

// Four independent exp splits + fixed-point operands, hand-interleaved (the four dependency chains advance in
// lock step) and with the shift taken in the integer domain: n sits in the low mantissa bits of
// tm = fma(x, log2e, magic), so  sh = min(bits(Nb + magic) - bits(tm), 63)  is two integer instructions instead
// of subtract / min / convert.  Same values as exp_parts + fix_term_from_parts for |x * log2 e| < 2^22:
//   pf[c] = (uint32) P(x_c),  sh[c] = min(Nb - n(x_c), 63),  term = ((uint64)pf << 32) >> sh.
__device__ __forceinline__ void exp_fix4(float x0, float x1, float x2, float x3, float Nb,
                                         uint32_t (&pf)[4], uint32_t (&sh)[4]) {
  float r0, r1, r2, r3;
  const float c4 = kC4, l2e = kLog2e, magic = kMagic;
  const uint32_t nbi = __float_as_uint(Nb + kMagic);
  asm volatile(
      "v_fmaak_f32 %4, %18, %12, 0x4b400000\n\t"
      "v_fmaak_f32 %5, %18, %13, 0x4b400000\n\t"
      "v_fmaak_f32 %6, %18, %14, 0x4b400000\n\t"
      "v_fmaak_f32 %7, %18, %15, 0x4b400000\n\t"
      "v_sub_f32 %8, %19, %4\n\t"
      "v_sub_f32 %9, %19, %5\n\t"
      "v_sub_f32 %10, %19, %6\n\t"
      "v_sub_f32 %11, %19, %7\n\t"
      "v_fmac_f32 %8, %18, %12\n\t"
      "v_fmac_f32 %9, %18, %13\n\t"
      "v_fmac_f32 %10, %18, %14\n\t"
      "v_fmac_f32 %11, %18, %15\n\t"
      "v_fmamk_f32 %0, %8, 0x49adfe07, %17\n\t"
      "v_fmamk_f32 %1, %9, 0x49adfe07, %17\n\t"
      "v_fmamk_f32 %2, %10, 0x49adfe07, %17\n\t"
      "v_fmamk_f32 %3, %11, 0x49adfe07, %17\n\t"
      "v_fmaak_f32 %0, %0, %8, 0x4c635b16\n\t"
      "v_fmaak_f32 %1, %1, %9, 0x4c635b16\n\t"
      "v_fmaak_f32 %2, %2, %10, 0x4c635b16\n\t"
      "v_fmaak_f32 %3, %3, %11, 0x4c635b16\n\t"
      "v_fmaak_f32 %0, %0, %8, 0x4d75fcd9\n\t"
      "v_fmaak_f32 %1, %1, %9, 0x4d75fcd9\n\t"
      "v_fmaak_f32 %2, %2, %10, 0x4d75fcd9\n\t"
      "v_fmaak_f32 %3, %3, %11, 0x4d75fcd9\n\t"
      "v_fmaak_f32 %0, %0, %8, 0x4e317216\n\t"
      "v_fmaak_f32 %1, %1, %9, 0x4e317216\n\t"
      "v_fmaak_f32 %2, %2, %10, 0x4e317216\n\t"
      "v_fmaak_f32 %3, %3, %11, 0x4e317216\n\t"
      "v_fmaak_f32 %0, %0, %8, 0x4e800000\n\t"
      "v_fmaak_f32 %1, %1, %9, 0x4e800000\n\t"
      "v_fmaak_f32 %2, %2, %10, 0x4e800000\n\t"
      "v_fmaak_f32 %3, %3, %11, 0x4e800000\n\t"
      "v_sub_u32 %4, %16, %4\n\t"
      "v_sub_u32 %5, %16, %5\n\t"
      "v_sub_u32 %6, %16, %6\n\t"
      "v_sub_u32 %7, %16, %7\n\t"
      "v_min_u32 %4, 63, %4\n\t"
      "v_min_u32 %5, 63, %5\n\t"
      "v_min_u32 %6, 63, %6\n\t"
      "v_min_u32 %7, 63, %7\n\t"
      "v_cvt_u32_f32 %0, %0\n\t"
      "v_cvt_u32_f32 %1, %1\n\t"
      "v_cvt_u32_f32 %2, %2\n\t"
      "v_cvt_u32_f32 %3, %3"
      : "=&v"(pf[0]), "=&v"(pf[1]), "=&v"(pf[2]), "=&v"(pf[3]), "=&v"(sh[0]), "=&v"(sh[1]), "=&v"(sh[2]),
        "=&v"(sh[3]), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(nbi), "v"(c4), "v"(l2e), "s"(magic));  // (a literal and an SGPR cannot share an instruction on gfx9)
}

// ln(S * 2^k) for integer S > 0 (atanh series in double, fixed op order)
__device__ inline double log_fix(uint64_t S, int32_t k) {
  double d = __builtin_fma((double)(uint32_t)(S >> 32), 4294967296.0, (double)(uint32_t)S);
  uint64_t bits = (uint64_t)__double_as_longlong(d);
  int32_t e = (int32_t)((bits >> 52) & 0x7ff) - 1023;
  bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m = __longlong_as_double((long long)bits);
  if (m > 1.4142135623730951) {
    m *= 0.5;
    e += 1;
  }
  double z = (m - 1.0) / (m + 1.0);
  double w = z * z;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, w, 1.0 / 19.0);
  p = __builtin_fma(p, w, 1.0 / 17.0);
  p = __builtin_fma(p, w, 1.0 / 15.0);
  p = __builtin_fma(p, w, 1.0 / 13.0);
  p = __builtin_fma(p, w, 1.0 / 11.0);
  p = __builtin_fma(p, w, 1.0 / 9.0);
  p = __builtin_fma(p, w, 1.0 / 7.0);
  p = __builtin_fma(p, w, 1.0 / 5.0);
  p = __builtin_fma(p, w, 1.0 / 3.0);
  p = __builtin_fma(p, w, 1.0);
  double lg = (2.0 * z) * p;
  return __builtin_fma((double)(e + k), kLn2D, lg);
}

// Philox4x32-10 (Salmon et al., SC'11)
__host__ __device__ inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                              uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- wave64 helpers (DPP: no LDS traffic) ----------------------------------------------------------
// dpp_ctrl encodings (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_u64_or0(uint64_t v) {
  const uint32_t lo = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)v);
  const uint32_t hi = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// inclusive prefix sum over the 64 lanes (lane 63 ends up with the wave total)
__device__ __forceinline__ uint64_t wave_scan_u64(uint64_t v) {
  v += dpp_u64_or0<0x111, 0xf>(v);
  v += dpp_u64_or0<0x112, 0xf>(v);
  v += dpp_u64_or0<0x114, 0xf>(v);
  v += dpp_u64_or0<0x118, 0xf>(v);
  v += dpp_u64_or0<0x142, 0xa>(v);
  v += dpp_u64_or0<0x143, 0xc>(v);
  return v;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// wave total, broadcast to every lane (wave-uniform value)
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) { return readlane_u64(wave_scan_u64(v), 63); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v) {
  const uint32_t ninf = 0xff800000u;
  const float o = __uint_as_float(dpp_u32<CTRL, ROW_MASK>(ninf, __float_as_uint(v)));
  return fmaxf(v, o);
}

// wave maximum, broadcast to every lane
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max_step<0x111, 0xf>(v);
  v = dpp_max_step<0x112, 0xf>(v);
  v = dpp_max_step<0x114, 0xf>(v);
  v = dpp_max_step<0x118, 0xf>(v);
  v = dpp_max_step<0x142, 0xa>(v);
  v = dpp_max_step<0x143, 0xc>(v);
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

}  // namespace glb
