// glb_log1p.hpp — log1p(x) for x in (-1, 0] exactly as the reference's host computes it.
//
// torch's CPU `exponential_` (README.md:87 and base.py:137-141 reach it through torch.multinomial) is
// E = (float)(-log1p(-u)) with the C library's double-precision log1p.  To draw the same E on the device the function
// itself has to be the same, bit for bit: this is glibc's algorithm (sysdeps/ieee754/dbl-64/s_log1p.c - Sun's fdlibm
// log1p with the degree-7 polynomial evaluated as R1 + z^2 R2 + z^4 R3 + z^6 R4), restated for the only arguments the
// stream produces (x = -u, u = k 2^-53, 0 <= k < 2^53) as a sequence of IEEE-754 double operations: every + - * / below is
// one correctly rounded operation (the library is built with -ffp-contract=off; gfx950's v_div sequence for doubles is
// correctly rounded), so host and device agree with each other and with glibc.  Pinned on the CPU against the C
// library's log1p (tests/test_mt_cpu.py: tens of millions of arguments, every branch) and through it against torch
// (tests/test_oracle.py).
//
// The algorithm and its constants are those of fdlibm's s_log1p.c (Copyright (C) 1993 by Sun Microsystems, Inc.; developed at
// SunPro, a Sun Microsystems, Inc. business; permission to use, copy, modify, and distribute that software is freely
// granted, provided that this notice is preserved), in the evaluation order glibc gives the polynomial.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef GLB_HD
#ifdef __HIPCC__
#define GLB_HD __host__ __device__
#else
#define GLB_HD
#endif
#endif

namespace glb {

GLB_HD static inline int32_t hi_word(double x) {
  uint64_t b;
  memcpy(&b, &x, 8);
  return (int32_t)(b >> 32);
}

GLB_HD static inline double with_hi_word(double x, int32_t hi) {
  uint64_t b;
  memcpy(&b, &x, 8);
  b = (b & 0xffffffffull) | ((uint64_t)(uint32_t)hi << 32);
  memcpy(&x, &b, 8);
  return x;
}

// x in (-1, 0]
GLB_HD static inline double log1p_neg(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double f = 0.0, c = 0.0, u;
  int32_t k = 1, hu = 0;
  const int32_t hx = hi_word(x), ax = hx & 0x7fffffff;
  if (ax < 0x3e200000) {  // |x| < 2^-29
    if (ax < 0x3c900000) return x;  // |x| < 2^-54 (here: x == -0.0)
    return x - x * x * 0.5;
  }
  if (hx > 0 || hx <= (int32_t)0xbfd2bec3) {  // -0.2929 < x: no argument reduction
    k = 0;
    f = x;
    hu = 1;
  }
  if (k != 0) {
    u = 1.0 + x;
    hu = hi_word(u);
    k = (hu >> 20) - 1023;
    // (glibc: c = the rounding error of 1 + x, divided by u.  For the stream's arguments - x = -k 2^-53 - the sum 1 + x is
    // exact, c is +0 and so is c / u: the division, a dozen double-precision instructions, is left out; tests/test_mt_cpu.py
    // holds the result against the C library's bit for bit)
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = with_hi_word(u, hu | 0x3ff00000);  // u in [1, sqrt 2)
    } else {
      k += 1;
      u = with_hi_word(u, hu | 0x3fe00000);  // u / 2 in [sqrt 2 / 2, 1)
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  const double hfsq = 0.5 * f * f;
  if (hu == 0) {  // |f| < 2^-20
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    const double R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double R1 = z * Lp1, z2 = z * z;
  const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
  const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
  const double R4 = Lp6 + z * Lp7;
  const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// one Exp(1) variate of torch's CPU exponential_ from one random64() = (first word << 32) | second word
GLB_HD static inline float exponential_from_words(uint32_t first, uint32_t second) {
  const uint64_t r = ((uint64_t)first << 32) | second;
  const double u = (double)(r & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
  return (float)(-log1p_neg(-u));
}

// ---- the same float four times cheaper ---------------------------------------------------------------------------------
// Only the FLOAT (float)(-log1p(-u)) is wanted, and glibc's double is within an ulp of the true logarithm.  Any other
// double within 2^-49 of it (relative) rounds to the same float unless a rounding boundary - the midpoint of two
// neighbouring floats - lies between the two, i.e. within 2^-46 |y| of either.  exponential_fast computes log1p(-u) by a
// 128-entry table and a degree-9 series (20 double-precision operations where fdlibm's division and its polynomial take
// 60; its error is the table's: 2^-46.5 |y| at worst, 2^-52 typically), rounds it, and REPORTS when the double lies within
// 2^-41 |y| of a boundary - 45 times the bound: one value in 10^5 -, for which the caller runs log1p_neg itself.  Every
// float it does not report is the float of the function above (tests/test_mt_cpu.py: 10^8 arguments against the C library).
//     w = 1 - u (exact) = m 2^e, m in [1, 2);  i = the top seven bits of m's fraction;  1 / c_i ~ inv[i] (c_0 = 1)
//     r = m inv[i] - 1 (one fma; |r| <= 2^-8);  log w = e ln2 + log(1 / inv[i]) + log1p(r)
//     u < 2^-7: r = -u, e = 0, log(1 / inv) = 0 (w next to 1: the table form would cancel)
constexpr int kLog1pEntries = 128;

// entry i of the table: inv = 1 / c_i rounded (c_i = 1 + (i + 1/2) / 128 the middle of the i-th interval; c_0 = 1), and
// hi = log(1 / inv) by log1p_neg itself (an ulp: 2^-53.5 absolute, 2^-46.5 of a result that is at least 2^-7) - host and
// device build the same table with the same operations
GLB_HD static inline void log1p_table_entry(int i, double *inv, double *hi) {
  if (i == 0) {
    *inv = 1.0, *hi = 0.0;
    return;
  }
  const double c = 1.0 + ((double)i + 0.5) * 0.0078125;
  *inv = 1.0 / c;
  *hi = -log1p_neg(*inv - 1.0);  // (inv - 1 in (-1/2, 0): exact)
}

// y ~ log1p(-u) for u = k 2^-53, 0 < k < 2^53
GLB_HD static inline double log1p_neg_fast(double u, const double *inv, const double *hi) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double w = 1.0 - u;  // (exact)
  uint64_t b;
  memcpy(&b, &w, 8);
  const int i = (int)((b >> 45) & 127);
  const uint64_t mb = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double m;
  memcpy(&m, &mb, 8);
  const bool small = u < 0.0078125;
  const double r = small ? -u : __builtin_fma(m, inv[i], -1.0);
  const double e = small ? 0.0 : (double)((int)(b >> 52) - 1023);
  const double lh = small ? 0.0 : hi[i];
  double p = 1.0 / 9.0;
  p = __builtin_fma(p, r, -1.0 / 8.0);
  p = __builtin_fma(p, r, 1.0 / 7.0);
  p = __builtin_fma(p, r, -1.0 / 6.0);
  p = __builtin_fma(p, r, 1.0 / 5.0);
  p = __builtin_fma(p, r, -1.0 / 4.0);
  p = __builtin_fma(p, r, 1.0 / 3.0);
  p = __builtin_fma(p, r, -0.5);
  const double tail = __builtin_fma(p, r * r, e * ln2_lo);
  const double head = __builtin_fma(e, ln2_hi, lh);
  return head + (r + tail);
}

// the variate of exponential_from_words, or *redo = true when only log1p_neg can tell (then the return value is a guess)
GLB_HD static inline float exponential_fast(uint32_t first, uint32_t second, const double *inv, const double *hi, bool *redo) {
  const uint64_t k = (((uint64_t)first << 32) | second) & ((1ull << 53) - 1);
  if (k == 0) {  // u = 0: -log1p(-0.0) = +0.0
    *redo = false;
    return 0.0f;
  }
  const double u = (double)k * (1.0 / 9007199254740992.0);
  const double E = -log1p_neg_fast(u, inv, hi);  // in [2^-53, 37]: a normal float
  const float f = (float)E;
  uint32_t fb;
  memcpy(&fb, &f, 4);
  const uint32_t lb = fb - 1u, ub = fb + 1u;  // the neighbouring floats (f > 0)
  float fl, fu;
  memcpy(&fl, &lb, 4);
  memcpy(&fu, &ub, 4);
  const double lower = 0.5 * ((double)f + (double)fl), upper = 0.5 * ((double)f + (double)fu);  // the rounding boundaries round f
  const double room = (E - lower) < (upper - E) ? (E - lower) : (upper - E);
  *redo = !(room >= E * 4.547473508864641e-13);  // 2^-41 (NaN: redo)
  return f;
}

}  // namespace glb
