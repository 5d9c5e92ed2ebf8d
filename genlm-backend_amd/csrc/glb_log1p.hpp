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
    c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);  // the rounding error of 1 + x
    c /= u;
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

}  // namespace glb
