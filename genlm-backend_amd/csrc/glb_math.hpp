// glb_math.hpp — "GLB math": the deterministic arithmetic contract of the hot path.
//
// Device restatement of the contract that oracle/glb_oracle.c states for the CPU.  Every floating
// point step is a single IEEE-754 operation (mul, add, fma, exact power-of-two scaling), the
// file is compiled with -ffp-contract=off, float32 additions happen only inside one (lane, class) of a chunk in a
// fixed order, and every sum above that level is an integer sum, so results do not depend on how rows are split
// over waves, workgroups or GPUs.
//
// Two generations live here: the chunked contract of the particle step (chunk_term and the (lane, class) partial
// sums, below) and the row-scale fixed-point term of the first round (exp_parts / fix_term), which
// glb_normalize_weights still uses for the short log-weight vector.
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

// wave maximum, broadcast to every lane.  One v_max_f32 with a DPP operand per step (a lane without a source keeps its
// value); written out because fmaxf() on a DPP move costs four instructions a step - the move, its -inf filler and a
// canonicalising max of either operand.  The input is canonicalised once (a signalling NaN would otherwise come out of
// v_max_f32 as a quiet NaN instead of being ignored), every later value is a v_max_f32 result.  s_nop 1: the two wait
// states a DPP read needs after the VALU write of its source (the compiler does not look inside an asm block).
__device__ __forceinline__ float wave_max(float v) {
  v = __builtin_canonicalizef(v);
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}


// =====================================================================================================
// Chunked contract of the particle step.  A row is cut into chunks of kChunk elements by
// vocabulary index.  Chunk c has its own binary scale N_c = exp_n(max of the chunk); every element gives
//     t = ldexp(P(f), n - N_c)    fp32, P = 2^(f-1) in [0.35, 0.71] (degree-4 Horner, result clamped to [0,1],
//                                 NaN -> 0), so t <= 0.7072 and t = 0 for -inf / NaN
// One wave holds a chunk: lane l owns elements (i * 64 + l) * EPV + k (EPV elements per 16-byte vector, vector i,
// component k).  The unit of floating-point summation is one (lane, class w = i mod 4): its 16 terms are added in
// fp32, round to nearest, one by one in (i, k) order from +0 (a masked sum skips forbidden elements = adds +0):
//     P_(l,w) = ((t_0 + t_1) + t_2) + ...        q_(l,w) = floor(P_(l,w) * 2^36) = (h << 18) + l
// Everything above is an integer sum (of the h and l words separately), so it does not matter which wave, workgroup
// or GPU holds which (lane, class) - only the order inside one is fixed, and it is the order the loads deliver.
// Three VALU instructions per element for both sums (one add, and under EXEC = allowed lanes a second one) where
// round 2's exact per-element integer terms took six.
// =====================================================================================================
constexpr int kChunk = 4096;         // elements per chunk (64 per lane of one wave)
constexpr int kFrac = 36;            // q = floor(P * 2^36)
constexpr int kGridHi = 18;          // q = (h << 18) + l
constexpr int kLowMassBits = 32;     // a chunk's bit-masked sum below 2^32 (on the chunk's scale) is redone on its own scale
// 2^(f-1) on |f| <= 1/2: degree-4 minimax polynomial (relative error, Remez; max 2.7e-6 as evaluated in fp32
// Horner form, mean +4e-8), coefficients of 2^f with the exponent lowered by one
constexpr float kD0 = __builtin_bit_cast(float, 0x3f7ffff4u - (1u << 23));
constexpr float kD1 = __builtin_bit_cast(float, 0x3f31706eu - (1u << 23));
constexpr float kD2 = __builtin_bit_cast(float, 0x3e76036du - (1u << 23));
constexpr float kD3 = __builtin_bit_cast(float, 0x3d650a20u - (1u << 23));
constexpr float kD4 = __builtin_bit_cast(float, 0x3c1ccbebu - (1u << 23));

// v_fma_f32 with the clamp modifier: result clamped to [0, 1], NaN -> 0 (DX10_CLAMP is on in HSA kernels)
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The two arithmetic contracts of a term.  kExpPoly (every element type): the polynomial above - every
// step an IEEE operation, restated bit for bit by oracle/glb_oracle.c.  kExpHw (16-bit rows, glb_step_args.flags &
// GLB_STEP_HW_EXP): t = v_exp_f32(fma(x, log2 e, -(N_c + 1))) - the hardware's 2^y, within one ulp of the true value
// (more accurate than the polynomial's 2.7e-6) and deterministic on gfx950, but not correctly rounded, so no CPU
// restatement reproduces its last bit: that contract is checked against the oracle's exp2f form by tolerance (logZ, lse)
// and token by token with the draws' distances from the CDF's boundaries (tests/test_step_gpu.py).  Everything above the
// term - chunks, (lane, class) sums in load order, integer sums, records, draws - is the same code.
enum { kExpPoly = 0, kExpHw = 1 };

// what chunk_term adds to x * log2(e): kMagic - N (exact; n - N lands in the low mantissa bits), or -(N + 1)
template <int EXPC>
__device__ __forceinline__ float term_bias(float N) {
  if constexpr (EXPC == kExpHw) return -1.0f - N;
  else return kMagic - N;
}

// t of one element against the chunk scale; bias = term_bias(N_c).  kExpPoly: the exponent n - N_c sits in the
// low mantissa bits of tm; out-of-range inputs (-inf, NaN, more than 2^22 binades away) end as t = 0 or as a
// harmless finite value because P is clamped and v_ldexp_f32 saturates.  kExpHw: y <= -1/2 by construction of N_c, the
// clamp modifier turns NaN (a NaN logit, -inf against an empty chunk's scale) into 0.
template <int EXPC = kExpPoly>
__device__ __forceinline__ float chunk_term(float x, float bias) {
  if constexpr (EXPC == kExpHw) {
    const float y = __builtin_fmaf(x, kLog2e, bias);
    float r;
    asm("v_exp_f32 %0, %1 clamp" : "=v"(r) : "v"(y));
    return r;
  } else {
    const float tm = __builtin_fmaf(x, kLog2e, bias);
    const float negn = bias - tm;
    const int np = (int)(__float_as_uint(tm) - kMagicBits);
    const float f = __builtin_fmaf(x, kLog2e, negn);
    float p = kD4;
    p = __builtin_fmaf(p, f, kD3);
    p = __builtin_fmaf(p, f, kD2);
    p = __builtin_fmaf(p, f, kD1);
    p = fma_clamp01(p, f, kD0);
    return __builtin_amdgcn_ldexpf(p, np);
  }
}

// floor(P * 2^36) of a (lane, class) partial sum as two words: h = floor(P * 2^18) < 2^22, l < 2^18.  Every step
// exact: P < 16, so P * 2^18 < 2^22 truncates exactly, and P - h * 2^-18 < 2^-18 is a multiple of ulp(P) >= 2^-41.
__device__ __forceinline__ void partial_q(float P, uint32_t &h, uint32_t &l) {
  const float hi = __builtin_truncf(P * 262144.0f);
  const float lo = __builtin_fmaf(hi, -0x1p-18f, P) * 0x1p36f;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(h) : "v"(hi));
  asm("v_cvt_u32_f32 %0, %1" : "=v"(l) : "v"(lo));
}

// Pm += t_k on the lanes named by the 64-bit lane masks M_k (k = 0..3, in this order: one dependent chain).  EXEC is
// saved and restored, so the block is correct under any caller EXEC (lanes outside it stay untouched: M_k is ANDed in).
__device__ __forceinline__ void masked_add4(float &Pm, float t0, float t1, float t2, float t3, uint64_t M0, uint64_t M1,
                                            uint64_t M2, uint64_t M3) {
  uint64_t saved;
  asm("s_mov_b64 %1, exec\n\t"
      "s_and_b64 exec, %1, %6\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "s_and_b64 exec, %1, %7\n\t"
      "v_add_f32 %0, %0, %3\n\t"
      "s_and_b64 exec, %1, %8\n\t"
      "v_add_f32 %0, %0, %4\n\t"
      "s_and_b64 exec, %1, %9\n\t"
      "v_add_f32 %0, %0, %5\n\t"
      "s_mov_b64 exec, %1"
      : "+v"(Pm), "=&s"(saved)
      : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(M0), "s"(M1), "s"(M2), "s"(M3)
      : "scc");
}

// zero-instruction barriers: the value is re-materialised in a VGPR here (keeps wave-uniform 64-bit compares out of
// hipcc's scalar lowering, which loses SCC on ROCm 7.2)
__device__ __forceinline__ void opaque_u32(uint32_t &r) { asm volatile("" : "+v"(r)); }

// 32-bit wave sum by DPP; every lane of the result is NOT the total - lane 63 holds it
__device__ __forceinline__ uint32_t wave_sum_u32_l63(uint32_t v) {
  v += dpp_u32<0x111, 0xf>(0u, v);
  v += dpp_u32<0x112, 0xf>(0u, v);
  v += dpp_u32<0x114, 0xf>(0u, v);
  v += dpp_u32<0x118, 0xf>(0u, v);
  v += dpp_u32<0x142, 0xa>(0u, v);
  v += dpp_u32<0x143, 0xc>(0u, v);
  return v;
}

// the first four steps of the DPP scan: inclusive sums within each 16-lane row (lanes 15, 31, 47, 63 = the rows' totals)
__device__ __forceinline__ uint32_t row_scan_u32(uint32_t v) {
  v += dpp_u32<0x111, 0xf>(0u, v);
  v += dpp_u32<0x112, 0xf>(0u, v);
  v += dpp_u32<0x114, 0xf>(0u, v);
  v += dpp_u32<0x118, 0xf>(0u, v);
  return v;
}

// its last two: a row scan taken on to the wave scan (lane 63 = the wave's total)
__device__ __forceinline__ uint32_t rows_scan_to_wave(uint32_t v) {
  v += dpp_u32<0x142, 0xa>(0u, v);
  v += dpp_u32<0x143, 0xc>(0u, v);
  return v;
}

// the four rows' totals of a row scan, added (wave-uniform)
__device__ __forceinline__ uint32_t rows_total(uint32_t rs) {
  return (uint32_t)__builtin_amdgcn_readlane((int)rs, 15) + (uint32_t)__builtin_amdgcn_readlane((int)rs, 31) +
         (uint32_t)__builtin_amdgcn_readlane((int)rs, 47) + (uint32_t)__builtin_amdgcn_readlane((int)rs, 63);
}

// sums of v over the four 16-lane rows of the wave (wave-uniform results): the first four steps of the DPP scan leave
// every row's total in its last lane - what the full scan passes through anyway
__device__ __forceinline__ void row_sums_u32(uint32_t v, uint32_t (&r)[4]) {
  v += dpp_u32<0x111, 0xf>(0u, v);
  v += dpp_u32<0x112, 0xf>(0u, v);
  v += dpp_u32<0x114, 0xf>(0u, v);
  v += dpp_u32<0x118, 0xf>(0u, v);
  r[0] = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
  r[1] = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
  r[2] = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
  r[3] = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

}  // namespace glb
