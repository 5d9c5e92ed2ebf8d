/*
 * glb_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the genlm-backend hot path (reference = genlm/genlm-backend) used as
 * the checker for the HIP library.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object; nothing under genlm-backend_amd/ does.
 *
 * Two layers live here:
 *   (A) reference semantics in the reference's own arithmetic (float32 torch-CPU style), each
 *       function citing the reference file:line it follows.  Pinned by the npz files under tests/golden,
 *       which were produced by the reference itself / torch-CPU (oracle/make_goldens.py).
 *   (B) the "GLB math" contract the HIP kernels implement: the same quantities
 *       computed with a fixed polynomial exp + 64-bit fixed-point sums so that results are
 *       bit-identical on any hardware.  Layer B is checked against layer A within 1e-4
 *       (tests/test_oracle.py) and the HIP path is checked against layer B bit-for-bit.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_F32 0
#define ORC_BF16 1
#define ORC_F16 2
#define ORC_MASK_NONE 0
#define ORC_MASK_BITS 1
#define ORC_MASK_F32 2
#define ORC_RNG_NONE 0
#define ORC_RNG_PHILOX 1
#define ORC_RNG_NOISE 2

/* ---------------------------------------------------------------- element loads */

static float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h >> 15) << 31;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) {
      u = sign;
    } else { /* subnormal half: value = m * 2^-24, exact in float */
      float f = (float)m * 5.9604644775390625e-08f;
      memcpy(&u, &f, 4);
      u |= sign;
    }
  } else if (e == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((e + 112) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static float load_elem(const void *base, int dtype, int64_t idx) {
  if (dtype == ORC_F32) return ((const float *)base)[idx];
  if (dtype == ORC_BF16) return bf16_to_f32(((const uint16_t *)base)[idx]);
  return f16_to_f32(((const uint16_t *)base)[idx]);
}

/* ---------------------------------------------------------------- GLB math (layer B) */

/* exp split: e^x = 2^n * 2^f,  n = x*log2e rounded to nearest in ONE rounding (the fma adds the magic
 * constant 1.5*2^23 before rounding, so n sits in the low mantissa bits of the result),  f = x*log2e - n
 * in one fma (|f| <= 1/2),  P = 2^30 * 2^f by a degree-5 minimax polynomial in Horner form with correctly
 * rounded FMAs.  Returns n (as float, integer valued) and P in [0.7071*2^30, 1.4143*2^30].
 * Valid for |x*log2e| < 2^22 (|x| < 2.9e6); the product uses the fp32 value of log2(e), i.e. the result is
 * e^(x(1+1.4e-8)). */
#define GLB_LOG2E 1.44269502162933349609375f /* 0x3FB8AA3B */
#define GLB_MAGIC 12582912.0f                /* 0x4B400000 = 1.5 * 2^23 */
static const uint32_t GLB_EXP_C[6] = {0x4e800000u, 0x4e317216u, 0x4d75fcd9u,
                                      0x4c635b16u, 0x4b1e7722u, 0x49adfe07u}; /* times 2^30 */
#define GLB_FIX_SHIFT 18 /* S has 44 fractional bits: 2^30 (Pfix) * 2^32 >> 18 */
#define GLB_FIX_FRAC 44

static float u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* binary exponent of e^x as an integer-valued float (row maxima, log-weights) */
static float glb_exp_n(float x) { return fmaf(x, GLB_LOG2E, GLB_MAGIC) - GLB_MAGIC; }

static void glb_exp_parts(float x, float *nf_out, float *P_out) {
  float tm = fmaf(x, GLB_LOG2E, GLB_MAGIC);
  float neg_n = GLB_MAGIC - tm;
  float f = fmaf(x, GLB_LOG2E, neg_n);
  float p = u2f(GLB_EXP_C[5]);
  p = fmaf(p, f, u2f(GLB_EXP_C[4]));
  p = fmaf(p, f, u2f(GLB_EXP_C[3]));
  p = fmaf(p, f, u2f(GLB_EXP_C[2]));
  p = fmaf(p, f, u2f(GLB_EXP_C[1]));
  p = fmaf(p, f, u2f(GLB_EXP_C[0]));
  *nf_out = tm - GLB_MAGIC;
  *P_out = p;
}

/* float -> uint32 as v_cvt_u32_f32 does it: NaN and negatives -> 0, >= 2^32 -> 0xffffffff, else truncate */
static uint32_t cvt_u32_sat(float v) {
  if (!(v > 0.0f)) return 0u;
  if (v >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)v;
}

/* fixed-point term of element x relative to row exponent Nf: floor(P*2^32 / 2^(18 + N - n)) */
static uint64_t glb_fix_term(float x, float Nf) {
  float nf, P;
  if (!(x > -INFINITY)) return 0; /* -inf (and NaN) contribute nothing */
  glb_exp_parts(x, &nf, &P);
  float sf = (Nf + (float)GLB_FIX_SHIFT) - nf;
  uint32_t s = cvt_u32_sat(fminf(sf, 63.0f)); /* fminf(NaN, 63) = 63 */
  uint32_t pfix = cvt_u32_sat(P);
  return ((uint64_t)pfix << 32) >> s;
}

/* natural log of S * 2^k for integer S > 0, in double, fixed op order (atanh series) */
#define GLB_LN2_D 0.693147180559945309417232121458
static double glb_log_fix(uint64_t S, int32_t k) {
  double d = fma((double)(uint32_t)(S >> 32), 4294967296.0, (double)(uint32_t)S);
  uint64_t bits;
  memcpy(&bits, &d, 8);
  int32_t e = (int32_t)((bits >> 52) & 0x7ff) - 1023;
  bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m;
  memcpy(&m, &bits, 8);
  if (m > 1.4142135623730951) {
    m *= 0.5;
    e += 1;
  }
  double z = (m - 1.0) / (m + 1.0);
  double w = z * z;
  double p = 1.0 / 21.0;
  p = fma(p, w, 1.0 / 19.0);
  p = fma(p, w, 1.0 / 17.0);
  p = fma(p, w, 1.0 / 15.0);
  p = fma(p, w, 1.0 / 13.0);
  p = fma(p, w, 1.0 / 11.0);
  p = fma(p, w, 1.0 / 9.0);
  p = fma(p, w, 1.0 / 7.0);
  p = fma(p, w, 1.0 / 5.0);
  p = fma(p, w, 1.0 / 3.0);
  p = fma(p, w, 1.0);
  double lg = (2.0 * z) * p;
  return fma((double)(e + k), GLB_LN2_D, lg);
}

/* Philox4x32-10 (Salmon et al. 2011), the counter-based generator of GLB_RNG_PHILOX */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
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

static uint64_t mulhi64(uint64_t a, uint64_t b) {
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

static int mask_allows(const uint32_t *bits, int64_t j) {
  return (int)((bits[j >> 5] >> (j & 31)) & 1u);
}

/* ---------------------------------------------------------------- chunked contract of the particle step
 *
 * (device side: genlm-backend_amd/csrc/glb_math.hpp "chunk_term" and glb_chunk.hpp.)
 * A row is cut into chunks of 4096 elements by vocabulary index.  Chunk c has its own binary scale
 * N_c = glb_exp_n(max of the chunk).  Every element gives
 *     t = ldexpf(P, n - N_c)     P = 2^(f-1) by a degree-4 Horner polynomial (GLB_EXP2_D), clamped to [0, 1]
 *                                with NaN -> 0
 * A chunk is held by one wave the way the kernels load it: with EPV elements per 16-byte vector of the logits (4 for
 * fp32, 8 for 16-bit types) lane l owns elements (i * 64 + l) * EPV + k of the chunk, vector i = 0 .. 64/EPV - 1,
 * component k = 0 .. EPV - 1.  The 64 elements of a lane fall into four CLASSES w = i mod 4 of 16 elements each.
 * The unit of floating-point summation is one (lane, class): its 16 terms are added in float32, round to nearest, one
 * by one in (i, k) order, starting from +0 (masked sums skip forbidden elements, which is the same as adding +0):
 *     P_(l,w) = (((t_0 + t_1) + t_2) + ...)          q_(l,w) = floor(P_(l,w) * 2^36)
 * Everything above that level is integer: S_c = sum of the 256 q's of the chunk; S_c^m likewise over the allowed
 * elements (on N_c, or on the allowed maximum's own scale for a low-mass chunk: glb_chunk_masked).  The row scale is
 * N = max N_c over the chunks with a non-zero sum, and the row sums are S = sum_c (S_c >> (N - N_c)); likewise for
 * the allowed sums.  sum_j e^(x_j) = 2^(N+1-36) S.  Philox draws take two stages (chunk, then lane / class / element
 * inside the chunk): orc_step.
 */
/* 2^(f-1) on |f| <= 1/2: degree-4 minimax polynomial (relative error; max 2.7e-6 as evaluated in fp32 Horner form):
 * coefficients of 2^f with the exponent lowered by one */
static const uint32_t GLB_EXP2_D[5] = {0x3f7ffff4u - (1u << 23), 0x3f31706eu - (1u << 23), 0x3e76036du - (1u << 23),
                                       0x3d650a20u - (1u << 23), 0x3c1ccbebu - (1u << 23)};
#define GLB_CHUNK 4096
#define GLB_FRAC 36
#define GLB_LOW_MASS_BITS 32 /* a chunk's bit-masked sum below 2^32 on the chunk's scale is redone on its own scale */

/* The second contract of a term (include/glb.h GLB_STEP_HW_EXP, 16-bit rows): t = 2^(fma(x, log2 e, -(N + 1))) by the
 * GPU's v_exp_f32, clamped to [0, 1] with NaN -> 0.  v_exp_f32 is within one ulp of 2^y but not correctly rounded, so
 * it has no bit-exact CPU restatement; the oracle states that contract with libm's exp2f (also within one ulp) and the
 * tests compare by tolerance and by the draws' distance from the CDF's boundaries (orc_step2's out_edge).  Selected per
 * call by orc_step2 (not thread safe: the oracle is called from one thread). */
static int g_expc = 0;
static float glb_bias(float N) { return g_expc ? -1.0f - N : GLB_MAGIC - N; }

static float glb_chunk_term(float x, float magicN) {
  if (g_expc) {
    float r = exp2f(fmaf(x, GLB_LOG2E, magicN));
    if (!(r > 0.0f)) r = 0.0f;
    if (r > 1.0f) r = 1.0f;
    return r;
  }
  float tm = fmaf(x, GLB_LOG2E, magicN);
  float negn = magicN - tm;
  uint32_t tb;
  memcpy(&tb, &tm, 4);
  int32_t np = (int32_t)(tb - 0x4B400000u);
  float f = fmaf(x, GLB_LOG2E, negn);
  float p = u2f(GLB_EXP2_D[4]);
  p = fmaf(p, f, u2f(GLB_EXP2_D[3]));
  p = fmaf(p, f, u2f(GLB_EXP2_D[2]));
  p = fmaf(p, f, u2f(GLB_EXP2_D[1]));
  p = fmaf(p, f, u2f(GLB_EXP2_D[0]));
  if (!(p > 0.0f)) p = 0.0f; /* clamp modifier: NaN and negatives -> 0 */
  if (p > 1.0f) p = 1.0f;
  if (np < -300) np = -300; /* ldexpf takes any int; keep it in range for the libm call (result is 0 either way) */
  if (np > 300) np = 300;
  return ldexpf(p, np);
}

/* floor(P * 2^36) of a float32 partial sum (P < 16: sixteen terms of at most 0.7072 each) */
static uint64_t glb_partial_q(float P) {
  if (!(P > 0.0f)) return 0;
  return (uint64_t)floor(ldexp((double)P, GLB_FRAC));
}

static int glb_epv(int dtype) { return dtype == ORC_F32 ? 4 : 8; }

/* index in the row of element (lane l, class w, position pos) of the chunk starting at lo */
static int64_t glb_lane_elem(int64_t lo, int epv, int l, int w, int pos) {
  int i = (pos / epv) * 4 + w, k = pos % epv;
  return lo + ((int64_t)i * 64 + l) * epv + k;
}

/* the 256 (lane, class) sums of the chunk starting at lo: q[l * 4 + w]; returns their total.  mb != NULL: only the
 * elements the bit mask allows. */
static uint64_t glb_chunk_sum(const float *y, int64_t lo, int64_t V, float magicN, int epv, const uint32_t *mb,
                              uint64_t *q /* [256] or NULL */) {
  uint64_t S = 0;
  for (int l = 0; l < 64; ++l)
    for (int w = 0; w < 4; ++w) {
      float P = 0.0f;
      for (int pos = 0; pos < 16; ++pos) {
        int64_t j = glb_lane_elem(lo, epv, l, w, pos);
        if (j >= V) continue;
        if (mb && !mask_allows(mb, j)) continue;
        P = P + glb_chunk_term(y[j], magicN);
      }
      uint64_t ql = glb_partial_q(P);
      if (q) q[l * 4 + w] = ql;
      S += ql;
    }
  return S;
}

typedef struct {
  float N;      /* chunk scale (or -inf) */
  uint64_t S;   /* chunk sum */
} chunk_stat;

/* statistics of y[0..V) by chunks; returns the number of chunks */
static int64_t glb_chunk_stats(const float *y, int64_t V, int epv, chunk_stat *out) {
  int64_t nch = (V + GLB_CHUNK - 1) / GLB_CHUNK;
  for (int64_t c = 0; c < nch; ++c) {
    int64_t lo = c * GLB_CHUNK, hi = lo + GLB_CHUNK < V ? lo + GLB_CHUNK : V;
    float m = -INFINITY;
    for (int64_t j = lo; j < hi; ++j)
      if (y[j] > m) m = y[j];
    out[c].N = glb_exp_n(m);
    out[c].S = glb_chunk_sum(y, lo, V, glb_bias(out[c].N), epv, NULL, NULL);
  }
  return nch;
}

/* bit-masked chunk sums: the same terms as the unmasked sum (chunk scale of x, so one exponential per element serves
 * both), allowed elements only - unless that leaves the sum of a chunk that allows anything at all below 2^32 (allowed
 * mass under about 2^-3.5 of the chunk's largest term): such a chunk sums its allowed values again on their own
 * maximum's scale.  out[c] = the
 * (scale, sum) of the allowed part of chunk c. */
static void glb_chunk_masked(const float *x, const uint32_t *mb, int64_t V, int epv, const chunk_stat *cs,
                             chunk_stat *out) {
  int64_t nch = (V + GLB_CHUNK - 1) / GLB_CHUNK;
  for (int64_t c = 0; c < nch; ++c) {
    int64_t lo = c * GLB_CHUNK, hi = lo + GLB_CHUNK < V ? lo + GLB_CHUNK : V;
    int any = 0;
    for (int64_t j = lo; j < hi && !any; ++j) any = mask_allows(mb, j);
    uint64_t S = glb_chunk_sum(x, lo, V, glb_bias(cs[c].N), epv, mb, NULL);
    out[c].N = cs[c].N;
    if (any && (S >> GLB_LOW_MASS_BITS) == 0) {
      float mk = -INFINITY;
      for (int64_t j = lo; j < hi; ++j)
        if (mask_allows(mb, j) && x[j] > mk) mk = x[j];
      out[c].N = glb_exp_n(mk);
      S = glb_chunk_sum(x, lo, V, glb_bias(out[c].N), epv, mb, NULL);
    }
    out[c].S = S;
  }
}

static uint64_t shr_sat(uint64_t v, float d) { /* d >= 0, integer valued */
  return d < 64.0f ? v >> (uint32_t)d : 0;
}

/*
 * Layer B particle step (contract of glb_logprob_mask_sample, include/glb.h).
 * Mirrors README.md:82-91 / cache.py:96 / base.py:136-141 semantics.
 */
/* out_edge (nullable, Philox draws): how far the draw is from the nearest boundary of the inverse CDF it walks - the
 * smaller of (target - lower end, upper end - target) of the chosen chunk's interval as a fraction of the row's allowed
 * sum (first stage) and of the chosen element's interval as a fraction of the chunk's allowed sum (second stage; the
 * lane's and the class's intervals contain it).  Two implementations whose terms differ in the last place can only
 * disagree on a token whose edge is of that order. */
static int step_impl(const void *logits, int dtype, int64_t n_rows, int64_t V, int64_t ld,
             float logit_scale, int64_t n_particles, const int32_t *row_of, int mask_kind,
             const void *mask, int64_t mask_ld, int64_t n_masks, const int32_t *mask_id,
             int rng_mode, const float *noise, int64_t noise_ld, uint64_t seed, uint64_t offset,
             int64_t particle_base, float *out_logZ, float *out_lse, int32_t *out_token, float *out_margin,
             float *out_edge) {
  int64_t nch = (V + GLB_CHUNK - 1) / GLB_CHUNK;
  const int epv = glb_epv(dtype);
  float *x = (float *)malloc(sizeof(float) * (size_t)V);
  float *y = (float *)malloc(sizeof(float) * (size_t)V);
  chunk_stat *ca = (chunk_stat *)malloc(sizeof(chunk_stat) * (size_t)nch);
  chunk_stat *cm = (chunk_stat *)malloc(sizeof(chunk_stat) * (size_t)nch);
  if (!x || !y || !ca || !cm) return 4;
  for (int64_t i = 0; i < n_particles; ++i) {
    int64_t r = row_of ? row_of[i] : i;
    if (r < 0 || r >= n_rows) { free(x); free(y); free(ca); free(cm); return 1; }
    int64_t mi = 0;
    if (mask_kind != ORC_MASK_NONE) mi = mask_id ? mask_id[i] : (n_masks == 1 ? 0 : i);
    const uint32_t *mb = mask_kind == ORC_MASK_BITS ? (const uint32_t *)mask + mi * mask_ld : NULL;
    const float *mf = mask_kind == ORC_MASK_F32 ? (const float *)mask + mi * mask_ld : NULL;
    for (int64_t j = 0; j < V; ++j) {
      float v = load_elem(logits, dtype, r * ld + j);
      v = v * logit_scale; /* x * 1.0f == x exactly */
      x[j] = v;
      float vm = v;
      if (mb) vm = mask_allows(mb, j) ? v : -INFINITY;
      if (mf) vm = v + mf[j];
      y[j] = vm;
    }
    /* ---- all elements */
    glb_chunk_stats(x, V, epv, ca);
    float N_all = -INFINITY;
    for (int64_t c = 0; c < nch; ++c)
      if (ca[c].S && ca[c].N > N_all) N_all = ca[c].N;
    uint64_t S_all = 0;
    for (int64_t c = 0; c < nch; ++c)
      if (ca[c].S) S_all += shr_sat(ca[c].S, N_all - ca[c].N);
    /* ---- allowed elements: cm[c] = (scale, sum) of the allowed part of chunk c - bit masks: glb_chunk_masked; float
     * masks: y = x + m has chunk scales of its own; no mask: the chunk itself.  Row level as for the full sum. */
    float N_msk = N_all;
    uint64_t S_msk = S_all;
    if (mb) glb_chunk_masked(x, mb, V, epv, ca, cm);
    else if (mf) glb_chunk_stats(y, V, epv, cm);
    else
      for (int64_t c = 0; c < nch; ++c) cm[c] = ca[c];
    if (mb || mf) {
      N_msk = -INFINITY;
      for (int64_t c = 0; c < nch; ++c)
        if (cm[c].S && cm[c].N > N_msk) N_msk = cm[c].N;
      S_msk = 0;
      for (int64_t c = 0; c < nch; ++c)
        if (cm[c].S) S_msk += shr_sat(cm[c].S, N_msk - cm[c].N);
    }
    double lse_all = S_all ? glb_log_fix(S_all, (int32_t)N_all + 1 - GLB_FRAC) : -INFINITY;
    double lse_mask = S_msk ? glb_log_fix(S_msk, (int32_t)N_msk + 1 - GLB_FRAC) : -INFINITY;
    if (out_lse) out_lse[i] = (float)lse_all;
    if (out_logZ) out_logZ[i] = (float)(lse_mask - lse_all);
    if (rng_mode == ORC_RNG_NONE || !out_token) continue;
    int32_t tok = -1;
    if (out_edge) out_edge[i] = 1.0f;
    if (S_msk != 0) {
      if (rng_mode == ORC_RNG_PHILOX) {
        uint64_t gp = (uint64_t)(particle_base + i);
        uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)offset,
                           (uint32_t)(offset >> 32)};
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, rnd[4];
        orc_philox4x32_10(ctr, key, rnd);
        uint64_t R = ((uint64_t)rnd[1] << 32) | rnd[0];
        uint64_t R2 = ((uint64_t)rnd[3] << 32) | rnd[2];
        uint64_t T = mulhi64(R, S_msk); /* uniform integer in [0, S_msk) */
        double edge = 1.0;
        {
          /* two stages, two independent 64-bit draws: the chunk by the shifted chunk sums (first draw), then inside
           * the chunk an inverse CDF on the chunk's own scale against its unshifted sum (second draw), level by level
           * down the summation tree of the contract: lane (sums of the four class terms, lanes in order), class, then
           * the element by the running float32 sum of its class - the first whose floor(c * 2^36) passes what is left
           * of the target.  Every level is an exact inverse CDF of the quantities the sums were made of. */
          for (int64_t c = 0; c < nch && tok < 0; ++c) {
            if (!cm[c].S) continue;
            float d = N_msk - cm[c].N;
            uint64_t sm = shr_sat(cm[c].S, d);
            if (T < sm) {
              uint64_t q[256];
              const float magicN = glb_bias(cm[c].N);
              {
                double lo1 = (double)T, hi1 = (double)(sm - T);
                edge = (lo1 < hi1 ? lo1 : hi1) / (double)S_msk;
              }
              const int64_t lo = c * GLB_CHUNK;
              glb_chunk_sum(y, lo, V, magicN, epv, NULL, q); /* y carries the mask: forbidden = -inf = term 0 */
              uint64_t Tc = mulhi64(R2, cm[c].S), acc = 0;
              for (int l = 0; l < 64 && tok < 0; ++l) {
                uint64_t ql = q[l * 4] + q[l * 4 + 1] + q[l * 4 + 2] + q[l * 4 + 3];
                if (acc + ql <= Tc) { acc += ql; continue; }
                uint64_t Tl = Tc - acc;
                for (int w = 0; w < 4 && tok < 0; ++w) {
                  if (Tl >= q[l * 4 + w]) { Tl -= q[l * 4 + w]; continue; }
                  float P = 0.0f;
                  uint64_t qprev = 0;
                  for (int pos = 0; pos < 16; ++pos) {
                    int64_t j = glb_lane_elem(lo, epv, l, w, pos);
                    if (j >= V) continue;
                    P = P + glb_chunk_term(y[j], magicN);
                    uint64_t qcur = glb_partial_q(P);
                    if (qcur > Tl) {
                      tok = (int32_t)j;
                      double lo2 = (double)(Tl - qprev), hi2 = (double)(qcur - Tl);
                      double e2 = (lo2 < hi2 ? lo2 : hi2) / (double)cm[c].S;
                      if (e2 < edge) edge = e2;
                      break;
                    }
                    qprev = qcur;
                  }
                  break;
                }
                break;
              }
              break;
            }
            T -= sm;
          }
        }
        if (out_edge) out_edge[i] = (float)edge;
      } else { /* exponential race against caller noise: first maximum of e_j / E_j, e_j on the masked row scale */
        const float *E = noise + i * noise_ld;
        float best = -1.0f, sec = -1.0f;
        for (int64_t j = 0; j < V; ++j) {
          if (!(y[j] > -INFINITY)) continue;
          float e = glb_chunk_term(y[j], glb_bias(N_msk));
          float g = e / E[j];
          if (g > best) { sec = best; best = g; tok = (int32_t)j; }
          else if (g > sec) sec = g;
        }
        if (out_margin) out_margin[i] = sec < 0.0f ? 1.0f : (best - sec) / best;
      }
    }
    out_token[i] = tok;
  }
  free(x); free(y); free(ca); free(cm);
  return 0;
}

int orc_step(const void *logits, int dtype, int64_t n_rows, int64_t V, int64_t ld,
             float logit_scale, int64_t n_particles, const int32_t *row_of, int mask_kind,
             const void *mask, int64_t mask_ld, int64_t n_masks, const int32_t *mask_id,
             int rng_mode, const float *noise, int64_t noise_ld, uint64_t seed, uint64_t offset,
             int64_t particle_base, float *out_logZ, float *out_lse, int32_t *out_token, float *out_margin) {
  return step_impl(logits, dtype, n_rows, V, ld, logit_scale, n_particles, row_of, mask_kind, mask, mask_ld, n_masks,
                   mask_id, rng_mode, noise, noise_ld, seed, offset, particle_base, out_logZ, out_lse, out_token,
                   out_margin, NULL);
}

/* orc_step under either contract of the terms (contract: 0 = polynomial, 1 = hardware exponential restated with exp2f),
 * with the draws' edge distances */
int orc_step2(const void *logits, int dtype, int64_t n_rows, int64_t V, int64_t ld,
              float logit_scale, int64_t n_particles, const int32_t *row_of, int mask_kind,
              const void *mask, int64_t mask_ld, int64_t n_masks, const int32_t *mask_id,
              int rng_mode, const float *noise, int64_t noise_ld, uint64_t seed, uint64_t offset,
              int64_t particle_base, float *out_logZ, float *out_lse, int32_t *out_token, float *out_margin,
              int contract, float *out_edge) {
  g_expc = contract ? 1 : 0;
  int rc = step_impl(logits, dtype, n_rows, V, ld, logit_scale, n_particles, row_of, mask_kind, mask, mask_ld, n_masks,
                     mask_id, rng_mode, noise, noise_ld, seed, offset, particle_base, out_logZ, out_lse, out_token,
                     out_margin, out_edge);
  g_expc = 0;
  return rc;
}

/* contract of glb_log_softmax_rows: out = x - (float)lse, lse from the chunked integer sums */
int orc_log_softmax_rows(const void *logits, int dtype, int64_t n_rows, int64_t V, int64_t ld,
                         float logit_scale, float *out, int64_t out_ld, float *out_lse) {
  int64_t nch = (V + GLB_CHUNK - 1) / GLB_CHUNK;
  float *x = (float *)malloc(sizeof(float) * (size_t)V);
  chunk_stat *ca = (chunk_stat *)malloc(sizeof(chunk_stat) * (size_t)nch);
  if (!x || !ca) return 4;
  for (int64_t r = 0; r < n_rows; ++r) {
    for (int64_t j = 0; j < V; ++j) x[j] = load_elem(logits, dtype, r * ld + j) * logit_scale;
    glb_chunk_stats(x, V, glb_epv(dtype), ca);
    float N = -INFINITY;
    for (int64_t c = 0; c < nch; ++c)
      if (ca[c].S && ca[c].N > N) N = ca[c].N;
    uint64_t S = 0;
    for (int64_t c = 0; c < nch; ++c)
      if (ca[c].S) S += shr_sat(ca[c].S, N - ca[c].N);
    float lse = S ? (float)glb_log_fix(S, (int32_t)N + 1 - GLB_FRAC) : -INFINITY;
    if (out_lse) out_lse[r] = lse;
    if (out)
      for (int64_t j = 0; j < V; ++j) out[r * out_ld + j] = x[j] - lse;
  }
  free(x); free(ca);
  return 0;
}

/* contract of glb_mask_f32_to_bits */
int orc_mask_f32_to_bits(const float *mask, int64_t n_masks, int64_t V, int64_t mask_ld,
                         uint32_t *out_bits, int64_t bits_ld, int32_t *out_nonbinary) {
  int nb = 0;
  for (int64_t k = 0; k < n_masks; ++k) {
    for (int64_t w = 0; w < bits_ld; ++w) out_bits[k * bits_ld + w] = 0;
    for (int64_t j = 0; j < V; ++j) {
      float v = mask[k * mask_ld + j];
      if (v == 0.0f) out_bits[k * bits_ld + (j >> 5)] |= 1u << (j & 31);
      else if (!(v == -INFINITY)) nb = 1;
    }
  }
  if (out_nonbinary) *out_nonbinary = nb;
  return 0;
}

/* contract of glb_normalize_weights (README.md:108-110) in GLB math */
int orc_normalize_weights(const float *lw, int64_t n, float *out_probs, float *out_stats) {
  float m = -INFINITY;
  for (int64_t i = 0; i < n; ++i)
    if (lw[i] > m) m = lw[i];
  float N = glb_exp_n(m);
  float N2 = glb_exp_n(m + m);
  uint64_t S = 0, S2 = 0;
  for (int64_t i = 0; i < n; ++i) {
    S += glb_fix_term(lw[i], N);
    S2 += glb_fix_term(lw[i] + lw[i], N2);
  }
  double lse = S ? glb_log_fix(S, (int32_t)N - GLB_FIX_FRAC) : -INFINITY;
  double lse2 = S2 ? glb_log_fix(S2, (int32_t)N2 - GLB_FIX_FRAC) : -INFINITY;
  float lsef = (float)lse;
  if (out_stats) {
    out_stats[0] = lsef;
    /* ESS = (sum w)^2 / sum w^2 = e^(2*lse - lse2) through the same exp split */
    float d = (float)(2.0 * lse - lse2), nf, P;
    if (!(d > -INFINITY) || !(d < INFINITY)) {
      out_stats[1] = 0.0f;
    } else {
      glb_exp_parts(d, &nf, &P);
      out_stats[1] = ldexpf(P, (int)nf - 30);
    }
  }
  if (out_probs)
    for (int64_t i = 0; i < n; ++i) {
      /* probs = e^(lw - lse) through the same exp split, scaled back to float */
      float d = lw[i] - lsef;
      if (!(d > -INFINITY)) { out_probs[i] = 0.0f; continue; }
      float nf, P;
      glb_exp_parts(d, &nf, &P);
      out_probs[i] = (nf < -120.0f) ? 0.0f : ldexpf(P, (int)nf - 30);
    }
  return 0;
}

/* contract of glb_resample_systematic: exact integer comb over the fixed-point weights (the
 * reference itself ends at the normalised weights, README.md:108-110) */
int orc_resample_systematic(const float *lw, int64_t n, uint64_t seed, uint64_t offset, int32_t *anc,
                            float *out_stats) {
  float m = -INFINITY;
  for (int64_t i = 0; i < n; ++i)
    if (lw[i] > m) m = lw[i];
  float N = glb_exp_n(m);
  uint64_t *cum = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  if (!cum) return 4;
  uint64_t S = 0;
  for (int64_t i = 0; i < n; ++i) {
    S += glb_fix_term(lw[i], N);
    cum[i] = S;
  }
  if (out_stats) out_stats[0] = S ? (float)glb_log_fix(S, (int32_t)N - GLB_FIX_FRAC) : -INFINITY;
  if (S == 0) {
    for (int64_t k = 0; k < n; ++k) anc[k] = (int32_t)k;
    free(cum);
    return 0;
  }
  uint32_t ctr[4] = {0xa5c3u, 0x5e5au, (uint32_t)offset, (uint32_t)(offset >> 32)};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, rnd[4];
  orc_philox4x32_10(ctr, key, rnd);
  uint64_t R = ((uint64_t)rnd[1] << 32) | rnd[0];
  uint64_t U0 = mulhi64(R, S), un = (uint64_t)n, a = S / un, b = S % un;
  int64_t i = 0;
  for (int64_t k = 0; k < n; ++k) { /* T_k is non-decreasing in k: one forward sweep */
    uint64_t Tk = (uint64_t)k * a + (U0 + (uint64_t)k * b) / un;
    while (cum[i] <= Tk) ++i;
    anc[k] = (int32_t)i;
  }
  free(cum);
  return 0;
}

/* contract of glb_trie_reduce = trie/base.py:346-393 (the numba loops): leaves take the token weights, internal nodes
 * the sum (or the maximum, starting from 0) of their children in ascending child order, accumulated in double and
 * stored as float32 per node (the reference keeps doubles throughout) */
int orc_trie_reduce(const float *ws, int64_t ld, int64_t n_rows, int64_t V, int64_t n_nodes, int64_t n_levels,
                    const int32_t *leaf_node, const int32_t *level_start, const int32_t *level_nodes,
                    const int32_t *child_ptr, const int32_t *child_idx, int op, int from_logprobs, float *out,
                    int64_t out_ld) {
  for (int64_t r = 0; r < n_rows; ++r) {
    float *o = out + r * out_ld;
    for (int64_t i = 0; i < n_nodes; ++i) o[i] = 0.0f;
    for (int64_t k = 0; k < V; ++k) {
      float v = ws[r * ld + k];
      o[leaf_node[k]] = from_logprobs ? expf(v) : v;
    }
    for (int64_t d = 0; d < n_levels; ++d)
      for (int32_t i = level_start[d]; i < level_start[d + 1]; ++i) {
        int32_t node = level_nodes[i];
        double acc = 0.0;
        for (int32_t c = child_ptr[node]; c < child_ptr[node + 1]; ++c) {
          double v = (double)o[child_idx[c]];
          if (op == 0) acc += v;
          else if (v > acc) acc = v;
        }
        o[node] = (float)acc;
      }
  }
  return 0;
}

/* ---------------------------------------------------------------- layer A: reference semantics */

/* MT19937 as used by torch's CPUGeneratorImpl (at::mt19937; the reference reaches it through
 * torch.multinomial in README.md:87 and base.py:137-141).  The algorithm is Matsumoto &
 * Nishimura's published one; torch seeds it with init_genrand(seed & 0xffffffff). */
typedef struct {
  uint32_t mt[624];
  int32_t idx;
} orc_mt19937;

void orc_mt19937_seed(orc_mt19937 *st, uint64_t seed) {
  st->mt[0] = (uint32_t)seed;
  for (int i = 1; i < 624; ++i)
    st->mt[i] = 1812433253u * (st->mt[i - 1] ^ (st->mt[i - 1] >> 30)) + (uint32_t)i;
  st->idx = 624;
}

static uint32_t orc_mt19937_next(orc_mt19937 *st) {
  if (st->idx >= 624) {
    uint32_t *mt = st->mt;
    for (int i = 0; i < 624; ++i) {
      uint32_t yv = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
      mt[i] = mt[(i + 397) % 624] ^ (yv >> 1) ^ ((yv & 1u) ? 0x9908b0dfu : 0u);
    }
    st->idx = 0;
  }
  uint32_t yv = st->mt[st->idx++];
  yv ^= yv >> 11;
  yv ^= (yv << 7) & 0x9d2c5680u;
  yv ^= (yv << 15) & 0xefc60000u;
  yv ^= yv >> 18;
  return yv;
}

/* torch.empty(n, dtype=float32).exponential_(1, generator) on CPU: each variate consumes one
 * random64() = (first word << 32 | second word), u = (r & (2^53-1)) * 2^-53 as double,
 * E = (float)(-log1p(-u)).  Verified bit-for-bit against torch 2.10 in tests/test_oracle.py. */
int orc_mt19937_exponential_f32(orc_mt19937 *st, float *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    uint64_t hi = orc_mt19937_next(st), lo = orc_mt19937_next(st);
    uint64_t r = (hi << 32) | lo;
    double u = (double)(r & ((1ULL << 53) - 1)) * (1.0 / 9007199254740992.0);
    out[i] = (float)(-log1p(-u));
  }
  return 0;
}

/* README.md:84-87 particle math in the reference's float32 arithmetic, given logps already
 * normalised (the tensor next_token_logprobs returns):
 *   masked = logps + mask; logZ = logsumexp(masked); p = exp(masked - logZ);
 *   token = multinomial(p, 1) == first argmax p / E   (torch CPU fast path, n_sample == 1).
 * logsumexp follows torch: max, exp(x - max) summed, log, + max (sum done in double here; torch
 * sums in float with its own blocking, the difference is < 1e-6 and is covered by the 1e-4 bar). */
int orc_ref_particle(const float *logps, const float *mask, int64_t V, const float *E,
                     float *out_logZ, int32_t *out_token) {
  float m = -INFINITY;
  for (int64_t j = 0; j < V; ++j) {
    float v = mask ? logps[j] + mask[j] : logps[j];
    if (v > m) m = v;
  }
  float mm = isinf(m) ? 0.0f : m;
  double s = 0;
  for (int64_t j = 0; j < V; ++j) {
    float v = mask ? logps[j] + mask[j] : logps[j];
    s += (double)expf(v - mm);
  }
  float logZ = (float)log(s) + mm;
  if (out_logZ) *out_logZ = logZ;
  if (E && out_token) {
    float best = -1.0f;
    int32_t tok = -1;
    for (int64_t j = 0; j < V; ++j) {
      float v = mask ? logps[j] + mask[j] : logps[j];
      float p = expf(v - logZ);
      float g = p / E[j];
      if (g > best) { best = g; tok = (int32_t)j; }
    }
    *out_token = tok;
  }
  return 0;
}

/* cache.py:96 — torch.log_softmax(row, 0) in float32: x - max - log(sum exp(x - max)) */
int orc_ref_log_softmax(const float *x, int64_t V, float *out) {
  float m = -INFINITY;
  for (int64_t j = 0; j < V; ++j)
    if (x[j] > m) m = x[j];
  double s = 0;
  for (int64_t j = 0; j < V; ++j) s += (double)expf(x[j] - m);
  float ls = (float)log(s);
  for (int64_t j = 0; j < V; ++j) out[j] = (x[j] - m) - ls;
  return 0;
}

/* hf.py:214-220 — dedup by tuple(prompt), groups numbered in first-appearance order (dict order) */
int orc_group_contexts(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                       int64_t n, int32_t *out_group_of, int32_t *out_rep, int32_t *out_n_groups) {
  int32_t ng = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t li = lengths[i];
    int32_t g = -1;
    for (int32_t k = 0; k < ng; ++k) {
      int64_t r = out_rep[k];
      if (lengths[r] == li &&
          memcmp(tokens + starts[r], tokens + starts[i], (size_t)li * 4) == 0) {
        g = k;
        break;
      }
    }
    if (g < 0) {
      g = ng;
      out_rep[ng++] = (int32_t)i;
    }
    out_group_of[i] = g;
  }
  *out_n_groups = ng;
  return 0;
}

/* hf.py:314-344 walk_cache, restricted to what batching needs: deepest KV-bearing prefix that is
 * a proper prefix of the context (KV is looked up *before* consuming the token at that depth,
 * so a prefix as long as the whole context never qualifies). */
int orc_match_prefixes(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                       int64_t n, const int32_t *ptok, const int64_t *pst, const int32_t *plen,
                       int64_t np, int32_t *out_prefix, int32_t *out_base) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t li = lengths[i];
    int32_t best = -1;
    int64_t bl = 0;
    for (int64_t k = 0; k < np; ++k) {
      int64_t lk = plen[k];
      if (lk >= li || lk <= bl) continue;
      if (lk > 0 && memcmp(ptok + pst[k], tokens + starts[i], (size_t)lk * 4) == 0) {
        best = (int32_t)k;
        bl = lk;
      }
    }
    out_prefix[i] = best;
    out_base[i] = (int32_t)bl;
  }
  return 0;
}

/* hf.py:55-70,232-246 — Query.prompt_padded / attention_mask / position_ids for a batch */
int orc_gather_padded(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                      const int32_t *sel, int64_t n_sel, const int32_t *base, int64_t pad_id, int64_t p_max,
                      int64_t l_max, int64_t *ids, int64_t *am, int64_t *pos, int32_t *last) {
  for (int64_t u = 0; u < n_sel; ++u) {
    int64_t s = sel ? sel[u] : u;
    int64_t b = base ? base[s] : 0;
    int64_t len = (int64_t)lengths[s] - b;
    if (len < 0 || len > l_max || b > p_max) return 1;
    for (int64_t t = 0; t < l_max; ++t) {
      ids[u * l_max + t] = t < len ? tokens[starts[s] + b + t] : pad_id;
      pos[u * l_max + t] = t < len ? b + t : 0;
    }
    for (int64_t p = 0; p < p_max + l_max; ++p) {
      int64_t v;
      if (p < p_max) v = p < b;
      else v = (p - p_max) < len;
      am[u * (p_max + l_max) + p] = v;
    }
    if (last) last[u] = (int32_t)(len - 1);
  }
  return 0;
}

/* hf.py:33-53,247-271 — zero-pad each query's past on the sequence axis and stack on batch */
int orc_gather_kv_padded(const void *const *slabs, const int32_t *slab_len, int64_t np,
                         const int32_t *prefix_of, int64_t n_rows, int64_t heads, int64_t hd,
                         int64_t p_max, int32_t eb, void *out) {
  (void)np;
  char *o = (char *)out;
  size_t rowb = (size_t)hd * eb;
  for (int64_t u = 0; u < n_rows; ++u) {
    int32_t k = prefix_of[u];
    for (int64_t h = 0; h < heads; ++h)
      for (int64_t p = 0; p < p_max; ++p) {
        char *dst = o + ((u * heads + h) * p_max + p) * rowb;
        if (k >= 0 && p < slab_len[k])
          memcpy(dst, (const char *)slabs[k] + (h * slab_len[k] + p) * rowb, rowb);
        else
          memset(dst, 0, rowb);
      }
  }
  return 0;
}

/* README.md:82-91 bookkeeping after the draw */
int orc_particles_advance(int32_t *ctx, int64_t ctx_ld, int32_t *len, int32_t *active,
                          float *lw, const float *logZ, const int32_t *tok, int64_t n,
                          int32_t eos, int32_t max_len) {
  for (int64_t i = 0; i < n; ++i) {
    if (!active[i]) continue;
    if (tok[i] == -2) continue; /* a failed launch (include/glb.h: out_token), never a result: left as it was */
    lw[i] += logZ[i];
    if (tok[i] == eos || tok[i] < 0) {
      active[i] = 0;
    } else {
      ctx[i * ctx_ld + len[i]] = tok[i];
      len[i] += 1;
      if (len[i] >= max_len) active[i] = 0;
    }
  }
  return 0;
}
