// glb_chunk.hpp — the fused particle step as a chunked streaming reduction (gfx950 / wave64).
//
// Replaces, per particle (reference = genlm/genlm-backend):
//   cache.py:96      logps  = log_softmax(logits_row)
//   README.md:84-87  masked = logps + mask; logZ = logsumexp(masked); token = multinomial(exp(masked - logZ))
//   base.py:136-141  the same draw with logits scaled by 1/temperature
//
// Work is cut by the arithmetic contract, not by the launch geometry (glb_math.hpp): a row is a
// sequence of 4096-element chunks, each with its own binary scale, summed per (lane, class) in float32 in a fixed
// order and as integers above that.  Two ROLES, normally waves of ONE launch (fused_step_kernel):
//   stats role    one WAVE per (row|particle, chunk): 16 KiB (fp32) / 8 KiB (16-bit) of the row in one burst of
//                 non-temporal global_load_dwordx4, chunk maximum by DPP, one polynomial exp per element, both sums (all /
//                 allowed) by plain float32 adds, allowed lanes selected by EXEC from a pre-transposed bit mask read
//                 through the scalar cache.  Out: one 128-byte record of twelve tagged 8-byte granules {value, epoch} -
//                 scales, the sum of all elements, the allowed sums by 16-lane row of the wave - written through to
//                 memory, fire and forget: no barrier, no fence, no wait, no cross-wave dependency.  Rows shared by
//                 several particles (dedup fan-out of hf.py:214-220,285-288) are reduced once.
//   finish role   one wave per PARTICLE, dealt at the END of the grid (inside it was measured: slower): sweeps its row's records until every tag carries
//                 this call's epoch (the data is the flag: MI355X guide, Guideline 16 form R2; bounded spin), folds them
//                 into (N, S_all, S_mask), lse / logZ by a double-precision log, picks the chunk with the first Philox
//                 word and the 16-lane row inside it (from the record) with the second, reloads that QUARTER chunk,
//                 rebuilds its (lane, class) partials bit for bit - finishing lane 16 w + l = class w of row lane l -
//                 and walks down the summation tree: lane, class, element.
// The same roles also exist as two plain launches (chunk_stats_kernel / chunk_stats_small_kernel, finish_kernel): under
// stream capture (the epoch is a launch argument), for workspaces nobody initialised, for launches of at most 512
// chunks (four waves per chunk), and for the parity-mode exponential race (four waves per particle over the whole row).
//   logprob_rows_waves_kernel   glb_log_softmax_rows (cache.py:93-98) in one launch: a wave keeps its one to three chunks
//                               in registers, meets its row-mates through the same tagged records (the row's first
//                               wave folds, the others wait for its result) and writes x - lse (float32, or
//                               rounded into the logits' own 16-bit type: cache.py:96 keeps the dtype); three launches
//                               (statistics, one wave per row, logprob_rows_kernel) serve rows of more than 64 chunks and
//                               workspaces without tags.
// A wave that gives up a wait inside a launch (kSpinTicks / glb_set_spin_limit) writes token -2 / NaN and counts in the
// workspace's error word (StepParams::err): include/glb.h, glb_workspace_check.
// mask_prepare_kernel builds the transposed masks ([mask][chunk][vector][component] 64-bit lane words).
#pragma once
#include <type_traits>

#ifndef GLB_STATS_WAVES_16
#define GLB_STATS_WAVES_16 5  // stats waves per SIMD for 2-byte elements (6: 80 registers, 42 spilled: 43 us against 34)
#endif

#include "glb_diag.hpp"
#include "glb_math.hpp"

namespace glb {

enum { kDtF32 = 0, kDtBf16 = 1, kDtF16 = 2 };
enum { kMaskNone = 0, kMaskBits = 1, kMaskF32 = 2, kMaskRaw = 3 };
// kMaskRaw (round 6; the one-launch step only): GLB_MASK_BITS rows as the caller holds them - bit j % 32 of word j / 32 -,
// read by the stats waves themselves: a lane fetches the words that hold ITS elements' bits with the chunk's loads, keeps
// them as one 64-bit value (bit i * EPV + k = element (i, k)) and ANDs every term with 0 / ~0 before the allowed sum.  No
// mask_prepare launch, no lane words written and read back: what a grammar that changes every particle's mask every step
// pays (kMaskBits' prepared lane words select by EXEC from scalar registers - two operations an element fewer - and stay the
// form for masks that are prepared once).  Same sums: a forbidden term enters as +0.
enum { kModeStats = 0, kModePhilox = 1, kModeNoise = 2 };

struct ChunkRec {  // one (row|particle, chunk) as the finish role holds it
  float Nc;            // chunk scale exp_n(max); -inf for an empty chunk
  float Nm;            // scale of the allowed sums - bit masks: Nc, or the allowed maximum's own scale for a low-mass
                       // chunk; float masks: exp_n(max(x + mask)); no mask: Nc
  uint32_t pA, pB;     // S_c   = (pA << 18) + pB     all elements
  uint32_t rA[4], rB[4];  // allowed elements on the scale Nm, by 16-lane row g of the wave that held the chunk (lanes
                          // 16 g .. 16 g + 15: a quarter of the chunk): (rA[g] << 18) + rB[g]; S_c^m = their sum
  uint32_t pAm, pBm;   // the sums of rA / rB (not stored)
};
// In memory a record is 128 bytes: twelve 8-byte granules {low word = value, high word = the call's epoch} in the
// order Nc, Nm, pA, pB, rA[0..3], rB[0..3] (+ 32 bytes of padding), each written by ONE agent-scope store and valid
// exactly when its tag equals the epoch of the call - so a reader needs no flag, no fence and no ordering between the
// granules.  The row sums cost the reducing wave nothing (its DPP sums pass through them) and let a draw re-reduce a
// quarter of a chunk instead of all of it.
constexpr int kRecWords = 16;
constexpr uint64_t kSpinTicks = 200000000ull;  // default watchdog: a waiting wave gives up after 2 s of s_memrealtime
                                               // (100 MHz): token -2 / NaN, the workspace's error word (glb_set_spin_limit)

struct StepParams {
  const void *logits;
  int64_t ld;
  int32_t V, nch;
  float scale;
  int32_t n_particles, n_pairs, n_masks;
  const int32_t *pair_row;   // [n_pairs] logits row of a reduction unit, null = identity
  const int32_t *pair_mask;  // [n_pairs] mask row, null = (n_masks == 1 ? 0 : identity)
  const int32_t *pair_of;    // [n_particles] reduction unit of a particle, null = identity
  const uint64_t *mask_t;    // transposed bit masks [n_masks][nch * 64]
  const uint64_t *mask_any;  // [n_masks][nch]: nonzero when the mask allows any token of the chunk
  const float *mask_f;       // float masks [n_masks][mask_ld]
  int64_t mask_ld;
  const uint32_t *mask_bits; // kMaskRaw: the caller's bit rows [n_masks][mask_bits_ld]
  int64_t mask_bits_ld;
  const float *noise;
  int64_t noise_ld;
  uint64_t seed, offset;
  int64_t particle_base;
  float *out_logZ, *out_lse;
  int32_t *out_token;
  float *out_margin;  // parity mode: relative gap between the two largest e_j / E_j of the race
  uint64_t *recs;     // [n_pairs][nch][kRecWords] granules
  uint32_t epoch;     // tag of this call's granules (never 0 on the fused path; 0 = tags are not looked at)
  // fused launch (one-wave workgroups): blocks [0, stats_blocks) reduce one (unit, chunk) each, then fin_blocks
  // finishing waves (wave f takes particles f, f + fin_blocks, ...)
  int32_t stats_blocks, fin_blocks;
  GLB_DIAG(DiagParams diag;)  // diagnostic build only (glb_diag.hpp)
  uint32_t *err;      // nullable: word a wave that gave up waiting adds 1 to (glb_workspace_check)
  uint64_t spin_ticks;  // the watchdog of the waits inside the launch, in s_memrealtime ticks
};

template <int DT>
struct ElemTraits {
  static constexpr int EPV = DT == kDtF32 ? 4 : 8;   // elements per 16-byte vector
  static constexpr int ES = DT == kDtF32 ? 4 : 2;
  static constexpr int NVC = 64 / EPV;               // vectors per lane per chunk
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Read-only tables addressed uniformly by a wave (mask words, index arrays): the constant address space makes hipcc
// use scalar loads (s_load -> SGPRs) although the kernel also stores to global memory.
typedef const __attribute__((address_space(4))) uint64_t *cu64_t;
typedef const __attribute__((address_space(4))) int32_t *ci32_t;
__device__ __forceinline__ cu64_t as_const(const uint64_t *q) { return (cu64_t)(uintptr_t)q; }
__device__ __forceinline__ ci32_t as_const(const int32_t *q) { return (ci32_t)(uintptr_t)q; }

__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int DT>
__device__ __forceinline__ void unpack_vec(const u32x4_t &r, float *x) {
  if constexpr (DT == kDtF32) {
    x[0] = __uint_as_float(r.x);
    x[1] = __uint_as_float(r.y);
    x[2] = __uint_as_float(r.z);
    x[3] = __uint_as_float(r.w);
  } else if constexpr (DT == kDtBf16) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  } else {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] & 0xffffu));
      x[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] >> 16));
    }
  }
}

// two float32 values as one word of 16-bit elements (round to nearest even; a NaN stays a NaN: plain casts, which
// hipcc lowers to v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950)
template <int DT>
__device__ __forceinline__ uint32_t pack16(float lo, float hi) {
  if constexpr (DT == kDtBf16) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
  } else {
    const _Float16 a = (_Float16)lo, b = (_Float16)hi;
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
  }
}

// one word of two 16-bit logits -> the word of their two log-probabilities, (x * scale) - lse each, rounded as pack16
// rounds: the pair stays a pair from the unpack to the packed conversion (two unpack operations, one packed subtraction,
// one v_cvt_pk per word - left to itself the compiler pairs elements of DIFFERENT words for the packed arithmetic and
// then spends two more operations per word putting the halves back in order)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int DT, bool SCALED>
__device__ __forceinline__ uint32_t logprob_word16(uint32_t w, float scale, float lse) {
  f32x2_t x;
  if constexpr (DT == kDtBf16) {
    x = f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
  } else {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    x = __builtin_convertvector(__builtin_bit_cast(f16x2_t, w), f32x2_t);
  }
  if constexpr (SCALED) x = x * scale;
  x = x - lse;
  if constexpr (DT == kDtBf16) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
  } else {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2_t));
  }
}

template <int DT>
__device__ __forceinline__ uint32_t neg_inf_word() {
  return DT == kDtF32 ? 0xff800000u : (DT == kDtBf16 ? 0xff80ff80u : 0xfc00fc00u);
}

// One 16-byte vector of a row starting at element e0 (element aligned only: a row pitch of 50257 leaves three rows
// in four misaligned and the unaligned dwordx4 runs at the same rate); elements at or beyond V read as -inf.
template <int DT>
__device__ __forceinline__ u32x4_t load_vec_guarded(const char *rowp, int e0, int V) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES;
  if (e0 + EPV <= V) return *reinterpret_cast<const u32x4_t *>(rowp + (int64_t)e0 * ES);
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = neg_inf_word<DT>();
  for (int k = 0; k < V - e0; ++k) {  // at most one lane of one vector per row gets here
    if constexpr (DT == kDtF32) {
      const uint32_t v = *reinterpret_cast<const uint32_t *>(rowp + (int64_t)(e0 + k) * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) w[c] = c == k ? v : w[c];
    } else {
      const uint32_t h = *reinterpret_cast<const uint16_t *>(rowp + (int64_t)(e0 + k) * 2);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c == (k >> 1)) w[c] = (k & 1) ? ((w[c] & 0xffffu) | (h << 16)) : ((w[c] & 0xffff0000u) | h);
    }
  }
  return u32x4_t{w[0], w[1], w[2], w[3]};
}

// the wave's vectors of chunk c as loaded: vector i covers elements c*4096 + (i*64 + lane)*EPV ...
template <int DT>
__device__ __forceinline__ void load_chunk_raw(const char *rowp, int e_base, int V, int lane,
                                               u32x4_t (&raw)[ElemTraits<DT>::NVC]) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  if (e_base + kChunk <= V) {  // wave-uniform: a full chunk, every load in range
    const char *q = rowp + ((int64_t)e_base + lane * EPV) * ES;
#pragma unroll
    // non-temporal: the row is read once (the draw's re-reading of one quarter chunk per particle aside) and must not
    // push what the launch before it left in L2 / Infinity Cache through the memory system ahead of the stream.  Same-box
    // A/B (tools/ab.sh, tools/ab_sis.sh): fp32 step 45.4 -> 42.7 us on rotating buffers and 44.5 -> 36.8 us inside the
    // SIS loop, right behind the lm_head GEMM that wrote the logits; 16-bit rows (bound by VALU issue) unchanged.
    for (int i = 0; i < NVC; ++i)
      raw[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(q + (int64_t)i * 64 * 16));
  } else {
#pragma unroll
    for (int i = 0; i < NVC; ++i) raw[i] = load_vec_guarded<DT>(rowp, e_base + (i * 64 + lane) * EPV, V);
  }
}

// one loaded vector as floats, times the caller's scale
template <int DT, bool SCALED>
__device__ __forceinline__ void unpack_scaled(const u32x4_t &r, float scale, float *x) {
  unpack_vec<DT>(r, x);
  if constexpr (SCALED) {
#pragma unroll
    for (int k = 0; k < ElemTraits<DT>::EPV; ++k) x[k] *= scale;
  }
}

// the wave's 64 elements per lane of chunk c
template <int DT, bool SCALED>
__device__ __forceinline__ void load_chunk(const char *rowp, int e_base, int V, int lane, float scale,
                                           float (&x)[64]) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  u32x4_t raw[NVC];
  load_chunk_raw<DT>(rowp, e_base, V, lane, raw);
#pragma unroll
  for (int i = 0; i < NVC; ++i) unpack_scaled<DT, SCALED>(raw[i], scale, &x[i * EPV]);
}

// one record out: lanes 0..11 store one granule each (the values are wave-uniform)
__device__ __forceinline__ void store_rec(uint64_t *dst, uint32_t epoch, int lane, float Nc, float Nm, uint32_t pA,
                                          uint32_t pB, const uint32_t (&rA)[4], const uint32_t (&rB)[4]) {
  uint32_t v = __float_as_uint(Nc);
  v = lane == 1 ? __float_as_uint(Nm) : v;
  v = lane == 2 ? pA : v;
  v = lane == 3 ? pB : v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    v = lane == 4 + g ? rA[g] : v;
    v = lane == 8 + g ? rB[g] : v;
  }
  if (lane < 12)
    __hip_atomic_store(dst + lane, ((uint64_t)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same record from the lanes that hold its words already.  F1 / F2: wave scans of the all-element words (lane 63 =
// pA / pB); R1 / R2: row scans of the allowed-element words (lanes 15, 31, 47, 63 = rA[0..3] / rB[0..3]).  Five DPP moves
// gather the twelve words in one register - R1's where they are, R2's one lane to the left (row_shl:1 writes lanes 0..14
// of a row, :2 lanes 0..13, ...: each move leaves the words already placed alone), pB / pA in lanes 61 / 60, Nm / Nc in
// 59 / 58 - and each of those twelve lanes stores its granule.  (Through scalar registers - eight v_readlane, then a
// compare, a move and a select per granule - the record cost the wave 50 vector instructions of its ~530.)
__device__ __forceinline__ void store_rec_lanes(uint64_t *dst, uint32_t epoch, int lane, float Nc, float Nm, uint32_t F1,
                                                uint32_t F2, uint32_t R1, uint32_t R2) {
  uint32_t v = R1;
  v = dpp_u32<0x101, 0xf>(v, R2);
  v = dpp_u32<0x102, 0xf>(v, F2);
  v = dpp_u32<0x103, 0xf>(v, F1);
  v = dpp_u32<0x104, 0xf>(v, __float_as_uint(Nm));
  v = dpp_u32<0x105, 0xf>(v, __float_as_uint(Nc));
  // granule of the lane: 58 -> 0 (Nc), 59 -> 1 (Nm), 60 -> 2 (pA), 61 -> 3 (pB), 16 r + 15 -> 4 + r, 16 r + 14 -> 8 + r
  const int t = lane & 15, hi = lane >> 4;
  const int slot = t >= 14 ? 4 + hi + ((15 - t) << 2) : t - 10;
  const uint32_t off = (uint32_t)slot << 3;
  const uint64_t val = ((uint64_t)epoch << 32) | v;
  uint64_t saved, keep = 0xFC00C000C000C000ull;
  // the twelve lanes store under an EXEC set from a constant, off a scalar base (agent-scope relaxed store: sc1, as
  // __hip_atomic_store makes it).  s_nop 3: with the two scalar instructions before it, the five wait states a VMEM read
  // of an SGPR needs after a VALU write of it (v_readfirstlane) - the compiler does not look inside an asm block
  asm volatile("s_mov_b64 %0, exec\n\t"
               "s_and_b64 exec, %0, %4\n\t"
               "s_nop 3\n\t"
               "global_store_dwordx2 %1, %2, %3 sc1\n\t"
               "s_mov_b64 exec, %0"
               : "=&s"(saved)
               : "v"(off), "v"(val), "s"(dst), "s"(keep)
               : "memory", "scc");
}

// one record in; true when every granule carries `epoch`.  COHERENT: agent-scope loads (past the L1: the stats waves of
// this very launch may still be writing); otherwise plain loads - records of an earlier launch, and when a thousand
// particles share one row (SIS step 0) the CU's L1 serves all but the first sweep.
template <bool COHERENT>
__device__ __forceinline__ bool load_rec(const uint64_t *src, uint32_t epoch, ChunkRec &r) {
  uint64_t g[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    if constexpr (COHERENT) g[k] = __hip_atomic_load(const_cast<uint64_t *>(src) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else g[k] = src[k];
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 12; ++k) ok = ok && (uint32_t)(g[k] >> 32) == epoch;
  r.Nc = __uint_as_float((uint32_t)g[0]);
  r.Nm = __uint_as_float((uint32_t)g[1]);
  r.pA = (uint32_t)g[2];
  r.pB = (uint32_t)g[3];
  r.pAm = r.pBm = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r.rA[k] = (uint32_t)g[4 + k];
    r.rB[k] = (uint32_t)g[8 + k];
    r.pAm += r.rA[k];
    r.pBm += r.rB[k];
  }
  return ok;
}

// the first two groups of mask words of a chunk (8 words each), fetched by the caller before it waits for the chunk's
// own loads and handed to class_partials
struct MaskAhead {
  uint64_t w[2][8];
};
template <int DT, int VSTEP = 1>
__device__ __forceinline__ void mask_ahead(cu64_t mt, MaskAhead &ma, int v0 = 0) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC / VSTEP, VPG = 8 / EPV;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int i = g * VPG + w / EPV;
      ma.w[g][w] = i < NVC ? mt[(v0 + i * VSTEP) * EPV + (w % EPV)] : 0ull;
    }
}

// The (lane, class) partial sums of one chunk held in x[64] (glb_math.hpp): P[w] over all elements of class w, Pm[w]
// over the allowed ones.  (v0, VSTEP): this wave's vectors are v0, v0 + VSTEP, ... and x holds them densely - one wave
// per chunk: (0, 1), four classes per lane; four waves per chunk: (wave, 4), every vector of a wave is of class `wave`
// and the sums land in P[0] / Pm[0].
template <int DT, bool MASKED, int VSTEP, bool FULL, int EXPC>
__device__ __forceinline__ void class_partials_body(const float (&x)[64], float magicN, int nv_valid, cu64_t mt, int v0,
                                                    const MaskAhead &ahead, float (&P)[4], float (&Pm)[4]) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC / VSTEP;
  // mask words travel in groups of 8 (one s_load_dwordx16) fetched two groups ahead of the adds that use them: a
  // scalar load that misses the scalar cache takes longer than the ~40 VALU issues of one group
  constexpr int GW = 8, VPG = GW / EPV > 0 ? GW / EPV : 1, NG = (NVC + VPG - 1) / VPG;  // vectors per group, groups
  uint64_t Mq[3][GW] = {};
  auto fetch = [&](int g, uint64_t (&dst)[GW]) {
#pragma unroll
    for (int w = 0; w < GW; ++w) {
      const int i = g * VPG + w / EPV;  // this wave's i-th vector
      dst[w] = i < NVC ? mt[(v0 + i * VSTEP) * EPV + (w % EPV)] : 0ull;
    }
  };
  if constexpr (MASKED) {
#pragma unroll
    for (int w = 0; w < GW; ++w) {
      Mq[0][w] = ahead.w[0][w];
      Mq[1][w] = ahead.w[1][w];
    }
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if constexpr (MASKED) {
      if (g + 2 < NG) fetch(g + 2, Mq[(g + 2) % 3]);
    }
#pragma unroll
    for (int iv = 0; iv < VPG; ++iv) {
      const int i = g * VPG + iv;
      if (i >= NVC) break;
      const int vi = v0 + i * VSTEP;  // the vector's index inside the chunk
      constexpr int kOne = VSTEP == 1 ? 3 : 0;
      const int cls = i & kOne;       // (compile-time after unrolling)
      if (FULL || vi < nv_valid) {  // wave-uniform: vectors wholly past the row end are skipped (they would add +0)
        float t[EPV];
#pragma unroll
        for (int k = 0; k < EPV; ++k) t[k] = chunk_term<EXPC>(x[i * EPV + k], magicN);
#pragma unroll
        for (int k = 0; k < EPV; ++k) P[cls] = P[cls] + t[k];
        if constexpr (MASKED) {
#pragma unroll
          for (int h = 0; h < EPV / 4; ++h) {
            const uint64_t *M = &Mq[g % 3][iv * EPV + 4 * h];
            masked_add4(Pm[cls], t[4 * h], t[4 * h + 1], t[4 * h + 2], t[4 * h + 3], M[0], M[1], M[2], M[3]);
          }
        }
      }
    }
  }
}

// (magicN: term_bias<EXPC>(the scale) - the name is the polynomial contract's)
template <int DT, bool MASKED, int VSTEP = 1, int EXPC = kExpPoly>
__device__ __forceinline__ void class_partials(const float (&x)[64], float magicN, int nv_valid, cu64_t mt,
                                               const MaskAhead &ahead, float (&P)[4], float (&Pm)[4], int v0 = 0) {
#pragma unroll
  for (int w = 0; w < 4; ++w) P[w] = Pm[w] = 0.0f;
  // a full chunk (all but the last of a row) runs as one straight block: the scheduler can then start the scalar loads
  // of the mask words well ahead of the adds that use them; the last chunk tests every vector against the row end
  if (nv_valid == ElemTraits<DT>::NVC)
    class_partials_body<DT, MASKED, VSTEP, true, EXPC>(x, magicN, nv_valid, mt, v0, ahead, P, Pm);
  else
    class_partials_body<DT, MASKED, VSTEP, false, EXPC>(x, magicN, nv_valid, mt, v0, ahead, P, Pm);
}

// kMaskRaw: bit i * EPV + k of the result = this lane's element (vector i, component k) of chunk c is allowed.  The EPV bits
// of a vector sit in one word of the row (fp32: 4 bits at a multiple of 4; 16-bit: one byte); elements past the row's end
// read as forbidden.
template <int DT>
__device__ __forceinline__ uint64_t lane_mask_bits(const uint32_t *brow, int e_base, int V, int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  const uint32_t last_word = (uint32_t)((V + 31) >> 5) - 1u;
  // e_base is a multiple of 4096: the vector's word is (e_base >> 5) + i * (2 * EPV) + ((lane * EPV) >> 5), its bits start
  // at (lane * EPV) & 31 in every one of them.  Words past the row's end are not read (the index is clamped; their bits
  // are cleared below) - a load under a branch per vector cost the 16-bit launch 8 scalar branches a wave
  const uint32_t w0 = (uint32_t)(e_base >> 5) + ((uint32_t)(lane * EPV) >> 5), sh = (uint32_t)(lane * EPV) & 31u;
  uint32_t w[NVC];
#pragma unroll
  for (int i = 0; i < NVC; ++i) {  // (loads first)
    uint32_t wi = w0 + (uint32_t)(i * 2 * EPV);
    wi = wi < last_word ? wi : last_word;
    w[i] = *(const uint32_t *)((const char *)brow + (wi << 2));
  }
  uint64_t X = 0;
  if (e_base + kChunk <= V) {  // wave-uniform: all but the last chunk of a row
#pragma unroll
    for (int i = 0; i < NVC; ++i) X |= (uint64_t)((w[i] >> sh) & ((1u << EPV) - 1u)) << (i * EPV);
  } else {
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int valid = V - (e_base + (i * 64 + lane) * EPV);
      uint32_t v = (w[i] >> sh) & ((1u << EPV) - 1u);
      if (valid < EPV) v &= valid > 0 ? ((1u << valid) - 1u) : 0u;
      X |= (uint64_t)v << (i * EPV);
    }
  }
  return X;
}

// class_partials on per-lane bits: P over all elements, Pm over the allowed ones - a term ANDed with 0 / ~0 by its bit
// (v_bfe_i32: a one-bit field, sign-extended), so a forbidden term enters the allowed sum as +0
template <int DT, int EXPC, bool ONLY_MASKED, bool FULL>
__device__ __forceinline__ void class_partials_raw_body(const float (&x)[64], float bias, int nv_valid, uint64_t X,
                                                        float (&P)[4], float (&Pm)[4]) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  const uint32_t Xlo = (uint32_t)X, Xhi = (uint32_t)(X >> 32);
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    if (FULL || i < nv_valid) {  // wave-uniform: vectors wholly past the row's end would add +0
#pragma unroll
      for (int k = 0; k < EPV; ++k) {
        const int j = i * EPV + k;  // (compile-time after unrolling)
        const float t = chunk_term<EXPC>(x[j], bias);
        if constexpr (!ONLY_MASKED) P[i & 3] = P[i & 3] + t;
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)(j < 32 ? Xlo : Xhi), (unsigned)(j & 31), 1u);
        Pm[i & 3] = Pm[i & 3] + __uint_as_float(__float_as_uint(t) & m);
      }
    }
  }
}

template <int DT, int EXPC, bool ONLY_MASKED = false>
__device__ __forceinline__ void class_partials_raw(const float (&x)[64], float bias, int nv_valid, uint64_t X, float (&P)[4],
                                                   float (&Pm)[4]) {
#pragma unroll
  for (int w = 0; w < 4; ++w) P[w] = Pm[w] = 0.0f;
  if (nv_valid == ElemTraits<DT>::NVC)  // (a full chunk runs as one straight block, as in class_partials)
    class_partials_raw_body<DT, EXPC, ONLY_MASKED, true>(x, bias, nv_valid, X, P, Pm);
  else
    class_partials_raw_body<DT, EXPC, ONLY_MASKED, false>(x, bias, nv_valid, X, P, Pm);
}

// this lane's payload words: the sum over its NCLS class partials of floor(P * 2^36), h and l words apart
template <int NCLS>
__device__ __forceinline__ void lane_payload(const float (&P)[4], uint32_t &h, uint32_t &l) {
  h = l = 0;
#pragma unroll
  for (int w = 0; w < NCLS; ++w) {
    uint32_t hw, lw;
    partial_q(P[w], hw, lw);
    h += hw;
    l += lw;
  }
}

// lane-63 value of a DPP scan, wave-uniform
__device__ __forceinline__ uint32_t last_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

__device__ __forceinline__ float chunk_max(const float (&x)[64]) {
  float m = kNegInf;
#pragma unroll
  for (int j = 0; j < 64; j += 2) m = max3(m, x[j], x[j + 1]);
  return wave_max(m);
}

// Bit-masked chunk: both sums at the chunk scale Nc (one exponential per element); if the allowed sum comes out
// below 2^32 although the mask allows something in this chunk (allows_any, from mask_prepare) - allowed mass under
// about 2^-3.5 of the chunk's largest term, e.g. the one likely token is the forbidden one - the chunk is loaded once
// more (wave-uniform, rare, L1 / L2 serve it; not keeping x alive for this is what lets the common path run without
// spills), its forbidden elements overwritten with -inf and the allowed ones summed again on their own maximum's
// scale.  Out (wave-uniform): the totals of all elements, the allowed elements' sums by 16-lane row, and their scale Nm.
template <int DT, bool SCALED, int EXPC>
__device__ __forceinline__ void chunk_reduce_bits(float (&x)[64], float Nc, int nv_valid, cu64_t mt,
                                                  const MaskAhead &ma, uint64_t allows_any, int lane, const char *rowp,
                                                  int e_base, int V, float scale, uint32_t &F1, uint32_t &F2,
                                                  uint32_t &R1, uint32_t &R2, float &Nm) {
  float P[4], Pm[4];
  class_partials<DT, true, 1, EXPC>(x, term_bias<EXPC>(Nc), nv_valid, mt, ma, P, Pm);
  uint32_t h, l, hm, lm;
  lane_payload<4>(P, h, l);
  lane_payload<4>(Pm, hm, lm);
  F1 = wave_sum_u32_l63(h);
  F2 = wave_sum_u32_l63(l);
  R1 = row_scan_u32(hm);
  R2 = row_scan_u32(lm);
  Nm = Nc;
  const uint64_t Sm = ((uint64_t)rows_total(R1) << kGridHi) + rows_total(R2);
  uint32_t top = (uint32_t)(Sm >> kLowMassBits), any = (uint32_t)allows_any;
  opaque_u32(top);
  opaque_u32(any);
  if (top == 0u && any != 0u) {
    float y[64];
    int lane_again = lane;  // (the lane's address is made again, from a lane id the compiler cannot match with the first:
    asm volatile("" : "+v"(lane_again));  // not held - or spilled - across the reduction for this rare branch)
    load_chunk<DT, SCALED>(rowp, e_base, V, lane_again, scale, y);
#pragma unroll
    for (int j = 0; j < 64; ++j)
      if (!((mt[j] >> lane) & 1ull)) y[j] = kNegInf;
    Nm = exp_n(chunk_max(y));
    class_partials<DT, false, 1, EXPC>(y, term_bias<EXPC>(Nm), nv_valid, nullptr, MaskAhead{}, P, Pm);
    lane_payload<4>(P, hm, lm);
    R1 = row_scan_u32(hm);
    R2 = row_scan_u32(lm);
  }
}

// the particle's two 64-bit draws (chunk stage, in-chunk stage): Philox4x32-10 keyed by the call's seed, counter =
// (global particle index, call offset)
__device__ __forceinline__ void philox_pair(const StepParams &p, int pidx, uint64_t &R1, uint64_t &R2) {
  const uint64_t gp = (uint64_t)(p.particle_base + pidx);
  const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
  const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
  uint32_t rnd[4];
  philox4x32_10(ctr, key, rnd);
  R1 = ((uint64_t)rnd[1] << 32) | rnd[0];
  R2 = ((uint64_t)rnd[3] << 32) | rnd[2];
}

// y = x + (float mask row) for the wave's chunk (same lane layout as x)
template <int DT>
__device__ __forceinline__ void add_float_mask(const float (&x)[64], const char *mrow, int e_base, int V, int lane,
                                               float (&y)[64]) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  if constexpr (EPV == 4) {
    load_chunk<kDtF32, false>(mrow, e_base, V, lane, 1.0f, y);
#pragma unroll
    for (int j = 0; j < 64; ++j) y[j] = x[j] + y[j];
  } else {
    // 16-bit logits: lane holds 8 consecutive elements per vector, i.e. two 16-byte vectors of the float mask
#pragma unroll
    for (int i = 0; i < NVC; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e0 = e_base + (i * 64 + lane) * 8 + 4 * h;
        const u32x4_t r = load_vec_guarded<kDtF32>(mrow, e0, V);
        float mk[4];
        unpack_vec<kDtF32>(r, mk);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[i * 8 + 4 * h + k] = x[i * 8 + 4 * h + k] + mk[k];
      }
  }
}

// first lane whose inclusive scan exceeds T (wave-uniform T); -1 if none
__device__ __forceinline__ int first_lane_above(uint64_t incl, uint64_t T) {
  return __ffsll((long long)__ballot(incl > T)) - 1;
}

// ---------------------------------------------------------------------------------------------------------
// The draw inside one chunk (second stage of the Philox draw): target T2 = floor(R2 * S_c / 2^64) against
// the chunk's allowed sums on the scale they were made on, walked down the summation tree: lane (inclusive scan of the
// lane payloads), class (the lane's four class terms), element (running float32 sum of the class, the first whose
// floor(c * 2^36) passes what is left of the target).
// ---------------------------------------------------------------------------------------------------------
// the lane.  Packed result (lane << 56) | (target left inside that lane's terms); ~0 for an empty chunk.
__device__ __forceinline__ uint64_t chunk_pick_lane(uint32_t inclA, uint32_t inclB, uint64_t R2) {
  const uint64_t incl = ((uint64_t)inclA << kGridHi) + inclB;
  const uint64_t Sc = readlane_u64(incl, 63);
  uint32_t nz = (uint32_t)Sc | (uint32_t)(Sc >> 32);
  opaque_u32(nz);
  if (nz == 0u) return ~0ull;
  const uint64_t T2 = __umul64hi(R2, Sc);  // uniform integer in [0, S_c)
  const int lsel = first_lane_above(incl, T2);
  const uint64_t before = readlane_u64(incl, lsel > 0 ? lsel - 1 : 0);
  const uint64_t Tl = T2 - (lsel > 0 ? before : 0ull);  // < that lane's total < 2^42
  return ((uint64_t)lsel << 56) | Tl;
}

// ---------------------------------------------------------------------------------------------------------
// stats role: one wave per (reduction unit, chunk)
// ---------------------------------------------------------------------------------------------------------
template <int DT, int MASK, bool SCALED, int EXPC>
__device__ __forceinline__ void stats_item(const StepParams &p, int item, int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  const int nch = p.nch;
  GLB_DIAG(const uint64_t stamp0 = GLB_NOW();)
  // (the quotient of two scalars comes out of the vector unit's reciprocal: told to be a scalar, or it - and the row's and the
  // record's addresses made from it - live in vector registers for the whole wave)
  int pr = __builtin_amdgcn_readfirstlane(item / nch), c = item - pr * nch;  // (dealt chunk-major instead - every row's chunk 0, then every row's chunk 1, ... - the launch is 2.5 us slower)
  GLB_DIAG(
    if (p.diag.short_last) {
      const int nfull = nch - 1, split = p.n_pairs * nfull;
      pr = item < split ? item / nfull : item - split;
      c = item < split ? item - pr * nfull : nfull;
    }
  )
  int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  GLB_DIAG(if (p.diag.dbg_mode == 2) row &= 7;)
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int V = p.V, e_base = c * kChunk;
  int nv_valid = (V - e_base + 64 * EPV - 1) / (64 * EPV);
  nv_valid = nv_valid < NVC ? nv_valid : NVC;
  GLB_DIAG(if (p.diag.dbg_mode == 1) nv_valid = 1;)

  // the chunk's loads go out first: everything below that has to wait for a scalar load (the mask id of the unit, then
  // the mask words) waits while they are in flight, not in front of them
  float x[64];
  load_chunk<DT, SCALED>(rowp, e_base, V, lane, p.scale, x);
  MaskAhead ma{};
  cu64_t mt = nullptr;
  uint64_t allows_any = 0;
  int mi = 0;
  if constexpr (MASK != kMaskNone) mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
  if constexpr (MASK == kMaskBits) {
    mt = as_const(p.mask_t + ((int64_t)mi * nch + c) * 64);
    mask_ahead<DT>(mt, ma);
    allows_any = as_const(p.mask_any)[(int64_t)mi * nch + c];
  }
  uint64_t X = 0;  // kMaskRaw: this lane's 64 allowed bits (their words travel with the chunk's loads)
  if constexpr (MASK == kMaskRaw) X = lane_mask_bits<DT>(p.mask_bits + (int64_t)mi * p.mask_bits_ld, e_base, V, lane);
  const float Nc = exp_n(chunk_max(x));
  uint32_t F1, F2, R1, R2;  // the record's words as the scans leave them in the lanes (store_rec_lanes)
  float Nm = Nc;
  if constexpr (MASK == kMaskBits) {
    chunk_reduce_bits<DT, SCALED, EXPC>(x, Nc, nv_valid, mt, ma, allows_any, lane, rowp, e_base, V, p.scale, F1, F2, R1, R2, Nm);
  } else if constexpr (MASK == kMaskRaw) {
    // chunk_reduce_bits on per-lane bits: both sums on the chunk's scale; a chunk that allows something but whose allowed
    // sum comes out below 2^32 sums its allowed values again on their own maximum's scale
    float P[4], Pm[4];
    class_partials_raw<DT, EXPC>(x, term_bias<EXPC>(Nc), nv_valid, X, P, Pm);
    uint32_t h, l, hm, lm;
    lane_payload<4>(P, h, l);
    lane_payload<4>(Pm, hm, lm);
    F1 = wave_sum_u32_l63(h);
    F2 = wave_sum_u32_l63(l);
    R1 = row_scan_u32(hm);
    R2 = row_scan_u32(lm);
    const uint64_t Sm = ((uint64_t)rows_total(R1) << kGridHi) + rows_total(R2);
    const uint64_t votes = __ballot(X != 0ull);
    uint32_t top = (uint32_t)(Sm >> kLowMassBits), any = (uint32_t)votes | (uint32_t)(votes >> 32);
    opaque_u32(top);
    opaque_u32(any);
    if (top == 0u && any != 0u) {
      float y[64];  // (loaded once more, as chunk_reduce_bits does: x is not kept alive for this rare branch - and neither
      int lane_again = lane;  // is the lane's address: made again from a lane id the compiler cannot match with the first)
      asm volatile("" : "+v"(lane_again));
      load_chunk<DT, SCALED>(rowp, e_base, V, lane_again, p.scale, y);
      float mm = kNegInf;
#pragma unroll
      for (int j = 0; j < 64; ++j) mm = fmaxf(mm, ((X >> j) & 1ull) ? y[j] : kNegInf);
      Nm = exp_n(wave_max(mm));
      class_partials_raw<DT, EXPC, true>(y, term_bias<EXPC>(Nm), nv_valid, X, P, Pm);
      lane_payload<4>(Pm, hm, lm);
      R1 = row_scan_u32(hm);
      R2 = row_scan_u32(lm);
    }
  } else {
    float P[4], Pm[4];
    uint32_t h, l;
    class_partials<DT, false, 1, EXPC>(x, term_bias<EXPC>(Nc), nv_valid, nullptr, ma, P, Pm);
    lane_payload<4>(P, h, l);
    R1 = row_scan_u32(h);
    R2 = row_scan_u32(l);
    F1 = rows_scan_to_wave(R1);  // (pA / pB: the row scans taken on to lane 63)
    F2 = rows_scan_to_wave(R2);
    if constexpr (MASK == kMaskF32) {  // general additive masks: y = x + m has its own maximum, scale and exp
      const char *mrow = (const char *)(p.mask_f + (int64_t)mi * p.mask_ld);
      float y[64];
      add_float_mask<DT>(x, mrow, e_base, V, lane, y);
      Nm = exp_n(chunk_max(y));
      class_partials<DT, false, 1, EXPC>(y, term_bias<EXPC>(Nm), nv_valid, nullptr, ma, P, Pm);
      lane_payload<4>(P, h, l);
      R1 = row_scan_u32(h);
      R2 = row_scan_u32(l);
    }
  }
  // (the record's address is made here, from scalars the compiler cannot see through: hoisted to the top of the wave, a
  // lane's 64-bit address is two registers held - or spilled: the raw-mask kernel did - across the whole reduction)
  int pr_s = __builtin_amdgcn_readfirstlane(pr), c_s = __builtin_amdgcn_readfirstlane(c);
  int lane_s = lane;
  asm volatile("" : "+s"(pr_s), "+s"(c_s), "+v"(lane_s));
  int rec = pr_s * nch + c_s;  // (< 2^31: a record is 128 bytes; as one scalar, so that its address is made by the scalar unit)
  asm volatile("" : "+s"(rec));
  store_rec_lanes(p.recs + (int64_t)rec * kRecWords, p.epoch, lane_s, Nc, Nm, F1, F2, R1, R2);
  GLB_DIAG(
    if (lane == 0) {
      uint64_t *r = p.recs + ((int64_t)pr * nch + c) * kRecWords;
      r[12] = stamp0;
      r[13] = GLB_NOW();
    }
  )
}

// ---------------------------------------------------------------------------------------------------------
// stats role in quarters, for launches that cannot fill the chip (a single shared row at SIS step 0, a lone API
// query): one WORKGROUP of four waves per (unit, chunk); wave w takes vectors w, w+4, ... of the chunk - exactly class w
// of every lane - the chunk maximum and the payload sums are combined through LDS.  Same records as the one-wave form by
// construction: the (lane, class) partials are the same sums in the same order, only held by another wave.  A wave
// alone on its SIMD issues one instruction every ~5 cycles, so a quarter of the elements per wave is what shortens the
// launch.  (As the tail of a big launch the quarters gain nothing: tools/dbg/stamps.py)  Bit masks / no
// mask only (float masks use the one-wave form).
// ---------------------------------------------------------------------------------------------------------
template <int DT, int MASK, bool SCALED, int EXPC = kExpPoly>
__global__ __launch_bounds__(256) void chunk_stats_small_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC, NVW = NVC / 4;
  static_assert(MASK != kMaskF32, "float masks take the one-wave-per-chunk form");
  __shared__ float s_max[4];
  __shared__ uint32_t s_pay[4][10];  // per wave: pA, pB, rA[4], rB[4]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x, nch = p.nch;
  GLB_DIAG(const uint64_t stamp0 = GLB_NOW();)
  const int pr = item / nch, c = item - pr * nch;
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int V = p.V, e_base = c * kChunk;
  int nv_valid = (V - e_base + 64 * EPV - 1) / (64 * EPV);
  nv_valid = nv_valid < NVC ? nv_valid : NVC;
  float x[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) x[j] = kNegInf;
#pragma unroll
  for (int j = 0; j < NVW; ++j) {
    const int e0 = e_base + ((wave + 4 * j) * 64 + lane) * EPV;
    const u32x4_t r = load_vec_guarded<DT>(rowp, e0 < V ? e0 : V, V);
    unpack_vec<DT>(r, &x[j * EPV]);
    if constexpr (SCALED) {
#pragma unroll
      for (int k = 0; k < EPV; ++k) x[j * EPV + k] *= p.scale;
    }
  }
  float m = kNegInf;
#pragma unroll
  for (int j = 0; j < NVW * EPV; j += 2) m = max3(m, x[j], x[j + 1]);
  m = wave_max(m);
  if (lane == 0) s_max[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  const float Nc = exp_n(m);
  float P[4], Pm[4];
  uint32_t h, l, hm, lm;
  if constexpr (MASK == kMaskBits) {
    const int mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
    const cu64_t mt = as_const(p.mask_t + ((int64_t)mi * nch + c) * 64);
    MaskAhead ma;
    mask_ahead<DT, 4>(mt, ma, wave);
    class_partials<DT, true, 4, EXPC>(x, term_bias<EXPC>(Nc), nv_valid, mt, ma, P, Pm, wave);
    lane_payload<1>(P, h, l);
    lane_payload<1>(Pm, hm, lm);
  } else {
    class_partials<DT, false, 4, EXPC>(x, term_bias<EXPC>(Nc), nv_valid, nullptr, MaskAhead{}, P, Pm, wave);
    lane_payload<1>(P, h, l);
    hm = h;
    lm = l;
  }
  uint32_t rA[4], rB[4];
  {
    const uint32_t wA = wave_sum_u32_l63(h), wB = wave_sum_u32_l63(l);
    row_sums_u32(hm, rA);
    row_sums_u32(lm, rB);
    if (lane == 63) {
      s_pay[wave][0] = wA;
      s_pay[wave][1] = wB;
    }
    if (lane < 4) {
      s_pay[wave][2 + lane] = lane == 0 ? rA[0] : lane == 1 ? rA[1] : lane == 2 ? rA[2] : rA[3];
      s_pay[wave][6 + lane] = lane == 0 ? rB[0] : lane == 1 ? rB[1] : lane == 2 ? rB[2] : rB[3];
    }
  }
  __syncthreads();
  uint32_t t[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) t[k] = s_pay[0][k] + s_pay[1][k] + s_pay[2][k] + s_pay[3][k];
  float Nm = Nc;
  if constexpr (MASK == kMaskBits) {
    // the low-mass rule of chunk_reduce_bits, with the chunk spread over four waves (workgroup-uniform branch)
    const uint64_t Sm = ((uint64_t)(t[2] + t[3] + t[4] + t[5]) << kGridHi) + (t[6] + t[7] + t[8] + t[9]);
    const int mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
    uint32_t top = (uint32_t)(Sm >> kLowMassBits), any = (uint32_t)as_const(p.mask_any)[(int64_t)mi * nch + c];
    opaque_u32(top);
    opaque_u32(any);
    if (top == 0u && any != 0u) {
      const cu64_t mt = as_const(p.mask_t + ((int64_t)mi * nch + c) * 64);
#pragma unroll
      for (int j = 0; j < NVW; ++j) {  // (loaded again: x is not kept alive for this rare branch)
        const int e0 = e_base + ((wave + 4 * j) * 64 + lane) * EPV;
        const u32x4_t r = load_vec_guarded<DT>(rowp, e0 < V ? e0 : V, V);
        unpack_vec<DT>(r, &x[j * EPV]);
#pragma unroll
        for (int k = 0; k < EPV; ++k) {
          if constexpr (SCALED) x[j * EPV + k] *= p.scale;
          if (!((mt[(wave + 4 * j) * EPV + k] >> lane) & 1ull)) x[j * EPV + k] = kNegInf;
        }
      }
      float mm = kNegInf;
#pragma unroll
      for (int j = 0; j < NVW * EPV; j += 2) mm = max3(mm, x[j], x[j + 1]);
      mm = wave_max(mm);
      __syncthreads();  // everybody has read s_max / s_pay
      if (lane == 0) s_max[wave] = mm;
      __syncthreads();
      Nm = exp_n(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])));
      class_partials<DT, false, 4, EXPC>(x, term_bias<EXPC>(Nm), nv_valid, nullptr, MaskAhead{}, P, Pm, wave);
      lane_payload<1>(P, hm, lm);
      row_sums_u32(hm, rA);
      row_sums_u32(lm, rB);
      if (lane < 4) {
        s_pay[wave][2 + lane] = lane == 0 ? rA[0] : lane == 1 ? rA[1] : lane == 2 ? rA[2] : rA[3];
        s_pay[wave][6 + lane] = lane == 0 ? rB[0] : lane == 1 ? rB[1] : lane == 2 ? rB[2] : rB[3];
      }
      __syncthreads();
#pragma unroll
      for (int k = 2; k < 10; ++k) t[k] = s_pay[0][k] + s_pay[1][k] + s_pay[2][k] + s_pay[3][k];
    }
  }
  if (wave == 0) {
    const uint32_t qa[4] = {t[2], t[3], t[4], t[5]}, qb[4] = {t[6], t[7], t[8], t[9]};
    store_rec(p.recs + ((int64_t)pr * nch + c) * kRecWords, p.epoch, lane, Nc, Nm, t[0], t[1], qa, qb);
  }
  GLB_DIAG(
    if (wave == 0 && lane == 0) {
      uint64_t *r = p.recs + ((int64_t)pr * nch + c) * kRecWords;
      r[12] = stamp0;
      r[13] = GLB_NOW();
    }
  )
}

// ---------------------------------------------------------------------------------------------------------
// finish: one wave per particle
// ---------------------------------------------------------------------------------------------------------
// the row (or float-mask) view one particle works on
template <int DT, int MASK>
struct RowView {
  const char *rowp;
  int V;
  float scale;
  const uint64_t *mt;   // transposed mask row (kMaskBits)
  const uint64_t *many; // its per-chunk "allows anything" words
  const char *mrow;     // float mask row (kMaskF32)
  static constexpr int EPV = ElemTraits<DT>::EPV;

  // masked values of vector (c, i) for this lane: y[k] = x'[k] (+ mask) or -inf when forbidden / past the row
  __device__ __forceinline__ void vec(int c, int i, int lane, float (&y)[EPV], int &e0) const {
    e0 = c * kChunk + (i * 64 + lane) * EPV;
    const u32x4_t r = load_vec_guarded<DT>(rowp, e0 < V ? e0 : V, V);
    unpack_vec<DT>(r, y);
#pragma unroll
    for (int k = 0; k < EPV; ++k) y[k] = y[k] * scale;
    if constexpr (MASK == kMaskBits) {
#pragma unroll
      for (int k = 0; k < EPV; ++k) {
        const uint64_t w = as_const(mt)[(int64_t)c * 64 + i * EPV + k];
        if (!((w >> lane) & 1ull)) y[k] = kNegInf;
      }
    } else if constexpr (MASK == kMaskF32) {
#pragma unroll
      for (int h = 0; h < EPV / 4; ++h) {
        const int eh = e0 + 4 * h;
        const u32x4_t mr = load_vec_guarded<kDtF32>(mrow, eh < V ? eh : V, V);
        float mk[4];
        unpack_vec<kDtF32>(mr, mk);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[4 * h + k] = y[4 * h + k] + mk[k];
      }
    }
#pragma unroll
    for (int k = 0; k < EPV; ++k)
      if (e0 + k >= V) y[k] = kNegInf;
  }
};

// ---------------------------------------------------------------------------------------------------------
// What one particle needs from the chunk records of its (row, mask) pair.  Every function here is executed by ONE
// wave; all results are wave-uniform.
// ---------------------------------------------------------------------------------------------------------
// the records of one reduction unit.  Rows of up to 64 chunks (V <= 262144) stay in registers after the sweep that
// waited for them: lane c holds record c.
struct Recs {
  const uint64_t *base;
  uint32_t epoch;
  bool cached;
  ChunkRec mine;
  __device__ __forceinline__ ChunkRec get(int c) const {  // always called with c = c0 + lane
    if (cached) return mine;
    ChunkRec r;
    load_rec<true>(base + (int64_t)c * kRecWords, epoch, r);
    return r;
  }
};

// POLL: sweep the unit's granules until every tag is this call's epoch (the stats waves of the same launch are still
// writing them; relaxed agent-scope loads, a short sleep between sweeps, a bounded wait).  Without POLL the records
// come from an earlier launch on the stream and the tags are not looked at.
template <bool POLL>
__device__ __forceinline__ bool recs_acquire(Recs &R, int nch, int lane, uint64_t spin_ticks) {
  R.cached = nch <= 64;
  R.mine = ChunkRec{kNegInf, kNegInf, 0u, 0u, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, 0u, 0u};
  uint64_t t0 = 0;
  if constexpr (POLL) t0 = GLB_NOW();
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const int c = c0 + lane;
    for (;;) {
      if constexpr (POLL) {
        // waiting costs one 8-byte load per record and sweep (the last granule of the record's one store instruction);
        // the full, validated read comes when every record shows it - a row's finishing wave may sweep for
        // microseconds, and twelve loads per record from a thousand waiting waves are L2 requests the streaming
        // reads of the still running stats waves queue behind
        bool seen = true;
        if (c < nch) {
          const uint64_t g = __hip_atomic_load(const_cast<uint64_t *>(R.base) + (int64_t)c * kRecWords + 11, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
          seen = (uint32_t)(g >> 32) == R.epoch;
        }
        if (__builtin_amdgcn_ballot_w64(!seen) != 0ull) {
          __builtin_amdgcn_s_sleep(2);
          if (GLB_NOW() - t0 > spin_ticks) return false;
          continue;
        }
      }
      bool ok = true;
      ChunkRec r = ChunkRec{kNegInf, kNegInf, 0u, 0u, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, 0u, 0u};
      if (c < nch) ok = load_rec<POLL>(R.base + (int64_t)c * kRecWords, R.epoch, r);
      if (c0 == 0) R.mine = r;
      if constexpr (!POLL) break;
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
      __builtin_amdgcn_s_sleep(2);
      if (GLB_NOW() - t0 > spin_ticks) return false;
    }
  }
  return true;
}

struct PairState {
  float N_all, N_msk;     // row scales (all / allowed)
  uint64_t S_all, S_msk;  // sums on them
};

// fold the chunk records: row scales (the largest chunk scale among non-empty chunks), then the sums shifted onto them
template <int MASK>
__device__ __forceinline__ void pair_fold(const Recs &recs, int nch, int lane, PairState &st) {
  float N_all = kNegInf, N_msk = kNegInf;
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const int c = c0 + lane;
    if (c < nch) {
      const ChunkRec r = recs.get(c);
      if (r.pA | r.pB) N_all = fmaxf(N_all, r.Nc);
      if (r.pAm | r.pBm) N_msk = fmaxf(N_msk, r.Nm);
    }
  }
  N_all = wave_max(N_all);
  N_msk = MASK == kMaskNone ? N_all : wave_max(N_msk);
  uint64_t S_all = 0, S_msk = 0;
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const int c = c0 + lane;
    uint64_t sa = 0, sm = 0;
    if (c < nch) {
      const ChunkRec r = recs.get(c);
      const uint64_t a = ((uint64_t)r.pA << kGridHi) + r.pB, m = ((uint64_t)r.pAm << kGridHi) + r.pBm;
      if (a) {
        const float d = N_all - r.Nc;  // >= 0, integer valued
        sa = d < 64.0f ? a >> (uint32_t)d : 0ull;
      }
      if (m) {
        const float d = N_msk - r.Nm;
        sm = d < 64.0f ? m >> (uint32_t)d : 0ull;
      }
    }
    S_all += wave_sum_u64(sa);
    if constexpr (MASK != kMaskNone) S_msk += wave_sum_u64(sm);
  }
  if constexpr (MASK == kMaskNone) S_msk = S_all;
  st.N_all = N_all;
  st.N_msk = N_msk;
  st.S_all = S_all;
  st.S_msk = S_msk;
}

// lse / logZ (sum of e^x = 2^(N + 1 - 36) * S), called by the whole wave: lane 1 takes the logarithm of the allowed
// sum while lane 0 takes the one of all elements - one pass through the double-precision instruction stream instead of
// two (a finishing wave is alone on its SIMD and these ~80 dependent fp64 operations sit on the launch's last
// microseconds).  Results wave-uniform.
__device__ __forceinline__ void pair_logs(const PairState &st, int lane, float &lse, float &logZ) {
  const bool msk = lane == 1;
  const uint64_t S = msk ? st.S_msk : st.S_all;
  const float N = msk ? st.N_msk : st.N_all;
  double l = (double)kNegInf;
  if (S) l = log_fix(S, (int32_t)N + 1 - kFrac);
  const uint64_t lb = (uint64_t)__double_as_longlong(l);
  const double lse_all = __longlong_as_double((long long)readlane_u64(lb, 0));
  const double lse_msk = __longlong_as_double((long long)readlane_u64(lb, 1));
  lse = (float)lse_all;
  logZ = (float)(lse_msk - lse_all);
}

// first stage of the Philox draw of particle pidx: the chunk, by a scan of the shifted chunk sums (first word)
struct ChunkPick {
  int csel;        // -1: nothing to draw from
  float Nms;       // the scale the chosen chunk's allowed terms sit on
  uint64_t R2;     // second Philox word
  uint64_t qg[4];  // the chosen chunk's allowed sums by 16-lane row
};

__device__ __forceinline__ ChunkPick pair_pick_chunk(const Recs &recs, const PairState &st, uint64_t R1, uint64_t R2,
                                                     int nch, int lane) {
  ChunkPick k{-1, 0.f, R2, {0ull, 0ull, 0ull, 0ull}};
  uint32_t nz = (uint32_t)st.S_msk | (uint32_t)(st.S_msk >> 32);
  opaque_u32(nz);
  if (nz == 0u) return k;
  uint64_t T = __umul64hi(R1, st.S_msk);  // uniform integer in [0, S_msk)
  for (int c0 = 0; c0 < nch && k.csel < 0; c0 += 64) {
    const int c = c0 + lane;
    uint64_t sm = 0;
    ChunkRec r{kNegInf, kNegInf, 0u, 0u, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, 0u, 0u};
    if (c < nch) {
      r = recs.get(c);
      const uint64_t m = ((uint64_t)r.pAm << kGridHi) + r.pBm;
      if (m) {
        const float d = st.N_msk - r.Nm;
        if (d < 64.0f) sm = m >> (uint32_t)d;
      }
    }
    const uint64_t incl = wave_scan_u64(sm);
    const int lsel = first_lane_above(incl, T);
    if (lsel >= 0) {
      k.csel = c0 + lsel;
      k.Nms = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(r.Nm), lsel));
#pragma unroll
      for (int g = 0; g < 4; ++g)
        k.qg[g] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r.rA[g], lsel) << kGridHi) +
                  (uint32_t)__builtin_amdgcn_readlane((int)r.rB[g], lsel);
    } else {
      T -= readlane_u64(incl, 63);
    }
  }
  return k;
}

// ---------------------------------------------------------------------------------------------------------
// finish role, statistics mode (no draw): one wave per particle
// ---------------------------------------------------------------------------------------------------------
template <int MASK, bool POLL>
__device__ __forceinline__ void finish_stats(const StepParams &p, int pidx, int lane) {
  const int nch = p.nch;
  const int pr = p.pair_of ? as_const(p.pair_of)[pidx] : pidx;
  Recs recs;
  recs.base = p.recs + (int64_t)pr * nch * kRecWords;
  recs.epoch = p.epoch;
  float lse = __uint_as_float(0x7fc00000u), logZ = lse;  // (a wave that gave up waiting: NaN, never in a healthy launch)
  if (recs_acquire<POLL>(recs, nch, lane, p.spin_ticks)) {
    PairState st;
    pair_fold<MASK>(recs, nch, lane, st);
    pair_logs(st, lane, lse, logZ);
  } else if (lane == 0 && p.err) {
    atomicAdd(p.err, 1u);
  }
  if (lane == 0) {
    if (p.out_lse) p.out_lse[pidx] = lse;
    if (p.out_logZ) p.out_logZ[pidx] = logZ;
  }
}

// ---------------------------------------------------------------------------------------------------------
// finish role with a Philox draw: one wave per particle.  Sweeps the records of the particle's row until they are
// complete (POLL), folds them, draws the chunk with the first Philox word and, with the second, walks down the chunk's
// summation tree: 16-lane row (the record carries the four row sums), lane, class, element.  Only the chosen ROW is
// reduced again - a quarter of the chunk, 4 KiB: finishing lane L = 4 l + w takes class w of the row's lane l (its 16
// elements, in the order the stats wave added them), so every (lane, class) partial is rebuilt bit for bit by one lane and
// the finishing lanes stand in the order of the summation tree - lane by lane, class by class inside a lane - : ONE
// inclusive scan of their payloads picks lane and class together (the first prefix above the target, exactly what picking
// the lane by its four-class total and then the class inside it gives: every sum is an integer sum).  The chosen (lane,
// class)'s running sum is rebuilt in order by 15 dependent DPP adds.
// The wave's serial chain is what the launch's last microseconds are made of (in-kernel stamps, DESIGN.md §5: fold 0.8 us,
// chunk pick 0.5, row pick + addresses 0.6, partials and picks 1.1 of 3.6 us from "records complete" to "token"), so
// only what the DRAW needs sits on it: the scale and the inclusive scan of the allowed sums (the scan's last lane is their
// total), the row pick in scalar registers; the sums of ALL elements and both logarithms - outputs, not inputs of the
// draw - are taken while the quarter chunk is on its way.
// Nothing per particle is ever written by the stats role: round 2 left 512 bytes of lane scans per chunk for this
// step - 2 us of a read-bound launch.
// ---------------------------------------------------------------------------------------------------------
// the row scale and the sum on it of ALL elements (off the draw's critical path); rows of up to 64 chunks: lane c = record c
__device__ __forceinline__ void fold_all_fast(const ChunkRec &r, bool have, float &N_all, uint64_t &S_all) {
  const uint64_t a = have ? ((uint64_t)r.pA << kGridHi) + r.pB : 0ull;
  N_all = wave_max(a ? r.Nc : kNegInf);
  uint64_t sa = 0;
  if (a) {
    const float d = N_all - r.Nc;  // >= 0, integer valued
    sa = d < 64.0f ? a >> (uint32_t)d : 0ull;
  }
  S_all = wave_sum_u64(sa);
}

template <int DT, int MASK, bool POLL, int EXPC>
__device__ __forceinline__ void finish_draw(const StepParams &p, int pidx, int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC, NVW = NVC / 4;
  const int nch = p.nch, V = p.V;
  const int pr = p.pair_of ? as_const(p.pair_of)[pidx] : pidx;
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const int mi = MASK == kMaskNone ? 0 : (p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr));
  GLB_DIAG(
    uint64_t *stamps = p.out_margin ? reinterpret_cast<uint64_t *>(p.out_margin) + (int64_t)pidx * 8 : nullptr;
    if (stamps && lane == 0) stamps[0] = GLB_NOW();
  )
  uint64_t R1, R2;
  philox_pair(p, pidx, R1, R2);  // (before the wait: the draws need nothing from the records)
  Recs recs;
  recs.base = p.recs + (int64_t)pr * nch * kRecWords;
  recs.epoch = p.epoch;
  if (!recs_acquire<POLL>(recs, nch, lane, p.spin_ticks)) {  // never in a healthy launch: say so in the outputs and leave
    if (lane == 0) {
      if (p.err) atomicAdd(p.err, 1u);
      const float nan = __uint_as_float(0x7fc00000u);
      if (p.out_lse) p.out_lse[pidx] = nan;
      if (p.out_logZ) p.out_logZ[pidx] = nan;
      p.out_token[pidx] = -2;
    }
    return;
  }
  GLB_DIAG(if (stamps && lane == 0) stamps[1] = GLB_NOW();)
  PairState st;
  ChunkPick pick{-1, 0.f, R2, {0ull, 0ull, 0ull, 0ull}};
  const bool fast = recs.cached;  // rows of up to 64 chunks: lane c holds record c
  if (fast) {
    // ---- the draw's side of the fold: scale and inclusive scan of the allowed sums; the scan's last lane is their total
    const ChunkRec &r = recs.mine;
    const bool have = lane < nch;
    const uint64_t mm = have ? ((uint64_t)r.pAm << kGridHi) + r.pBm : 0ull;
    st.N_msk = wave_max(mm ? r.Nm : kNegInf);
    uint64_t sm = 0;
    if (mm) {
      const float d = st.N_msk - r.Nm;
      sm = d < 64.0f ? mm >> (uint32_t)d : 0ull;
    }
    const uint64_t incl = wave_scan_u64(sm);
    st.S_msk = readlane_u64(incl, 63);
    GLB_DIAG(if (stamps && lane == 0) stamps[4] = GLB_NOW();)
    uint32_t nz = (uint32_t)st.S_msk | (uint32_t)(st.S_msk >> 32);
    opaque_u32(nz);
    if (nz != 0u) {
      const uint64_t T = __umul64hi(R1, st.S_msk);  // uniform integer in [0, S_msk)
      const int lsel = first_lane_above(incl, T);
      pick.csel = lsel;
      pick.Nms = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(r.Nm), lsel));
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pick.qg[g] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r.rA[g], lsel) << kGridHi) +
                     (uint32_t)__builtin_amdgcn_readlane((int)r.rB[g], lsel);
    }
  } else {
    pair_fold<MASK>(recs, nch, lane, st);
    GLB_DIAG(if (stamps && lane == 0) stamps[4] = GLB_NOW();)
    pick = pair_pick_chunk(recs, st, R1, R2, nch, lane);
  }
  GLB_DIAG(if (stamps && lane == 0) stamps[5] = GLB_NOW();)
  int32_t tok = -1;
  if (pick.csel >= 0) {
    // ---- the 16-lane row: four wave-uniform sums, picked in scalar registers
    const uint64_t i0 = pick.qg[0], i1 = i0 + pick.qg[1], i2 = i1 + pick.qg[2], Sc = i2 + pick.qg[3];
    const uint64_t T2 = __umul64hi(pick.R2, Sc);  // uniform integer in [0, S_c)
    const int gsel = T2 < i0 ? 0 : (T2 < i1 ? 1 : (T2 < i2 ? 2 : 3));  // (S_c > 0: the chunk was drawn)
    const uint64_t Tg = T2 - (gsel == 0 ? 0ull : (gsel == 1 ? i0 : (gsel == 2 ? i1 : i2)));
    // ---- this lane's (row lane, class): 16 elements; finishing lane L = 4 l + w
    const int c = pick.csel, e_base = c * kChunk;
    const int w = lane & 3, sl = 16 * gsel + (lane >> 2);  // class, lane of the stats wave
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    const float magicN = term_bias<EXPC>(pick.Nms);
    float y[16];
    const bool whole = e_base + kChunk <= V;  // wave-uniform: every chunk of a row but its last - no per-lane guards
#pragma unroll
    for (int j = 0; j < NVW; ++j) {
      const int iv = w + 4 * j;
      const int e0 = e_base + (iv * 64 + sl) * EPV;
      const u32x4_t r = whole ? *reinterpret_cast<const u32x4_t *>(rowp + (int64_t)e0 * ES)
                              : load_vec_guarded<DT>(rowp, e0 < V ? e0 : V, V);
      unpack_vec<DT>(r, &y[j * EPV]);
#pragma unroll
      for (int k = 0; k < EPV; ++k) y[j * EPV + k] *= p.scale;  // (x * 1.0f == x: the scaled form serves every call)
      if constexpr (MASK == kMaskBits) {
        // (the 32-bit half of the lane word that holds this lane's bit: one register per element in flight, not two)
        const uint32_t *mw = reinterpret_cast<const uint32_t *>(p.mask_t + ((int64_t)mi * nch + c) * 64 + iv * EPV) + (sl >> 5);
#pragma unroll
        for (int k = 0; k < EPV; ++k)
          if (!((mw[2 * k] >> (sl & 31)) & 1u)) y[j * EPV + k] = kNegInf;
      } else if constexpr (MASK == kMaskRaw) {  // the caller's bit row: the vector's EPV bits sit in one word
        const int wi = e0 >> 5;
        const uint32_t wbits = wi < ((V + 31) >> 5) ? (p.mask_bits + (int64_t)mi * p.mask_bits_ld)[wi] : 0u;
#pragma unroll
        for (int k = 0; k < EPV; ++k)
          if (!((wbits >> ((e0 & 31) + k)) & 1u)) y[j * EPV + k] = kNegInf;
      } else if constexpr (MASK == kMaskF32) {
        const char *mrow = (const char *)(p.mask_f + (int64_t)mi * p.mask_ld);
#pragma unroll
        for (int h = 0; h < EPV / 4; ++h) {
          const int eh = e0 + 4 * h;
          const u32x4_t mr = load_vec_guarded<kDtF32>(mrow, eh < V ? eh : V, V);
          float mk[4];
          unpack_vec<kDtF32>(mr, mk);
#pragma unroll
          for (int k = 0; k < 4; ++k) y[j * EPV + 4 * h + k] = y[j * EPV + 4 * h + k] + mk[k];
        }
      }
    }
    GLB_DIAG(if (stamps && lane == 0) stamps[6] = GLB_NOW();)
    {  // while the quarter chunk is on its way: the sums of all elements and the two logarithms (outputs, not inputs of the draw)
      if (fast) {
        if constexpr (MASK == kMaskNone) st.N_all = st.N_msk, st.S_all = st.S_msk;
        else fold_all_fast(recs.mine, lane < nch, st.N_all, st.S_all);
      }
      float lse, logZ;
      pair_logs(st, lane, lse, logZ);
      if (lane == 0) {
        if (p.out_lse) p.out_lse[pidx] = lse;
        if (p.out_logZ) p.out_logZ[pidx] = logZ;
      }
    }
    GLB_DIAG(
      if (stamps && lane == 0) stamps[7] = GLB_NOW();
      asm volatile("" ::"v"(y[15]));
      if (stamps && lane == 0) stamps[3] = GLB_NOW();
    )
    float t[16], P = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      t[j] = chunk_term<EXPC>(y[j], magicN);
      P = P + t[j];
    }
    uint32_t h, l;
    partial_q(P, h, l);
    // ---- lane and class at once: the finishing lanes stand in summation order, one scan of their payloads
    const uint64_t qown = ((uint64_t)h << kGridHi) + l;  // q of (row lane lane >> 2, class lane & 3)
    const uint64_t incl2 = wave_scan_u64(qown);
    const int Lsel = first_lane_above(incl2, Tg);
    if (Lsel >= 0) {
      const uint64_t Tw = Tg - (Lsel > 0 ? readlane_u64(incl2, Lsel > 0 ? Lsel - 1 : 0) : 0ull);
      const int lsel = Lsel >> 2, wsel = Lsel & 3;
      // ---- the element: lane pos < 16 gets term pos of finishing lane Lsel, then the running sum in order
      float tv = 0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float tj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t[j]), Lsel));
        tv = (lane & 15) == j ? tj : tv;
      }
      float cs = tv;
#pragma unroll
      for (int s2 = 1; s2 < 16; ++s2) cs = __uint_as_float(dpp_u32<0x111, 0xf>(0u, __float_as_uint(cs))) + tv;
      uint32_t hc, lc;
      partial_q(cs, hc, lc);
      const uint64_t qc = lane < 16 ? ((uint64_t)hc << kGridHi) + lc : 0ull;
      const int psel = first_lane_above(qc, Tw);
      if (psel >= 0) tok = e_base + ((wsel + 4 * (psel / EPV)) * 64 + 16 * gsel + lsel) * EPV + (psel % EPV);
    }
  } else {
    if (fast) {
      if constexpr (MASK == kMaskNone) st.N_all = st.N_msk, st.S_all = st.S_msk;
      else fold_all_fast(recs.mine, lane < nch, st.N_all, st.S_all);
    }
    float lse, logZ;
    pair_logs(st, lane, lse, logZ);
    if (lane == 0) {
      if (p.out_lse) p.out_lse[pidx] = lse;
      if (p.out_logZ) p.out_logZ[pidx] = logZ;
    }
  }
  if (lane == 0) p.out_token[pidx] = tok;
  GLB_DIAG(if (stamps && lane == 0) stamps[2] = GLB_NOW();)
}

// ---------------------------------------------------------------------------------------------------------
// The step in ONE launch of one-wave workgroups: blocks [0, stats_blocks) are stats waves, the blocks behind them
// finishing waves.  (One wave per workgroup: a four-wave workgroup gives its slots back only when its slowest wave is
// done - measured, 3 us of the fp32 launch and 5 us of the bf16 one.)  Workgroups are dispatched in index order, so
// finishing waves become resident when the last stats waves have been placed; correctness needs no such order: a
// finishing wave only waits for stats waves, which wait for nobody, and there are never more finishing waves than half
// the chip's wave slots (glb_api.hip), so they cannot keep a stats wave from getting one.  Every wait is bounded
// (kSpinTicks).
// ---------------------------------------------------------------------------------------------------------
// waves per SIMD the kernels are compiled for.  Float masks hold x[64] and y[64] in the stats role: three leave them
// the registers.  16-bit rows under the polynomial contract are bound by VALU issue and gain from a fifth wave (96
// registers fit without spills); fp32 rows spill at five and stream as fast with four.  Under the hardware-exponential
// contract (kExpHw) 16-bit rows are bound by memory, not issue: four and five waves measure the same, six spill
// (profiles/r06/ab_contract_*), and a tenth fewer vector instructions a wave change nothing (ab_libs_valu_v5.log).
template <int DT, int MASK>
struct StatsWaves {
  static constexpr int value = MASK == kMaskF32 ? 3 : (DT == kDtF32 ? 4 : GLB_STATS_WAVES_16);
};

// (Round 6, under the hardware-exponential contract, where the stream is bound by memory: workgroups of two and four waves
// again - 27.2 / 27.3 us against 26.6 at 512 x 128256 bf16 -, two chunks per wave - 30 to 34 us: the doubled body spills -,
// the chunk kept as loaded in 32 registers at six and eight waves a SIMD - no gain, spills at eight -, default-policy loads
// - 28.7 us; and VERDICT r5 #7's proposal, the stats wave making the in-chunk draw itself for units of one particle and
// leaving the token in a thirteenth granule (commit 42bad02: bit-exact, the finishing wave 1.7 us shorter - and the launch
// 4.4 us LONGER at 1024 x 50257 fp32, 5.7 us at 512 x 128256 bf16: every stats wave lives a Philox block, two scans and
// sixteen terms longer, and the stream is bound by how many waves have loads in flight); EXPERIMENTS.md.
// Kept at the end of the round: the (unit, chunk) quotient as a scalar, the record gathered in the lanes by DPP moves and
// stored off a scalar base, the wave maximum as DPP v_max_f32 - no spill in the hot path of any form, ~50 vector
// instructions a wave fewer; raw bit rows read by the stats waves themselves (kMaskRaw).)
template <int DT, int MASK, bool SCALED, int MODE, int EXPC = kExpPoly>
__global__ __launch_bounds__(64, (StatsWaves<DT, MASK>::value)) void fused_step_kernel(const StepParams p) {
  const int lane = threadIdx.x;
  const int blk = blockIdx.x;
  int item = blk < p.stats_blocks ? blk : -1, pfirst = blk - p.stats_blocks;
  GLB_DIAG(
    if (p.diag.il_lag >= 0) {
      const int head = p.diag.il_lag * p.nch, b = blk - head;
      item = blk;
      if (b >= 0) {
        const int per = p.nch + 1, n_mid = p.n_pairs - p.diag.il_lag;
        const int q = (int)((unsigned)b / (unsigned)per), r = b - q * per;
        item = q < n_mid && r < p.nch ? (p.diag.il_lag + q) * p.nch + r : -1;
        pfirst = q < n_mid ? q : n_mid + (b - n_mid * per);
      }
    }
  )
  if (item >= 0) {
    stats_item<DT, MASK, SCALED, EXPC>(p, item, lane);
    return;
  }
  for (int pidx = pfirst; pidx < p.n_particles; pidx += p.fin_blocks) {
    if constexpr (MODE == kModePhilox) finish_draw<DT, MASK, true, EXPC>(p, pidx, lane);
    else finish_stats<MASK, true>(p, pidx, lane);
  }
}

template <int DT, int MASK, bool SCALED, int EXPC = kExpPoly>
__global__ __launch_bounds__(64, (StatsWaves<DT, MASK>::value)) void chunk_stats_kernel(const StepParams p) {
  stats_item<DT, MASK, SCALED, EXPC>(p, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// finish as a launch of its own (records of an earlier launch, tags not looked at): one workgroup per particle - one
// wave for the statistics / Philox modes (launched with 64 threads), four for parity mode, which deals the row's vectors
// to all of them.
// ---------------------------------------------------------------------------------------------------------
template <int DT, int MASK, int MODE, int EXPC = kExpPoly>
__global__ __launch_bounds__(256) void finish_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  __shared__ float s_bestg[4], s_secg[4];
  __shared__ int32_t s_bestj[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pidx = blockIdx.x;
  if constexpr (MODE == kModeStats) {
    if (wave == 0) finish_stats<MASK, false>(p, pidx, lane);
    return;
  }
  if constexpr (MODE == kModePhilox) {
    if (wave == 0) {
      if (p.out_token) finish_draw<DT, MASK, false, EXPC>(p, pidx, lane);
      else finish_stats<MASK, false>(p, pidx, lane);
    }
    return;
  }
  const int nch = p.nch, V = p.V;
  const int pr = p.pair_of ? as_const(p.pair_of)[pidx] : pidx;
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const int mi = MASK == kMaskNone ? 0 : (p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr));
  Recs recs;
  recs.base = p.recs + (int64_t)pr * nch * kRecWords;
  recs.epoch = p.epoch;
  recs_acquire<false>(recs, nch, lane, 0);

  RowView<DT, MASK> rv;
  rv.rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  rv.V = V;
  rv.scale = p.scale;
  rv.mt = MASK == kMaskBits ? p.mask_t + (int64_t)mi * nch * 64 : nullptr;
  rv.many = MASK == kMaskBits ? p.mask_any + (int64_t)mi * nch : nullptr;
  rv.mrow = MASK == kMaskF32 ? (const char *)(p.mask_f + (int64_t)mi * p.mask_ld) : nullptr;

  PairState st;
  pair_fold<MASK>(recs, nch, lane, st);
  if (wave == 0) {
    float lse, logZ;
    pair_logs(st, lane, lse, logZ);
    if (lane == 0) {
      if (p.out_lse) p.out_lse[pidx] = lse;
      if (p.out_logZ) p.out_logZ[pidx] = logZ;
    }
  }
  if (!p.out_token) return;
  {
    // ---- parity mode: exponential race against the caller's noise, first maximum of e_j / E_j (README.md:87
    //      through torch.multinomial's CPU algorithm).  e_j = ldexp(P, n_j - N_msk): any common scale gives the same
    //      comparisons; N_msk is the largest scale any chunk's allowed terms were summed on, so every allowed term that
    //      can win is representable.  Vectors are dealt round
    //      robin to the four waves; ties resolve to the smallest index at every level ------------------------------
    int32_t tok = -1;
    if (st.S_msk != 0) {  // workgroup-uniform: every wave folded the same records
      const float magicN = term_bias<EXPC>(st.N_msk);
      const float *E = p.noise + (int64_t)pidx * p.noise_ld;
      float best = -1.0f, sec = -1.0f;  // sec: runner-up of the race (for the reported margin)
      int32_t bj = -1;
      constexpr int NVW = NVC / 4;  // this wave's vectors of a chunk: all their loads (logits, mask, noise) go out first
      for (int c = 0; c < nch; ++c) {
        float y[NVW][EPV];
        int e0[NVW];
        u32x4_t en[NVW][EPV / 4];
#pragma unroll
        for (int jv = 0; jv < NVW; ++jv) {
          const int i = wave + 4 * jv;
          e0[jv] = V;
          if (c * kChunk + i * 64 * EPV < V) {  // wave-uniform
            rv.vec(c, i, lane, y[jv], e0[jv]);
#pragma unroll
            for (int h = 0; h < EPV / 4; ++h) {
              const int eh = e0[jv] + 4 * h;
              en[jv][h] = load_vec_guarded<kDtF32>((const char *)E, eh < V ? eh : V, V);
            }
          }
        }
#pragma unroll
        for (int jv = 0; jv < NVW; ++jv) {
          if (c * kChunk + (wave + 4 * jv) * 64 * EPV >= V) break;
#pragma unroll
          for (int k = 0; k < EPV; ++k) {
            const int j = e0[jv] + k;
            const bool ok = j < V && y[jv][k] > kNegInf;
            const float e = chunk_term<EXPC>(y[jv][k], magicN);
            const uint32_t eb = k % 4 == 0 ? en[jv][k / 4].x : (k % 4 == 1 ? en[jv][k / 4].y : (k % 4 == 2 ? en[jv][k / 4].z : en[jv][k / 4].w));
            const float g = ok ? e / __uint_as_float(eb) : -1.0f;
            const bool better = g > best;  // strict: the first maximum in this lane's (increasing) order stays
            sec = better ? best : fmaxf(sec, g);
            best = better ? g : best;
            bj = better ? j : bj;
          }
        }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float og = __shfl_xor(best, o, 64), os = __shfl_xor(sec, o, 64);
        const int32_t oj = __shfl_xor(bj, o, 64);
        const bool take = og > best || (og == best && oj >= 0 && (bj < 0 || oj < bj));
        sec = fmaxf(fmaxf(sec, os), take ? best : og);
        best = take ? og : best;
        bj = take ? oj : bj;
      }
      if (lane == 0) {
        s_bestg[wave] = best;
        s_secg[wave] = sec;
        s_bestj[wave] = bj;
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float og = s_bestg[w], os = s_secg[w];
          const int32_t oj = s_bestj[w];
          const bool take = og > best || (og == best && oj >= 0 && (bj < 0 || oj < bj));
          sec = fmaxf(fmaxf(sec, os), take ? best : og);
          best = take ? og : best;
          bj = take ? oj : bj;
        }
        // how far the draw is from a tie: (winner - runner-up) / winner; 1 when there is no runner-up
        if (lane == 0 && p.out_margin) p.out_margin[pidx] = sec < 0.0f ? 1.0f : (best - sec) / best;
      }
      tok = bj;
    }
    if (wave == 0 && lane == 0) p.out_token[pidx] = tok;
  }
}

// ---------------------------------------------------------------------------------------------------------
// log-probability rows: out[r, j] = x'[r, j] - lse[r]  (one thread per 16-byte input vector).  The last of three
// launches (chunk statistics, one wave per row for lse, this) that serve what the kernel of independent waves below
// does not: rows of more than 64 chunks, workspaces without tags, stream capture.
// ---------------------------------------------------------------------------------------------------------
template <int DT, bool OUT16>
__global__ __launch_bounds__(256) void logprob_rows_kernel(const void *logits, int64_t ld, int V, float scale,
                                                          const float *lse, void *out_v, int64_t out_ld,
                                                          int n_rows) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES;
  const int nv = (V + EPV - 1) / EPV;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)n_rows * nv) return;
  const int r = (int)(gid / nv), e0 = (int)(gid % nv) * EPV;
  const char *rowp = (const char *)logits + (int64_t)r * ld * ES;
  float x[EPV];
  unpack_vec<DT>(load_vec_guarded<DT>(rowp, e0, V), x);
  const float l = lse[r];
#pragma unroll
  for (int k = 0; k < EPV; ++k) x[k] = x[k] * scale - l;  // (x * 1.0f == x: the scaled form serves every call)
  if constexpr (OUT16) {
    uint16_t *o = reinterpret_cast<uint16_t *>(out_v) + (int64_t)r * out_ld + e0;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = pack16<DT>(x[2 * k], x[2 * k + 1]);
    if (e0 + EPV <= V) *reinterpret_cast<u32x4_t *>(o) = u32x4_t{w[0], w[1], w[2], w[3]};
    else
      for (int k = 0; k < V - e0; ++k) o[k] = (uint16_t)(w[k >> 1] >> ((k & 1) * 16));
  } else {
    float *o = reinterpret_cast<float *>(out_v) + (int64_t)r * out_ld + e0;
    if (e0 + EPV <= V) {
#pragma unroll
      for (int h = 0; h < EPV / 4; ++h) {
        float4 v;
        v.x = x[4 * h];
        v.y = x[4 * h + 1];
        v.z = x[4 * h + 2];
        v.w = x[4 * h + 3];
        *reinterpret_cast<float4 *>(o + 4 * h) = v;
      }
    } else {
      for (int k = 0; k < V - e0; ++k) o[k] = x[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// log-probability rows, one HBM reading, no workgroup structure: one-wave workgroups, block b = (row b / wpr, wave
// b % wpr of the row), a wave owning CPW consecutive chunks.  A wave loads its chunks and keeps them in registers as
// loaded, reduces each and publishes (scale, sum) as three tagged granules of the chunk's record (words 0, 2, 3 - the
// step kernels' layout).  The row's first wave then sweeps the row's records until every tag is this call's epoch (its
// row-mates are its neighbours in the grid: they were placed within microseconds of it), folds lse and publishes it as
// a fourth tagged granule; the other waves poll that one granule; every wave writes x - lse from its registers.
// Waves of different rows are at different points of this at any moment, so reading, arithmetic and writing overlap
// across the chip by themselves; a workgroup per row with the row in its registers (tried: one 1024-thread workgroup
// per CU walking its share of the rows) marches in step with all the others - every CU reads, then every CU computes,
// then every CU writes: 104 us at 1024 x 50257 fp32 where the bytes alone take 70; this kernel: 74 us.  A wave waits only
// for waves placed before or right behind it (the oldest unfinished row of the grid has all its waves resident, on
// whatever XCD they were placed: it completes), the wait is bounded (NaN out after the watchdog's ticks), rows of up to
// 64 chunks.
// ---------------------------------------------------------------------------------------------------------
// STORE (16-bit rows; a lane's eight elements of a vector are 32 bytes of output): 0 = two 16-byte stores per lane, each
// instruction covering every other 16 bytes of a 2 KB span; 2 = through 2 KB of LDS so that each store instruction
// writes 1 KB of consecutive addresses.
// OUT16 (16-bit rows only): the log-probabilities leave in the logits' own element type, as the reference returns them
// (cache.py:96 keeps the dtype) - the float32 result rounded to nearest even; a lane's eight outputs of a vector are one
// 16-byte store and every store instruction writes 1 KB of consecutive addresses, so nothing passes through LDS.
// CPW: chunks per wave.  With one chunk per wave, what a wave does for its chunk takes 4 us at 512 x 128256 bf16 and its
// wait for the row 13 us (tools/dbg/stamps_lsm.py, profiles/r04/stamps_lsm_before_v4.log: the row's last record is out
// 6 us after the average one, the first wave has folded 5 us later, the row-mates have the result 1.7 us after that):
// four resident waves in five are waiting and the launch runs at the rate the remaining fifth can load - 67 us where
// the same kernel without the wait takes 46 (profiles/r04/ab_lsm_nowait_nostore_v4.log).  A 16-bit chunk is 8 KB, half
// the registers of a float32 chunk: two or three chunks per wave put that many times the bytes behind every wait and
// divide the waves a row waits for (52 us; the launcher picks CPW by the row's length).
// a loaded vector as a value the compiler knows nothing about: what it unpacked from the vector before is not carried
// over (16-bit rows: the unpacked form of a chunk is 64 registers a lane, the loaded form 32 - kept across the passes
// of a wave it is the difference between five waves a SIMD with spills and without)
template <int DT>
__device__ __forceinline__ u32x4_t as_loaded(u32x4_t v) {
  if constexpr (DT != kDtF32) asm volatile("" : "+v"(v));  // (float32: the loaded form is the unpacked form)
  return v;
}

template <int DT, bool SCALED>
__device__ __forceinline__ void chunk_scale_sum(const u32x4_t (&raw)[ElemTraits<DT>::NVC], int nv_valid, float scale, float &Nc,
                                                uint32_t &pA, uint32_t &pB) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  float m = kNegInf;
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    float t[EPV];
    unpack_scaled<DT, SCALED>(raw[i], scale, t);
#pragma unroll
    for (int k = 0; k < EPV; k += 2) m = max3(m, t[k], t[k + 1]);
  }
  Nc = exp_n(wave_max(m));
  const float magicN = kMagic - Nc;
  float P[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    if (i < nv_valid) {  // wave-uniform; vectors wholly past the row end would add +0
      float t[EPV];
      unpack_scaled<DT, SCALED>(as_loaded<DT>(raw[i]), scale, t);
#pragma unroll
      for (int k = 0; k < EPV; ++k) t[k] = chunk_term(t[k], magicN);
#pragma unroll
      for (int k = 0; k < EPV; ++k) P[i & 3] = P[i & 3] + t[k];
    }
    __builtin_amdgcn_sched_barrier(0);  // vector by vector: only the packed form stays live
  }
  uint32_t h, l;
  lane_payload<4>(P, h, l);
  pA = last_lane(wave_sum_u32_l63(h));
  pB = last_lane(wave_sum_u32_l63(l));
}

// x - lse of one chunk, from the registers it was loaded into
template <int DT, bool SCALED, int STORE, bool OUT16>
__device__ __forceinline__ void chunk_store_logprobs(const u32x4_t (&raw)[ElemTraits<DT>::NVC], float scale, float lse, void *out_v,
                                                     int64_t out_off, int e_base, int V, int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  if constexpr (OUT16) {
    uint16_t *o16 = reinterpret_cast<uint16_t *>(out_v) + out_off + e_base + lane * 8;
    const bool full = e_base + kChunk <= V;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const u32x4_t x = as_loaded<DT>(raw[i]);
      const uint32_t w[4] = {logprob_word16<DT, SCALED>(x.x, scale, lse), logprob_word16<DT, SCALED>(x.y, scale, lse),
                             logprob_word16<DT, SCALED>(x.z, scale, lse), logprob_word16<DT, SCALED>(x.w, scale, lse)};
      const int e0 = e_base + (i * 64 + lane) * 8;
      if (full || e0 + 8 <= V) {
        // (rows are element aligned only: a 16-byte store to an address that is not 16-byte aligned is legal on gfx950 and
        // what the unaligned loads of the same rows already do)
        __builtin_nontemporal_store(u32x4_t{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4_t *>(o16 + i * 512));
      } else {
        for (int k = 0; k < V - e0; ++k) o16[i * 512 + k] = (uint16_t)(w[k >> 1] >> ((k & 1) * 16));
      }
    }
  } else {
    float *out = reinterpret_cast<float *>(out_v) + out_off;
    float *o = out + e_base + lane * EPV;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    if (e_base + kChunk <= V) {  // wave-uniform: a full chunk, straight stores
      if constexpr (EPV == 8 && STORE == 2) {
        __shared__ f32x4_t s_t[2][128];  // two vectors' worth: the next one is written while the last one's reads land
        float *ot = out + e_base + lane * 4;
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
          float t[EPV];
          unpack_scaled<DT, SCALED>(as_loaded<DT>(raw[i]), scale, t);
          f32x4_t *buf = s_t[i & 1];
          buf[2 * lane] = f32x4_t{t[0] - lse, t[1] - lse, t[2] - lse, t[3] - lse};
          buf[2 * lane + 1] = f32x4_t{t[4] - lse, t[5] - lse, t[6] - lse, t[7] - lse};
          const f32x4_t a = buf[lane], b = buf[64 + lane];  // same wave: program order holds within LDS
          __builtin_nontemporal_store(a, reinterpret_cast<f32x4_t *>(ot + i * 512));
          __builtin_nontemporal_store(b, reinterpret_cast<f32x4_t *>(ot + i * 512 + 256));
        }
      } else {
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
          float t[EPV];
          unpack_scaled<DT, SCALED>(as_loaded<DT>(raw[i]), scale, t);
#pragma unroll
          for (int hh = 0; hh < EPV / 4; ++hh) {
            const f32x4_t v{t[4 * hh] - lse, t[4 * hh + 1] - lse, t[4 * hh + 2] - lse, t[4 * hh + 3] - lse};
            if constexpr (STORE == 1) *reinterpret_cast<f32x4_t *>(o + i * 64 * EPV + 4 * hh) = v;
            else __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t *>(o + i * 64 * EPV + 4 * hh));
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NVC; ++i) {
        const int e0 = e_base + (i * 64 + lane) * EPV;
        float t[EPV];
        unpack_scaled<DT, SCALED>(as_loaded<DT>(raw[i]), scale, t);
        for (int k = 0; k < EPV; ++k)
          if (e0 + k < V) o[i * 64 * EPV + k] = t[k] - lse;
      }
    }
  }
}

// grid: n_rows * wpr blocks, wpr = ceil(nch / CPW) waves per row
template <int DT, bool SCALED, int WPS, int STORE, bool OUT16 = false, int CPW = 1>
__global__ __launch_bounds__(64, WPS) void logprob_rows_waves_kernel(
    const void *logits, int64_t ld, int V, int nch, int wpr, float scale, void *out_v, int64_t out_ld, float *out_lse,
    uint64_t *recs, uint32_t epoch, uint32_t *err, uint64_t spin_ticks) {
  static_assert(!OUT16 || DT != kDtF32, "16-bit output goes with 16-bit logits");
  static_assert(CPW == 1 || DT != kDtF32, "a float32 chunk fills the registers by itself");
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  const int lane = threadIdx.x;
  GLB_DIAG(const uint64_t stamp0 = GLB_NOW();)
  const int r = (int)(blockIdx.x / (unsigned)wpr), w = (int)(blockIdx.x - (unsigned)r * (unsigned)wpr);
  const int c0 = w * CPW;
  const char *rowp = (const char *)logits + (int64_t)r * ld * ES;
  u32x4_t raw[CPW][NVC];
#pragma unroll
  for (int j = 0; j < CPW; ++j)
    if (j == 0 || c0 + j < nch) load_chunk_raw<DT>(rowp, (c0 + j) * kChunk, V, lane, raw[j]);  // wave-uniform
  uint64_t *row_recs = recs + (int64_t)r * nch * kRecWords;
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    if (j == 0 || c0 + j < nch) {
      const int e_base = (c0 + j) * kChunk;
      int nv_valid = (V - e_base + 64 * EPV - 1) / (64 * EPV);
      nv_valid = nv_valid < NVC ? nv_valid : NVC;
      float Nc;
      uint32_t pA, pB;
      chunk_scale_sum<DT, SCALED>(raw[j], nv_valid, scale, Nc, pA, pB);
      uint32_t v = __float_as_uint(Nc);
      v = lane == 2 ? pA : v;
      v = lane == 3 ? pB : v;
      if (lane == 0 || lane == 2 || lane == 3)
        __hip_atomic_store(row_recs + (int64_t)(c0 + j) * kRecWords + lane, ((uint64_t)epoch << 32) | v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  GLB_DIAG(
    uint64_t *stamps = row_recs + (int64_t)c0 * kRecWords + 12;
    if (lane == 0) stamps[0] = stamp0, stamps[1] = GLB_NOW();
  )
  // The row's FIRST wave (placed before its row-mates) sweeps the records - lane j takes record j: the three granules in
  // one sweep, so the sweep that finds every tag in place already holds the values -, folds lse and publishes it as one
  // more tagged granule (word 1 of chunk 0's record); the other waves of the row wait for that one granule: one 8-byte
  // poll per sweep instead of nch of them, and one pass through the double-precision logarithm per row instead of one per
  // wave (fp64 instructions issue at a fraction of the fp32 rate).
  float lse = __builtin_nanf("");
  const uint64_t t0 = GLB_NOW();
  if (w != 0) {
    for (;;) {
      const uint64_t g = __hip_atomic_load(row_recs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(g >> 32) == epoch) {
        lse = __uint_as_float((uint32_t)g);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      if (GLB_NOW() - t0 > spin_ticks) {
        if (lane == 0 && err) atomicAdd(err, 1u);  // (the chunks come out as NaN: a failed launch, never a result)
        break;
      }
    }
  } else {
    float Nj = kNegInf;
    uint64_t Sj = 0;
    bool have = true;
    const uint64_t *q = row_recs + (int64_t)(lane < nch ? lane : 0) * kRecWords;
    for (;;) {
      bool ok = true;
      if (lane < nch) {
        const uint64_t g0 = __hip_atomic_load(const_cast<uint64_t *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t g2 = __hip_atomic_load(const_cast<uint64_t *>(q) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t g3 = __hip_atomic_load(const_cast<uint64_t *>(q) + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (uint32_t)(g0 >> 32) == epoch && (uint32_t)(g2 >> 32) == epoch && (uint32_t)(g3 >> 32) == epoch;
        Nj = __uint_as_float((uint32_t)g0);
        Sj = ((uint64_t)(uint32_t)g2 << kGridHi) + (uint32_t)g3;
      }
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
      __builtin_amdgcn_s_sleep(8);
      if (GLB_NOW() - t0 > spin_ticks) {
        have = false;
        break;
      }
    }
    if (!have && lane == 0 && err) atomicAdd(err, 1u);  // (the row comes out as NaN: a failed launch, never a result)
    if (have) {
      const bool on = lane < nch && Sj != 0;
      const float N = wave_max(on ? Nj : kNegInf);
      uint64_t sa = 0;
      if (on) {
        const float d = N - Nj;
        sa = d < 64.0f ? Sj >> (uint32_t)d : 0ull;
      }
      const uint64_t S = wave_sum_u64(sa);
      lse = S ? (float)log_fix(S, (int32_t)N + 1 - kFrac) : kNegInf;
    }
    if (lane == 0) {  // (a NaN too: the row-mates then leave with it instead of waiting out their own watchdog)
      __hip_atomic_store(row_recs + 1, ((uint64_t)epoch << 32) | __float_as_uint(lse), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (out_lse) out_lse[r] = lse;
    }
  }
  GLB_DIAG(if (lane == 0) stamps[2] = GLB_NOW();)
  if (out_v) {
    // (the lane index as a new value: element indices of the ragged last chunk's stores are otherwise computed next to
    // those of its loads, ahead of the wait, and spilled across it)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
#pragma unroll
    for (int j = 0; j < CPW; ++j)
      if (j == 0 || c0 + j < nch)
        chunk_store_logprobs<DT, SCALED, STORE, OUT16>(raw[j], scale, lse, out_v, (int64_t)r * out_ld, (c0 + j) * kChunk, V, lane_s);
  }
  GLB_DIAG(
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) stamps[3] = GLB_NOW();
  )
}

// ---------------------------------------------------------------------------------------------------------
// mask preparation: bit rows [n_masks, mask_ld] -> transposed lane words.  One WAVE per (chunk c, mask k).  Lane l
// gathers the 64 bits of ITS elements of the chunk - bit j = i * EPV + kk of X[l] is element (i * 64 + l) * EPV + kk; the
// EPV bits of a vector sit in one word of the row (fp32: bits 4 (i*64 + l) .. +3 of the chunk; 16-bit: one byte), so that
// is NVC loads served by the L1 - and the wave transposes the 64 x 64 bit matrix in six butterfly exchanges: lane word
// j (bit l = lane l's element j allowed) ends up in lane j, and the wave leaves with ONE coalesced 512-byte store.
// (Round 3's form - 256-thread workgroups, a ballot per (vector, component), lane 0 of every wave storing its words one
// by one - took 40 us for 1024 masks of 50257 tokens, as long as the step that reads them; one-wave workgroups with a
// ballot per word 18 us; the butterfly does 64 words' worth of ballots in 6 x ~14 instructions.)
// ---------------------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ uint64_t bit_transpose_step(uint64_t x, int lane) {
  constexpr uint64_t low = S == 32 ? 0x00000000ffffffffull : S == 16 ? 0x0000ffff0000ffffull : S == 8 ? 0x00ff00ff00ff00ffull
                         : S == 4 ? 0x0f0f0f0f0f0f0f0full : S == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
  const uint32_t tl = (uint32_t)__shfl_xor((int)(uint32_t)x, S, 64), th = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), S, 64);
  const uint64_t t = ((uint64_t)th << 32) | tl;
  return (lane & S) ? ((x & ~low) | ((t & ~low) >> S)) : ((x & low) | ((t & low) << S));
}

template <int EPV>
__global__ __launch_bounds__(64) void mask_prepare_kernel(const uint32_t *bits, int64_t mask_ld, int V, int nch,
                                                         uint64_t *mask_t, uint64_t *mask_any, const int32_t *which, int n_masks) {
  constexpr int NVC = 64 / EPV;
  // which: the masks to (re)prepare - only the rows whose mask changed since the last call (glb_mask_prepare_rows); null: all
  const int k = which ? as_const(which)[blockIdx.y] : (int)blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
  if (k < 0 || k >= n_masks) return;
  const uint32_t *row = bits + (int64_t)k * mask_ld;
  uint64_t *dst = mask_t + ((int64_t)k * nch + c) * 64;
  const int n_words = (V + 31) >> 5;
  uint32_t w[NVC];
#pragma unroll
  for (int i = 0; i < NVC; ++i) {  // the word that holds this lane's EPV bits of vector i (loads first)
    const int wi = (c * kChunk + (i * 64 + lane) * EPV) >> 5;
    w[i] = wi < n_words ? row[wi] : 0u;
  }
  uint64_t x = 0;
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    const int e0 = c * kChunk + (i * 64 + lane) * EPV;
    const int valid = V - e0;  // elements of this vector that exist: the rest read as forbidden
    uint32_t v = (w[i] >> (e0 & 31)) & ((1u << EPV) - 1u);
    if (valid < EPV) v &= valid > 0 ? ((1u << valid) - 1u) : 0u;
    x |= (uint64_t)v << (i * EPV);
  }
  // 64 x 64 bit-matrix transpose across the wave: exchange the off-diagonal s x s blocks, s = 32 .. 1
  x = bit_transpose_step<32>(x, lane);
  x = bit_transpose_step<16>(x, lane);
  x = bit_transpose_step<8>(x, lane);
  x = bit_transpose_step<4>(x, lane);
  x = bit_transpose_step<2>(x, lane);
  x = bit_transpose_step<1>(x, lane);
  dst[lane] = x;  // lane j holds lane word j = (vector j / EPV, component j % EPV)
  const uint64_t any = __ballot(x != 0ull);
  if (lane == 0) mask_any[(int64_t)k * nch + c] = any ? 1ull : 0ull;
}

}  // namespace glb
