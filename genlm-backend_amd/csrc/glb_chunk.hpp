// glb_chunk.hpp — the fused particle step as a chunked streaming reduction (gfx950 / wave64).
//
// Replaces, per particle (reference = genlm/genlm-backend):
//   cache.py:96      logps  = log_softmax(logits_row)
//   README.md:84-87  masked = logps + mask; logZ = logsumexp(masked); token = multinomial(exp(masked - logZ))
//   base.py:136-141  the same draw with logits scaled by 1/temperature
//
// Work is cut by the arithmetic contract, not by the launch geometry (glb_math.hpp): a row is a
// sequence of 4096-element chunks, each with its own binary scale, and every sum is an integer sum.  Three
// kernels:
//   chunk_stats_kernel   one WAVE per (row|particle, chunk): 16 KiB (fp32) / 8 KiB (16-bit) of the row in one
//                        burst of global_load_dwordx4, chunk maximum by DPP, one polynomial exp per element,
//                        both sums (all / allowed) by round-toward-zero fp32 adds on two grids, allowed lanes
//                        selected by EXEC from a pre-transposed bit mask read through the scalar cache; then, still
//                        from registers, the in-chunk stage of the Philox draw for the unit's first particles.  No
//                        barrier, no cross-wave dependency: the launch is pure streaming and the unit of
//                        scheduling is 16 KiB, so any row count / row length fills the chip.  Rows shared by
//                        several particles (dedup fan-out of hf.py:214-220,285-288) are reduced once.
//   finish_kernel        one wave per PARTICLE: folds the chunk records of its row into (N, S_all, S_mask), lse / logZ
//                        by a double-precision log, then the chunk stage of the draw (first Philox word against a
//                        scan of the chunk sums) and a look-up of the token the reducing wave drew in that chunk
//                        (second word); particles nobody drew for reload the chunk and reduce it again.  Also the
//                        parity-mode exponential race (four waves).
//   logprob_rows_kernel  x - lse for the API path that materialises log-probabilities (cache.py:93-98).
// mask_prepare_kernel builds the transposed masks ([mask][chunk][vector][component] 64-bit lane words).
#pragma once
#include <type_traits>

#include "glb_math.hpp"

namespace glb {

enum { kDtF32 = 0, kDtBf16 = 1, kDtF16 = 2 };
enum { kMaskNone = 0, kMaskBits = 1, kMaskF32 = 2 };
enum { kModeStats = 0, kModePhilox = 1, kModeNoise = 2 };

struct ChunkRec {  // 32 bytes per (row|particle, chunk)
  float Nc;            // chunk scale exp_n(max); -inf for an empty chunk
  uint32_t pA, pB;     // S_c   = (pA << 18) + pB     all elements
  uint32_t pAm, pBm;   // S_c^m = (pAm << 18) + pBm   allowed elements, on the scale Nm
  float Nm;            // bit masks: Nc, or the allowed maximum's own scale for a low-mass chunk; float masks:
                       // exp_n(max(x + mask)); no mask: Nc
  uint32_t pad[2];
};
static_assert(sizeof(ChunkRec) == 32, "ChunkRec layout");

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
  const float *noise;
  int64_t noise_ld;
  uint64_t seed, offset;
  int64_t particle_base;
  float *out_logZ, *out_lse;
  int32_t *out_token;
  float *out_margin;  // parity mode: relative gap between the two largest e_j / E_j of the race
  ChunkRec *recs;  // [n_pairs][nch]
  uint32_t *lanes;  // [n_pairs][nch][64][2] (Philox draws): per-lane inclusive scans of the allowed payload words of
                    // every chunk, as the reducing wave held them - what the per-particle launch picks the lane from
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

// the wave's 64 elements per lane of chunk c: vector i covers elements c*4096 + (i*64 + lane)*EPV ...
template <int DT, bool SCALED>
__device__ __forceinline__ void load_chunk(const char *rowp, int e_base, int V, int lane, float scale,
                                           float (&x)[64]) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  u32x4_t raw[NVC];
  if (e_base + kChunk <= V) {  // wave-uniform: a full chunk, every load in range
    const char *q = rowp + ((int64_t)e_base + lane * EPV) * ES;
#pragma unroll
    for (int i = 0; i < NVC; ++i) raw[i] = *reinterpret_cast<const u32x4_t *>(q + (int64_t)i * 64 * 16);
  } else {
#pragma unroll
    for (int i = 0; i < NVC; ++i) raw[i] = load_vec_guarded<DT>(rowp, e_base + (i * 64 + lane) * EPV, V);
  }
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    unpack_vec<DT>(raw[i], &x[i * EPV]);
    if constexpr (SCALED) {
#pragma unroll
      for (int k = 0; k < EPV; ++k) x[i * EPV + k] *= scale;
    }
  }
}

__device__ __forceinline__ void store_rec(ChunkRec *dst, float Nc, uint32_t pA, uint32_t pB, uint32_t pAm,
                                          uint32_t pBm, float Nm) {
  u32x4_t a{__float_as_uint(Nc), pA, pB, pAm}, b{pBm, __float_as_uint(Nm), 0u, 0u};
  u32x4_t *o = reinterpret_cast<u32x4_t *>(dst);
  o[0] = a;
  o[1] = b;
}

// sums of one chunk held in x[64]: returns the lane-63 totals of the four payload words.  (v0, VSTEP): this wave's
// vectors are v0, v0 + VSTEP, ... (one wave per chunk: 0, 1; four waves per chunk: wave, 4) and x holds them densely.
// the first two groups of mask words of a chunk (8 words each), fetched by the caller before it waits for the chunk's
// own loads and hands to chunk_sums
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

template <int DT, bool MASKED, int VSTEP, bool FULL>
__device__ __forceinline__ void chunk_sums_body(const float (&x)[64], float magicN, int nv_valid, cu64_t mt, int v0,
                                                const MaskAhead &ahead, float &A0, float &B0, float &A1, float &B1, float &Am0, float &Bm0,
                                                float &Am1, float &Bm1) {
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
      if (FULL || vi < nv_valid) {  // wave-uniform: vectors wholly past the row end are skipped
        float t[EPV];
#pragma unroll
        for (int k = 0; k < EPV; ++k) t[k] = chunk_term(x[i * EPV + k], magicN);
#pragma unroll
        for (int h = 0; h < EPV / 4; ++h) {
          const uint64_t *M = &Mq[g % 3][iv * EPV + 4 * h];
          rtz_acc4<MASKED>(t[4 * h], t[4 * h + 1], t[4 * h + 2], t[4 * h + 3], A0, B0, A1, B1, Am0, Bm0, Am1, Bm1,
                           M[0], M[1], M[2], M[3]);
        }
      }
    }
  }
}

template <int DT, bool MASKED, int VSTEP = 1>
__device__ __forceinline__ void chunk_sums(const float (&x)[64], float magicN, int nv_valid, cu64_t mt,
                                           const MaskAhead &ahead, uint32_t &pA, uint32_t &pB, uint32_t &pAm,
                                           uint32_t &pBm, int v0 = 0) {
  float A0 = __uint_as_float(kA0Bits), A1 = A0, Am0 = A0, Am1 = A0;
  float B0 = __uint_as_float(kB0Bits), B1 = B0, Bm0 = B0, Bm1 = B0;
  // a full chunk (all but the last of a row) runs as one straight block: the scheduler can then start the scalar loads
  // of the mask words well ahead of the adds that use them; the last chunk tests every vector against the row end
  if (nv_valid == ElemTraits<DT>::NVC)
    chunk_sums_body<DT, MASKED, VSTEP, true>(x, magicN, nv_valid, mt, v0, ahead, A0, B0, A1, B1, Am0, Bm0, Am1, Bm1);
  else
    chunk_sums_body<DT, MASKED, VSTEP, false>(x, magicN, nv_valid, mt, v0, ahead, A0, B0, A1, B1, Am0, Bm0, Am1, Bm1);
  pA = wave_sum_u32_l63((__float_as_uint(A0) - kA0Bits) + (__float_as_uint(A1) - kA0Bits));
  pB = wave_sum_u32_l63((__float_as_uint(B0) - kB0Bits) + (__float_as_uint(B1) - kB0Bits));
  pAm = pA;
  pBm = pB;
  if constexpr (MASKED) {
    pAm = wave_sum_u32_l63((__float_as_uint(Am0) - kA0Bits) + (__float_as_uint(Am1) - kA0Bits));
    pBm = wave_sum_u32_l63((__float_as_uint(Bm0) - kB0Bits) + (__float_as_uint(Bm1) - kB0Bits));
  }
}

__device__ __forceinline__ float chunk_max(const float (&x)[64]) {
  float m = kNegInf;
#pragma unroll
  for (int j = 0; j < 64; j += 2) m = max3(m, x[j], x[j + 1]);
  return wave_max(m);
}

// Bit-masked chunk: both sums at the chunk scale Nc (one exponential per element); if the allowed sum comes out
// below 2^32 although the mask allows something in this chunk (allows_any, from mask_prepare) - allowed mass under
// about 2^-3.5 of the chunk's largest term, e.g. the one likely token is the forbidden one - the forbidden elements of x are overwritten with -inf and the allowed ones summed
// again on their own maximum's scale (wave-uniform, rare, everything still in registers).  (pAm, pBm) are per-lane
// inclusive scans as chunk_sums returns them, on the scale Nm; `redone` tells the caller x is now the masked chunk.
template <int DT>
__device__ __forceinline__ void chunk_reduce_bits(float (&x)[64], float Nc, int nv_valid, cu64_t mt,
                                                  const MaskAhead &ma, uint64_t allows_any, int lane, uint32_t &pA,
                                                  uint32_t &pB, uint32_t &pAm, uint32_t &pBm, float &Nm,
                                                  bool &redone) {
  chunk_sums<DT, true>(x, kMagic - Nc, nv_valid, mt, ma, pA, pB, pAm, pBm);
  Nm = Nc;
  redone = false;
  const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)pAm, 63);
  const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)pBm, 63);
  const uint64_t Sm = ((uint64_t)ta << kGridHi) + tb;
  uint32_t top = (uint32_t)(Sm >> kLowMassBits), any = (uint32_t)allows_any;
  opaque_u32(top);
  opaque_u32(any);
  if (top == 0u && any != 0u) {
#pragma unroll
    for (int j = 0; j < 64; ++j)
      if (!((mt[j] >> lane) & 1ull)) x[j] = kNegInf;
    Nm = exp_n(chunk_max(x));
    uint32_t d0, d1;
    chunk_sums<DT, false>(x, kMagic - Nm, nv_valid, nullptr, MaskAhead{}, pAm, pBm, d0, d1);
    redone = true;
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
// the chunk's allowed terms on the scale they were summed on, taken lane by lane and inside a lane in register order
// (vector, component) - the order the reducing wave held them in.  The reducing wave leaves the per-lane inclusive
// scans of its allowed payload words behind (512 bytes per chunk); the per-particle launch picks the lane from them
// (stage 2a) and the element from one gathered load of that lane's 64 values (stage 2b).
// ---------------------------------------------------------------------------------------------------------
// stage 2a: the lane.  Packed result (lane << 56) | (target left inside that lane's terms); ~0 for an empty chunk.
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

// stage 2b, common end: every lane holds one of the chosen lane's 64 terms as (h, l); the element is the first whose
// running sum passes Tl.  Returns its index 0..63 in the lane's register order, -1 if none (consistent sums rule it out).
__device__ __forceinline__ int pick_in_lane(uint32_t h, uint32_t l, uint64_t Tl) {
  const uint32_t sh = wave_sum_u32_l63(h), sl = wave_sum_u32_l63(l);  // inclusive scans; totals < 2^24 each
  const uint64_t inc2 = ((uint64_t)sh << kGridHi) + sl;
  return first_lane_above(inc2, Tl);
}

// ---------------------------------------------------------------------------------------------------------
// chunk statistics: one wave per (reduction unit, chunk)
// ---------------------------------------------------------------------------------------------------------
// LANES (calls that draw with Philox): the wave also leaves the per-lane inclusive scans of its allowed payload words
// in the workspace (8 bytes per lane, one coalesced 512-byte store) - the per-particle launch draws from them.
template <int DT, int MASK, bool SCALED, bool LANES>
__global__ __launch_bounds__(256, 5) void chunk_stats_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  const int lane = threadIdx.x & 63;
  const int item = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int nch = p.nch;
  if (item >= p.n_pairs * nch) return;  // whole waves only
  const int pr = item / nch, c = item - pr * nch;  // (dealt row-interleaved instead, the launch is 2.5 us slower)
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int V = p.V, e_base = c * kChunk;
  int nv_valid = (V - e_base + 64 * EPV - 1) / (64 * EPV);
  nv_valid = nv_valid < NVC ? nv_valid : NVC;

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
  float x[64];
  load_chunk<DT, SCALED>(rowp, e_base, V, lane, p.scale, x);
  const float Nc = exp_n(chunk_max(x));
  uint32_t pA, pB, pAm, pBm;
  float Nm = Nc;
  if constexpr (MASK == kMaskBits) {
    bool redone;
    chunk_reduce_bits<DT>(x, Nc, nv_valid, mt, ma, allows_any, lane, pA, pB, pAm, pBm, Nm, redone);
  } else {
    chunk_sums<DT, false>(x, kMagic - Nc, nv_valid, nullptr, ma, pA, pB, pAm, pBm);
    if constexpr (MASK == kMaskF32) {  // general additive masks: y = x + m has its own maximum, scale and exp
      const char *mrow = (const char *)(p.mask_f + (int64_t)mi * p.mask_ld);
      float y[64];
      add_float_mask<DT>(x, mrow, e_base, V, lane, y);
      Nm = exp_n(chunk_max(y));
      uint32_t d0, d1;
      chunk_sums<DT, false>(y, kMagic - Nm, nv_valid, nullptr, ma, pAm, pBm, d0, d1);
    }
  }
  if constexpr (LANES) {  // (a chunk without allowed mass is never drawn from: sparse masks skip most of these stores)
    if ((__builtin_amdgcn_readlane((int)pAm, 63) | __builtin_amdgcn_readlane((int)pBm, 63)) != 0) {
      uint32_t *ls = p.lanes + (((int64_t)pr * nch + c) * 64 + lane) * 2;
      __builtin_nontemporal_store(((uint64_t)pBm << 32) | pAm, reinterpret_cast<uint64_t *>(ls));
    }
  }
  if (lane == 63) store_rec(p.recs + (int64_t)pr * nch + c, Nc, pA, pB, pAm, pBm, Nm);
}

// ---------------------------------------------------------------------------------------------------------
// chunk statistics for launches that cannot fill the chip (a single shared row at SIS step 0, a lone API query):
// one WORKGROUP of four waves per (unit, chunk); wave w takes vectors w, w+4, ... of the chunk, the chunk maximum and
// the four payload sums are combined through LDS.  Same records as chunk_stats_kernel (integer sums do not care how
// the elements were dealt out); a wave alone on its SIMD issues one instruction every ~5 cycles, so a quarter of the
// elements per wave is what shortens the launch.  Bit masks / no mask only (float masks use the one-wave kernel).
// ---------------------------------------------------------------------------------------------------------
template <int DT, int MASK, bool SCALED, bool LANES>
__global__ __launch_bounds__(256) void chunk_stats_small_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC, NVW = NVC / 4;
  static_assert(MASK != kMaskF32, "float masks take the one-wave-per-chunk kernel");
  __shared__ float s_max[4];
  __shared__ uint32_t s_pay[4][4];
  __shared__ uint32_t s_lane[LANES ? 4 : 1][64][2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x, nch = p.nch;
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
  uint32_t pA, pB, pAm, pBm;
  if constexpr (MASK == kMaskBits) {
    const int mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
    const cu64_t mt = as_const(p.mask_t + ((int64_t)mi * nch + c) * 64);
    MaskAhead ma;
    mask_ahead<DT, 4>(mt, ma, wave);
    chunk_sums<DT, true, 4>(x, kMagic - Nc, nv_valid, mt, ma, pA, pB, pAm, pBm, wave);
  } else {
    chunk_sums<DT, false, 4>(x, kMagic - Nc, nv_valid, nullptr, MaskAhead{}, pA, pB, pAm, pBm, wave);
  }
  if (lane == 63) {
    s_pay[wave][0] = pA;
    s_pay[wave][1] = pB;
    s_pay[wave][2] = pAm;
    s_pay[wave][3] = pBm;
  }
  __syncthreads();
  uint32_t t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = s_pay[0][k] + s_pay[1][k] + s_pay[2][k] + s_pay[3][k];
  float Nm = Nc;
  uint32_t lastA = pAm, lastB = pBm;  // this wave's per-lane scans of the allowed words, as finally summed
  (void)lastA;
  (void)lastB;
  if constexpr (MASK == kMaskBits) {
    // the low-mass rule of chunk_reduce_bits, with the chunk spread over four waves (workgroup-uniform branch)
    const uint64_t Sm = ((uint64_t)t[2] << kGridHi) + t[3];
    const int mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
    uint32_t top = (uint32_t)(Sm >> kLowMassBits), any = (uint32_t)as_const(p.mask_any)[(int64_t)mi * nch + c];
    opaque_u32(top);
    opaque_u32(any);
    if (top == 0u && any != 0u) {
      const cu64_t mt = as_const(p.mask_t + ((int64_t)mi * nch + c) * 64);
#pragma unroll
      for (int j = 0; j < NVW; ++j)
#pragma unroll
        for (int k = 0; k < EPV; ++k)
          if (!((mt[(wave + 4 * j) * EPV + k] >> lane) & 1ull)) x[j * EPV + k] = kNegInf;
      float mm = kNegInf;
#pragma unroll
      for (int j = 0; j < NVW * EPV; j += 2) mm = max3(mm, x[j], x[j + 1]);
      mm = wave_max(mm);
      __syncthreads();  // everybody has read s_max / s_pay
      if (lane == 0) s_max[wave] = mm;
      __syncthreads();
      Nm = exp_n(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])));
      uint32_t qa, qb, d0, d1;
      chunk_sums<DT, false, 4>(x, kMagic - Nm, nv_valid, nullptr, MaskAhead{}, qa, qb, d0, d1, wave);
      lastA = qa;
      lastB = qb;
      if (lane == 63) {
        s_pay[wave][2] = qa;
        s_pay[wave][3] = qb;
      }
      __syncthreads();
      t[2] = s_pay[0][2] + s_pay[1][2] + s_pay[2][2] + s_pay[3][2];
      t[3] = s_pay[0][3] + s_pay[1][3] + s_pay[2][3] + s_pay[3][3];
    }
  }
  if (threadIdx.x == 0) store_rec(p.recs + (int64_t)pr * nch + c, Nc, t[0], t[1], t[2], t[3], Nm);
  if constexpr (LANES) {
    // the per-lane scans of the allowed words over the WHOLE chunk: every wave's lanes hold scans over their quarter
    // of the vectors; their own parts are summed per lane through LDS and scanned again by wave 0
    const uint32_t ia = MASK == kMaskBits ? lastA : pA, ib = MASK == kMaskBits ? lastB : pB;
    const uint32_t pa = __shfl_up(ia, 1, 64), pb = __shfl_up(ib, 1, 64);
    s_lane[wave][lane][0] = ia - (lane ? pa : 0u);
    s_lane[wave][lane][1] = ib - (lane ? pb : 0u);
    __syncthreads();
    if (wave == 0) {
      const uint32_t a = s_lane[0][lane][0] + s_lane[1][lane][0] + s_lane[2][lane][0] + s_lane[3][lane][0];
      const uint32_t b2 = s_lane[0][lane][1] + s_lane[1][lane][1] + s_lane[2][lane][1] + s_lane[3][lane][1];
      const uint32_t sa = wave_sum_u32_l63(a), sb = wave_sum_u32_l63(b2);
      *reinterpret_cast<uint64_t *>(p.lanes + (((int64_t)pr * nch + c) * 64 + lane) * 2) = ((uint64_t)sb << 32) | sa;
    }
  }
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

// one element of the row (unscaled)
template <int DT>
__device__ __forceinline__ float load_elem(const char *rowp, int j) {
  if constexpr (DT == kDtF32) return *reinterpret_cast<const float *>(rowp + (int64_t)j * 4);
  const uint32_t h = *reinterpret_cast<const uint16_t *>(rowp + (int64_t)j * 2);
  if constexpr (DT == kDtBf16) return __uint_as_float(h << 16);
  return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}

// stage 2b: lane j loads the chosen lane's j-th element (eight or sixteen 16-byte runs a kilobyte apart), applies scale
// and mask exactly as the reduction did, and the scan of pick_in_lane names the element.  Nm: the scale the chunk's
// allowed terms were summed on (its record).
template <int DT, int MASK>
__device__ __forceinline__ int32_t chunk_pick_element_mem(const RowView<DT, MASK> &rv, int c, uint64_t pick, float Nm,
                                                          int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV;
  if (pick == ~0ull) return -1;
  const int lsel = (int)(pick >> 56);
  const uint64_t Tl = pick & ((1ull << 56) - 1ull);
  const int e = c * kChunk + ((lane / EPV) * 64 + lsel) * EPV + (lane % EPV);
  float y = kNegInf;
  if (e < rv.V) {
    y = load_elem<DT>(rv.rowp, e) * rv.scale;
    if constexpr (MASK == kMaskBits) {
      if (!((rv.mt[(int64_t)c * 64 + lane] >> lsel) & 1ull)) y = kNegInf;
    } else if constexpr (MASK == kMaskF32) {
      y = y + reinterpret_cast<const float *>(rv.mrow)[e];
    }
  }
  uint32_t h, l;
  term_q_parts(y, kMagic - Nm, h, l);
  const int jsel = pick_in_lane(h, l, Tl);
  if (jsel < 0) return -1;
  return __builtin_amdgcn_readlane(e, jsel);
}

// ---------------------------------------------------------------------------------------------------------
// What one particle needs from the chunk records of its (row, mask) pair.  Every function here is executed by ONE
// wave; all results are wave-uniform.
// ---------------------------------------------------------------------------------------------------------
struct RecsGlobal {
  const ChunkRec *r;
  __device__ __forceinline__ ChunkRec get(int c) const { return r[c]; }
};
struct PairState {
  float N_all, N_msk;     // row scales (all / allowed)
  uint64_t S_all, S_msk;  // sums on them
};

// fold the chunk records: row scales (the largest chunk scale among non-empty chunks), then the sums shifted onto them
template <int MASK, class Recs>
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

// lse / logZ (sum of e^x = 2^(N + 1 - 36) * S) - call from one lane
__device__ __forceinline__ void pair_logs(const PairState &st, float &lse, float &logZ) {
  const double lse_all = st.S_all ? log_fix(st.S_all, (int32_t)st.N_all + 1 - kFrac) : (double)kNegInf;
  const double lse_msk = st.S_msk ? log_fix(st.S_msk, (int32_t)st.N_msk + 1 - kFrac) : (double)kNegInf;
  lse = (float)lse_all;
  logZ = (float)(lse_msk - lse_all);
}

// The Philox draw of particle pidx, in two calls so that the caller can put other work between a load and its use:
// pair_pick_chunk - the chunk by a scan of the shifted chunk sums (first Philox word), and the request for this lane's
// scan word of that chunk (what the reducing wave left in the workspace); pair_pick_token - with the second word, the
// lane from those scans, then the element from a gathered load of that lane's 64 values.
struct ChunkPick {
  int csel;       // -1: nothing to draw from
  float Nms;      // the scale the chosen chunk's allowed terms sit on
  uint64_t R2;    // second Philox word
  uint64_t scan;  // this lane's (A, B) inclusive scan words of the chosen chunk
};

template <class Recs>
__device__ __forceinline__ ChunkPick pair_pick_chunk(const StepParams &p, const Recs &recs, const PairState &st, int pidx,
                                                     int nch, int lane, const uint32_t *lanes) {
  ChunkPick k{-1, 0.f, 0ull, 0ull};
  uint32_t nz = (uint32_t)st.S_msk | (uint32_t)(st.S_msk >> 32);
  opaque_u32(nz);
  if (nz == 0u) return k;
  uint64_t R1;
  philox_pair(p, pidx, R1, k.R2);
  uint64_t T = __umul64hi(R1, st.S_msk);  // uniform integer in [0, S_msk)
  for (int c0 = 0; c0 < nch && k.csel < 0; c0 += 64) {
    const int c = c0 + lane;
    uint64_t sm = 0;
    float Nm = kNegInf;
    if (c < nch) {
      const ChunkRec r = recs.get(c);
      const uint64_t m = ((uint64_t)r.pAm << kGridHi) + r.pBm;
      Nm = r.Nm;
      if (m) {
        const float d = st.N_msk - r.Nm;
        if (d < 64.0f) sm = m >> (uint32_t)d;
      }
    }
    const uint64_t incl = wave_scan_u64(sm);
    const int lsel = first_lane_above(incl, T);
    if (lsel >= 0) {
      k.csel = c0 + lsel;
      k.Nms = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(Nm), lsel));
    } else {
      T -= readlane_u64(incl, 63);
    }
  }
  if (k.csel >= 0) k.scan = *reinterpret_cast<const uint64_t *>(lanes + ((int64_t)k.csel * 64 + lane) * 2);
  return k;
}

template <int DT, int MASK>
__device__ __forceinline__ int32_t pair_pick_token(const RowView<DT, MASK> &rv, const ChunkPick &k, int lane) {
  if (k.csel < 0) return -1;  // (with a non-zero sum, consistent records rule this out)
  const uint64_t pick = chunk_pick_lane((uint32_t)k.scan, (uint32_t)(k.scan >> 32), k.R2);
  return chunk_pick_element_mem(rv, k.csel, pick, k.Nms, lane);
}

// ---------------------------------------------------------------------------------------------------------
// finish: one workgroup per particle - a single wave for the statistics / Philox modes (launched with 64 threads),
// four waves for parity mode, which deals the row's vectors to all of them.
// ---------------------------------------------------------------------------------------------------------
template <int DT, int MASK, int MODE>
__global__ __launch_bounds__(256) void finish_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  __shared__ float s_bestg[4], s_secg[4];
  __shared__ int32_t s_bestj[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (MODE != kModeNoise && wave != 0) return;
  const int pidx = blockIdx.x;
  const int nch = p.nch, V = p.V;
  const int pr = p.pair_of ? as_const(p.pair_of)[pidx] : pidx;
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const int mi = MASK == kMaskNone ? 0 : (p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr));
  const RecsGlobal recs{p.recs + (int64_t)pr * nch};

  RowView<DT, MASK> rv;
  rv.rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  rv.V = V;
  rv.scale = p.scale;
  rv.mt = MASK == kMaskBits ? p.mask_t + (int64_t)mi * nch * 64 : nullptr;
  rv.many = MASK == kMaskBits ? p.mask_any + (int64_t)mi * nch : nullptr;
  rv.mrow = MASK == kMaskF32 ? (const char *)(p.mask_f + (int64_t)mi * p.mask_ld) : nullptr;

  PairState st;
  pair_fold<MASK>(recs, nch, lane, st);
  ChunkPick pick{-1, 0.f, 0ull, 0ull};
  if constexpr (MODE == kModePhilox) {  // the chunk, and the request for its scan words: in flight during the logs
    if (p.out_token) pick = pair_pick_chunk(p, recs, st, pidx, nch, lane, p.lanes + (int64_t)pr * nch * 128);
  }
  if (wave == 0 && lane == 0) {
    float lse, logZ;
    pair_logs(st, lse, logZ);
    if (p.out_lse) p.out_lse[pidx] = lse;
    if (p.out_logZ) p.out_logZ[pidx] = logZ;
  }
  if constexpr (MODE == kModeStats) return;
  if (!p.out_token) return;

  if constexpr (MODE == kModePhilox) {
    const int32_t tok = pair_pick_token<DT, MASK>(rv, pick, lane);
    if (lane == 0) p.out_token[pidx] = tok;
  } else {
    // ---- parity mode: exponential race against the caller's noise, first maximum of e_j / E_j (README.md:87
    //      through torch.multinomial's CPU algorithm).  e_j = ldexp(P, n_j - N_msk): any common scale gives the same
    //      comparisons; N_msk is the largest scale any chunk's allowed terms were summed on, so every allowed term that
    //      can win is representable.  Vectors are dealt round
    //      robin to the four waves; ties resolve to the smallest index at every level ------------------------------
    int32_t tok = -1;
    if (st.S_msk != 0) {  // workgroup-uniform: every wave folded the same records
      const float magicN = kMagic - st.N_msk;
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
            const float e = chunk_term(y[jv][k], magicN);
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
// log-probability rows: out[r, j] = x'[r, j] - lse[r]  (one thread per 16-byte input vector)
// ---------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void logprob_rows_kernel(const void *logits, int64_t ld, int V, float scale,
                                                          const float *lse, float *out, int64_t out_ld,
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
  float *o = out + (int64_t)r * out_ld + e0;
  if (e0 + EPV <= V) {
#pragma unroll
    for (int h = 0; h < EPV / 4; ++h) {
      float4 v;
      v.x = x[4 * h] * scale - l;
      v.y = x[4 * h + 1] * scale - l;
      v.z = x[4 * h + 2] * scale - l;
      v.w = x[4 * h + 3] * scale - l;
      *reinterpret_cast<float4 *>(o + 4 * h) = v;
    }
  } else {
    for (int k = 0; k < V - e0; ++k) o[k] = x[k] * scale - l;
  }
}

// ---------------------------------------------------------------------------------------------------------
// log-probability rows in one launch: one 1024-thread workgroup per row.  Wave w reduces chunks w, w+16, ... of the
// row (same per-chunk arithmetic as chunk_stats_kernel, records kept in LDS), one wave folds them into lse, then every
// wave writes x - lse for its chunks: straight from its registers when the row has at most 16 chunks (ONE: a wave
// owns one chunk - gpt2-sized rows), else after streaming them again (read microseconds ago: L2 / Infinity Cache serve
// most of it).  HBM traffic ~ V*s + 4V per row instead of 2*V*s + 4V of the three-launch path; used when there are
// enough rows to fill the chip (the three-launch path spreads a few rows over the chip chunk by chunk).
// ---------------------------------------------------------------------------------------------------------
template <int DT, bool SCALED, bool ONE>
__global__ __launch_bounds__(1024) void logprob_rows_fused_kernel(const void *logits, int64_t ld, int V, int nch,
                                                                 float scale, float *out, int64_t out_ld,
                                                                 float *out_lse) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  extern __shared__ uint64_t s_rec[];  // [nch] chunk sums, then [nch] floats of chunk scales
  float *s_N = reinterpret_cast<float *>(s_rec + nch);
  __shared__ float s_lse;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = blockIdx.x;
  const char *rowp = (const char *)logits + (int64_t)r * ld * ES;
  float x[64];
  for (int c = wave; c < nch; c += 16) {
    const int e_base = c * kChunk;
    int nv_valid = (V - e_base + 64 * EPV - 1) / (64 * EPV);
    nv_valid = nv_valid < NVC ? nv_valid : NVC;
    load_chunk<DT, SCALED>(rowp, e_base, V, lane, scale, x);
    const float Nc = exp_n(chunk_max(x));
    uint32_t pA, pB, pAm, pBm;
    chunk_sums<DT, false>(x, kMagic - Nc, nv_valid, nullptr, MaskAhead{}, pA, pB, pAm, pBm);
    if (lane == 63) {
      s_rec[c] = ((uint64_t)pA << kGridHi) + pB;
      s_N[c] = Nc;
    }
  }
  __syncthreads();
  if (wave == 0) {
    float N = kNegInf;
    for (int c = lane; c < nch; c += 64)
      if (s_rec[c]) N = fmaxf(N, s_N[c]);
    N = wave_max(N);
    uint64_t S = 0;
    for (int c0 = 0; c0 < nch; c0 += 64) {
      const int c = c0 + lane;
      uint64_t sa = 0;
      if (c < nch && s_rec[c]) {
        const float d = N - s_N[c];
        sa = d < 64.0f ? s_rec[c] >> (uint32_t)d : 0ull;
      }
      S += wave_sum_u64(sa);
    }
    if (lane == 0) {
      const float lse = S ? (float)log_fix(S, (int32_t)N + 1 - kFrac) : kNegInf;
      s_lse = lse;
      if (out_lse) out_lse[r] = lse;
    }
  }
  __syncthreads();
  if (!out) return;
  const float l = s_lse;
  float *orow = out + (int64_t)r * out_ld;
  for (int c = wave; c < nch; c += 16) {
    const int e_base = c * kChunk;
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int e0 = e_base + (i * 64 + lane) * EPV;
      if (e0 >= V) continue;
      float y[EPV];
      if constexpr (ONE) {  // the wave's only chunk is still in its registers (already scaled)
#pragma unroll
        for (int k = 0; k < EPV; ++k) y[k] = x[i * EPV + k];
      } else {
        unpack_vec<DT>(load_vec_guarded<DT>(rowp, e0, V), y);
        if constexpr (SCALED) {
#pragma unroll
          for (int k = 0; k < EPV; ++k) y[k] *= scale;
        }
      }
      if (e0 + EPV <= V) {
#pragma unroll
        for (int h = 0; h < EPV / 4; ++h) {
          float4 v;
          v.x = y[4 * h] - l;
          v.y = y[4 * h + 1] - l;
          v.z = y[4 * h + 2] - l;
          v.w = y[4 * h + 3] - l;
          *reinterpret_cast<float4 *>(orow + e0 + 4 * h) = v;
        }
      } else {
        for (int k = 0; k < V - e0; ++k) orow[e0 + k] = y[k] - l;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// mask preparation: bit rows [n_masks, mask_ld] -> transposed lane words.  Block (c, k) transposes chunk c of mask k.
// ---------------------------------------------------------------------------------------------------------
template <int EPV>
__global__ __launch_bounds__(256) void mask_prepare_kernel(const uint32_t *bits, int64_t mask_ld, int V, int nch,
                                                          uint64_t *mask_t, uint64_t *mask_any) {
  __shared__ uint64_t s_any[4];
  const int k = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t *row = bits + (int64_t)k * mask_ld;
  uint64_t *dst = mask_t + ((int64_t)k * nch + c) * 64;
  bool on[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {  // word w = wave + 4j = vector (w / EPV), component (w % EPV); loads first
    const int w = wave + 4 * j, i = w / EPV, kk = w % EPV;
    const int e = c * kChunk + (i * 64 + lane) * EPV + kk;
    on[j] = e < V && ((row[e < V ? e >> 5 : 0] >> (e & 31)) & 1u);
  }
  uint64_t any = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint64_t m = __ballot(on[j]);
    any |= m;
    if (lane == 0) dst[wave + 4 * j] = m;
  }
  if (lane == 0) s_any[wave] = any;
  __syncthreads();
  if (tid == 0) mask_any[(int64_t)k * nch + c] = (s_any[0] | s_any[1] | s_any[2] | s_any[3]) ? 1ull : 0ull;
}

}  // namespace glb
