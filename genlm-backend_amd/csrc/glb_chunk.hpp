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
//                        selected by EXEC from a pre-transposed bit mask read through the scalar cache.  No
//                        LDS, no barrier, no cross-wave dependency: the launch is pure streaming and the unit
//                        of scheduling is 16 KiB, so any row count / row length fills the chip.  Rows shared by
//                        several particles (dedup fan-out of hf.py:214-220,285-288) are reduced once.
//   finish_kernel        one wave per PARTICLE: folds the <= 64 chunk records of its row into (N, S_all, S_mask),
//                        lse / logZ by a double-precision log, then the draw: Philox target -> chunk (scan of the
//                        chunk sums) -> vector -> lane -> element, recomputing only the one chunk it lands in.
//                        Also the rare own-scale redo (allowed mass below 2^-4 of the row's largest term) and the
//                        parity-mode exponential race.
//   logprob_rows_kernel  x - lse for the API path that materialises log-probabilities (cache.py:93-98).
// mask_prepare_kernel builds the transposed masks ([mask][chunk][vector][component] 64-bit lane words) and a
// short list of allowed ids for sparse masks (the README's EOS-only mask, README.md:62-66).
#pragma once
#include <type_traits>

#include "glb_math.hpp"

namespace glb {

enum { kDtF32 = 0, kDtBf16 = 1, kDtF16 = 2 };
enum { kMaskNone = 0, kMaskBits = 1, kMaskF32 = 2 };
enum { kModeStats = 0, kModePhilox = 1, kModeNoise = 2 };

constexpr int kInfoWords = 64;  // per mask: allowed-token count, then up to 63 ids (when count <= 63)

struct ChunkRec {  // 32 bytes per (row|particle, chunk)
  float Nc;            // chunk scale exp_n(max); -inf for an empty chunk
  uint32_t pA, pB;     // S_c   = (pA << 18) + pB     all elements
  uint32_t pAm, pBm;   // S_c^m = (pAm << 18) + pBm   allowed elements (bit masks: on Nc; float masks: on Nm)
  float Nm;            // float masks: scale of the masked chunk exp_n(max(x + mask))
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
  const int32_t *mask_info;  // [n_masks][kInfoWords]
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
template <int DT, bool MASKED, int VSTEP = 1>
__device__ __forceinline__ void chunk_sums(const float (&x)[64], float magicN, int nv_valid, cu64_t mt,
                                           uint32_t &pA, uint32_t &pB, uint32_t &pAm, uint32_t &pBm, int v0 = 0) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC / VSTEP;
  float A0 = __uint_as_float(kA0Bits), A1 = A0, Am0 = A0, Am1 = A0;
  float B0 = __uint_as_float(kB0Bits), B1 = B0, Bm0 = B0, Bm1 = B0;
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    const int vi = v0 + i * VSTEP;  // the vector's index inside the chunk
    if (vi < nv_valid) {  // wave-uniform: vectors wholly past the row end are skipped
      float t[EPV];
#pragma unroll
      for (int k = 0; k < EPV; ++k) t[k] = chunk_term(x[i * EPV + k], magicN);
#pragma unroll
      for (int h = 0; h < EPV / 4; ++h) {
        uint64_t M0 = 0, M1 = 0, M2 = 0, M3 = 0;
        if constexpr (MASKED) {
          M0 = mt[vi * EPV + 4 * h + 0];
          M1 = mt[vi * EPV + 4 * h + 1];
          M2 = mt[vi * EPV + 4 * h + 2];
          M3 = mt[vi * EPV + 4 * h + 3];
        }
        rtz_acc4<MASKED>(t[4 * h], t[4 * h + 1], t[4 * h + 2], t[4 * h + 3], A0, B0, A1, B1, Am0, Bm0, Am1, Bm1,
                         M0, M1, M2, M3);
      }
    }
  }
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

// ---------------------------------------------------------------------------------------------------------
// chunk statistics: one wave per (reduction unit, chunk)
// ---------------------------------------------------------------------------------------------------------
#ifndef GLB_K1_MINW
#define GLB_K1_MINW 1
#endif
template <int DT, int MASK, bool SCALED>
__global__ __launch_bounds__(256, GLB_K1_MINW) void chunk_stats_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  const int lane = threadIdx.x & 63;
  const int item = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int nch = p.nch;
  if (item >= p.n_pairs * nch) return;  // whole waves only
  const int pr = item / nch, c = item - pr * nch;
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int V = p.V, e_base = c * kChunk;
  int nv_valid = (V - e_base + 64 * EPV - 1) / (64 * EPV);
  nv_valid = nv_valid < NVC ? nv_valid : NVC;

  float x[64];
  load_chunk<DT, SCALED>(rowp, e_base, V, lane, p.scale, x);
  const float Nc = exp_n(chunk_max(x));
  uint32_t pA, pB, pAm, pBm;
  float Nm = Nc;
  if constexpr (MASK == kMaskBits) {
    const int mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
    const cu64_t mt = as_const(p.mask_t + ((int64_t)mi * nch + c) * 64);
    chunk_sums<DT, true>(x, kMagic - Nc, nv_valid, mt, pA, pB, pAm, pBm);
  } else {
    chunk_sums<DT, false>(x, kMagic - Nc, nv_valid, nullptr, pA, pB, pAm, pBm);
    if constexpr (MASK == kMaskF32) {  // general additive masks: y = x + m has its own maximum, scale and exp
      const int mi = p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr);
      const char *mrow = (const char *)(p.mask_f + (int64_t)mi * p.mask_ld);
      float y[64];
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
      Nm = exp_n(chunk_max(y));
      uint32_t d0, d1;
      chunk_sums<DT, false>(y, kMagic - Nm, nv_valid, nullptr, pAm, pBm, d0, d1);
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
template <int DT, int MASK, bool SCALED>
__global__ __launch_bounds__(256) void chunk_stats_small_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC, NVW = NVC / 4;
  static_assert(MASK != kMaskF32, "float masks take the one-wave-per-chunk kernel");
  __shared__ float s_max[4];
  __shared__ uint32_t s_pay[4][4];
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
    chunk_sums<DT, true, 4>(x, kMagic - Nc, nv_valid, mt, pA, pB, pAm, pBm, wave);
  } else {
    chunk_sums<DT, false, 4>(x, kMagic - Nc, nv_valid, nullptr, pA, pB, pAm, pBm, wave);
  }
  if (lane == 63) {
    s_pay[wave][0] = pA;
    s_pay[wave][1] = pB;
    s_pay[wave][2] = pAm;
    s_pay[wave][3] = pBm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = s_pay[0][k] + s_pay[1][k] + s_pay[2][k] + s_pay[3][k];
    store_rec(p.recs + (int64_t)pr * nch + c, Nc, t[0], t[1], t[2], t[3], Nc);
  }
}

// ---------------------------------------------------------------------------------------------------------
// finish: one wave per particle
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t philox_bits(const StepParams &p, int pidx) {
  const uint64_t gp = (uint64_t)(p.particle_base + pidx);
  const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
  const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
  uint32_t rnd[4];
  philox4x32_10(ctr, key, rnd);
  return ((uint64_t)rnd[1] << 32) | rnd[0];
}

// first lane whose inclusive scan exceeds T (wave-uniform T); -1 if none
__device__ __forceinline__ int first_lane_above(uint64_t incl, uint64_t T) {
  return __ffsll((long long)__ballot(incl > T)) - 1;
}

// the row (or float-mask) view one particle works on
template <int DT, int MASK>
struct RowView {
  const char *rowp;
  int V;
  float scale;
  const uint64_t *mt;   // transposed mask row (kMaskBits)
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

// Walk vectors [i_lo, i_hi) of chunk c in vocabulary order at the fixed scale magicN until the running sum of q
// passes T.  Returns the token (wave-uniform) or -1; T is updated (minus everything walked past).
template <int DT, int MASK>
__device__ __forceinline__ int32_t walk_vectors(const RowView<DT, MASK> &rv, int c, int i_lo, int i_hi, int lane,
                                                float magicN, uint64_t &T) {
  constexpr int EPV = ElemTraits<DT>::EPV;
  for (int i = i_lo; i < i_hi; ++i) {
    if (c * kChunk + i * 64 * EPV >= rv.V) return -1;
    float y[EPV];
    int e0;
    rv.vec(c, i, lane, y, e0);
    uint64_t q[EPV], s = 0;
#pragma unroll
    for (int k = 0; k < EPV; ++k) {
      q[k] = term_q(y[k], magicN);
      s += q[k];
    }
    const uint64_t incl = wave_scan_u64(s);
    const int lsel = first_lane_above(incl, T);
    if (lsel >= 0) {
      uint64_t Tl = T - (incl - s);
      int32_t tok = -1;
#pragma unroll
      for (int k = 0; k < EPV; ++k) {
        if (tok < 0) {
          if (Tl < q[k]) tok = e0 + k;
          else Tl -= q[k];
        }
      }
      return __builtin_amdgcn_readlane(tok, lsel);
    }
    T -= readlane_u64(incl, 63);
  }
  return -1;
}

// The draw inside one chunk, by the four waves of the particle's workgroup: wave w takes vectors w, w+4, ... of the
// chunk (one burst of loads), sums their allowed terms with the round-toward-zero adds and parks the per-vector
// totals in LDS; every wave then finds the vector holding the target by a scan over the <= 16 totals, and the wave
// that loaded that vector walks it element by element straight from its registers.  Every wave must call this (it
// contains a workgroup barrier); the owning wave returns the token, the others -1.
template <int DT, int MASK>
__device__ __forceinline__ int32_t draw_in_chunk(const RowView<DT, MASK> &rv, int c, int lane, int wave,
                                                 float magicN, uint64_t T, uint64_t *s_tot) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC, NVW = NVC / 4;
  const int e_base = c * kChunk;
  const cu64_t mt = MASK == kMaskBits ? as_const(rv.mt + (int64_t)c * 64) : nullptr;
  u32x4_t raw[NVW];
#pragma unroll
  for (int j = 0; j < NVW; ++j) {
    const int e0 = e_base + ((wave + 4 * j) * 64 + lane) * EPV;
    raw[j] = load_vec_guarded<DT>(rv.rowp, e0 < rv.V ? e0 : rv.V, rv.V);
  }
  float y[NVW][EPV];  // this wave's (masked-by-value for float masks) logits, kept for the walk
#pragma unroll
  for (int j = 0; j < NVW; ++j) {
    const int i = wave + 4 * j;
    unpack_vec<DT>(raw[j], y[j]);
#pragma unroll
    for (int k = 0; k < EPV; ++k) y[j][k] *= rv.scale;
    if constexpr (MASK == kMaskF32) {
#pragma unroll
      for (int h = 0; h < EPV / 4; ++h) {
        const int e0 = e_base + (i * 64 + lane) * EPV + 4 * h;
        const u32x4_t r = load_vec_guarded<kDtF32>(rv.mrow, e0 < rv.V ? e0 : rv.V, rv.V);
        float mk[4];
        unpack_vec<kDtF32>(r, mk);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[j][4 * h + k] += mk[k];
      }
    }
    float t[EPV];
#pragma unroll
    for (int k = 0; k < EPV; ++k) t[k] = chunk_term(y[j][k], magicN);
    float Am = __uint_as_float(kA0Bits), Bm = __uint_as_float(kB0Bits);
#pragma unroll
    for (int h = 0; h < EPV / 4; ++h) {
      uint64_t M0 = 0, M1 = 0, M2 = 0, M3 = 0;
      if constexpr (MASK == kMaskBits) {
        M0 = mt[i * EPV + 4 * h + 0];
        M1 = mt[i * EPV + 4 * h + 1];
        M2 = mt[i * EPV + 4 * h + 2];
        M3 = mt[i * EPV + 4 * h + 3];
      }
      rtz_vec4<MASK == kMaskBits>(t[4 * h], t[4 * h + 1], t[4 * h + 2], t[4 * h + 3], Am, Bm, M0, M1, M2, M3);
    }
    const uint32_t pa = wave_sum_u32_l63(__float_as_uint(Am) - kA0Bits);
    const uint32_t pb = wave_sum_u32_l63(__float_as_uint(Bm) - kB0Bits);
    if (lane == 63) s_tot[i] = ((uint64_t)pa << kGridHi) + pb;
  }
  __syncthreads();
  const uint64_t wv = lane < NVC ? s_tot[lane] : 0ull;
  const uint64_t incl = wave_scan_u64(wv);
  const int isel = first_lane_above(incl, T);
  if (isel < 0 || (isel & 3) != wave) return -1;
  T -= readlane_u64(incl - wv, isel);
  // walk vector isel = wave + 4 * jsel from the registers: lane -> element, in vocabulary order
  const int jsel = isel >> 2;
  uint64_t q[EPV], s = 0;
#pragma unroll
  for (int k = 0; k < EPV; ++k) {
    float v = y[0][k];
#pragma unroll
    for (int j = 1; j < NVW; ++j) v = jsel == j ? y[j][k] : v;
    const int e = e_base + (isel * 64 + lane) * EPV + k;
    bool ok = e < rv.V;
    if constexpr (MASK == kMaskBits) ok = ok && ((mt[isel * EPV + k] >> lane) & 1ull);
    q[k] = ok ? term_q(v, magicN) : 0ull;
    s += q[k];
  }
  const uint64_t inc2 = wave_scan_u64(s);
  const int lsel = first_lane_above(inc2, T);
  if (lsel < 0) return -1;
  uint64_t Tl = T - (inc2 - s);
  int32_t tok = -1;
#pragma unroll
  for (int k = 0; k < EPV; ++k) {
    if (tok < 0) {
      if (Tl < q[k]) tok = e_base + (isel * 64 + lane) * EPV + k;
      else Tl -= q[k];
    }
  }
  return __builtin_amdgcn_readlane(tok, lsel);
}

// masked maximum / sum of the whole row at a given scale (own-scale redo; rare)
template <int DT, int MASK>
__device__ __forceinline__ float row_masked_max(const RowView<DT, MASK> &rv, int nch, int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  float m = kNegInf;
  for (int c = 0; c < nch; ++c)
    for (int i = 0; i < NVC; ++i) {
      if (c * kChunk + i * 64 * EPV >= rv.V) break;
      float y[EPV];
      int e0;
      rv.vec(c, i, lane, y, e0);
#pragma unroll
      for (int k = 0; k < EPV; ++k) m = fmaxf(m, y[k]);
    }
  return wave_max(m);
}
template <int DT, int MASK>
__device__ __forceinline__ uint64_t row_masked_sum(const RowView<DT, MASK> &rv, int nch, int lane, float magicN) {
  constexpr int EPV = ElemTraits<DT>::EPV, NVC = ElemTraits<DT>::NVC;
  uint64_t s = 0;
  for (int c = 0; c < nch; ++c)
    for (int i = 0; i < NVC; ++i) {
      if (c * kChunk + i * 64 * EPV >= rv.V) break;
      float y[EPV];
      int e0;
      rv.vec(c, i, lane, y, e0);
#pragma unroll
      for (int k = 0; k < EPV; ++k) s += term_q(y[k], magicN);
    }
  return wave_sum_u64(s);
}

// one element of the row, scaled (sparse-mask path)
template <int DT>
__device__ __forceinline__ float load_elem(const char *rowp, int j) {
  if constexpr (DT == kDtF32) return *reinterpret_cast<const float *>(rowp + (int64_t)j * 4);
  const uint32_t h = *reinterpret_cast<const uint16_t *>(rowp + (int64_t)j * 2);
  if constexpr (DT == kDtBf16) return __uint_as_float(h << 16);
  return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}

template <int DT, int MASK, int MODE>
__global__ __launch_bounds__(256) void finish_kernel(const StepParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV, ES = ElemTraits<DT>::ES, NVC = ElemTraits<DT>::NVC;
  __shared__ uint64_t s_tot[16];
  __shared__ float s_bestg[4], s_secg[4];
  __shared__ int32_t s_bestj[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pidx = blockIdx.x;  // one workgroup of four waves per particle
  const int nch = p.nch, V = p.V;
  const int pr = p.pair_of ? as_const(p.pair_of)[pidx] : pidx;
  const int row = p.pair_row ? as_const(p.pair_row)[pr] : pr;
  const int mi = MASK == kMaskNone ? 0 : (p.pair_mask ? as_const(p.pair_mask)[pr] : (p.n_masks == 1 ? 0 : pr));
  const ChunkRec *recs = p.recs + (int64_t)pr * nch;

  RowView<DT, MASK> rv;
  rv.rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  rv.V = V;
  rv.scale = p.scale;
  rv.mt = MASK == kMaskBits ? p.mask_t + (int64_t)mi * nch * 64 : nullptr;
  rv.mrow = MASK == kMaskF32 ? (const char *)(p.mask_f + (int64_t)mi * p.mask_ld) : nullptr;

  // the particle's 64 random bits do not depend on the records: computed while the first loads are in flight
  uint64_t R = 0;
  if constexpr (MODE == kModePhilox) R = philox_bits(p, pidx);

  // ---- fold the chunk records (every wave, same values): row scales, then the sums shifted onto them ------------
  float N_all = kNegInf, N_msk = kNegInf;
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const int c = c0 + lane;
    if (c < nch) {
      const ChunkRec r = recs[c];
      if (r.pA | r.pB) N_all = fmaxf(N_all, r.Nc);
      if (r.pAm | r.pBm) N_msk = fmaxf(N_msk, MASK == kMaskF32 ? r.Nm : r.Nc);
    }
  }
  N_all = wave_max(N_all);
  N_msk = MASK == kMaskF32 ? wave_max(N_msk) : N_all;  // bit masks: allowed terms sit on the row's scale
  uint64_t S_all = 0, S_msk = 0;
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const int c = c0 + lane;
    uint64_t sa = 0, sm = 0;
    if (c < nch) {
      const ChunkRec r = recs[c];
      const uint64_t a = ((uint64_t)r.pA << kGridHi) + r.pB, m = ((uint64_t)r.pAm << kGridHi) + r.pBm;
      if (a) {
        const float d = N_all - r.Nc;  // >= 0, integer valued
        sa = d < 64.0f ? a >> (uint32_t)d : 0ull;
      }
      if (m) {
        const float d = N_msk - (MASK == kMaskF32 ? r.Nm : r.Nc);
        sm = d < 64.0f ? m >> (uint32_t)d : 0ull;
      }
    }
    S_all += wave_sum_u64(sa);
    if constexpr (MASK != kMaskNone) S_msk += wave_sum_u64(sm);
  }
  if constexpr (MASK == kMaskNone) S_msk = S_all;

  // ---- bit masks: allowed mass below 2^-4 of the row's largest term -> masked sum on its own scale (every wave
  //      redundantly: rare, and the branch must be taken by all four for the barriers further down) -----------------
  bool own = false;       // workgroup-uniform
  int sparse_n = -1;      // >= 0: the mask allows that many (<= 63) tokens, listed in mask_info
  float x_sp = kNegInf;   // sparse path: this lane's allowed logit
  int j_sp = -1;
  if constexpr (MASK == kMaskBits) {
    uint32_t top = (uint32_t)(S_msk >> kLowMassBits);
    opaque_u32(top);
    if (top == 0u) {
      own = true;
      const int32_t *info = p.mask_info + (int64_t)mi * kInfoWords;
      const int cnt = info[0];
      float mk;
      if (cnt <= kInfoWords - 1) {
        sparse_n = cnt;
        if (lane < cnt) {
          j_sp = info[1 + lane];
          x_sp = load_elem<DT>(rv.rowp, j_sp) * p.scale;
        }
        mk = wave_max(x_sp);
      } else {
        mk = row_masked_max(rv, nch, lane);
      }
      N_msk = exp_n(mk);
      if (!(mk > kNegInf)) {
        S_msk = 0;
      } else if (sparse_n >= 0) {
        S_msk = wave_sum_u64(lane < sparse_n ? term_q(x_sp, kMagic - N_msk) : 0ull);
      } else {
        S_msk = row_masked_sum(rv, nch, lane, kMagic - N_msk);
      }
    }
  }

  // ---- lse / logZ (sum of e^x = 2^(N + 1 - 36) * S): one lane of the last wave, beside the draw ---------------------
  if (wave == 3 && lane == 0) {
    const double lse_all = S_all ? log_fix(S_all, (int32_t)N_all + 1 - kFrac) : (double)kNegInf;
    const double lse_msk = S_msk ? log_fix(S_msk, (int32_t)N_msk + 1 - kFrac) : (double)kNegInf;
    if (p.out_lse) p.out_lse[pidx] = (float)lse_all;
    if (p.out_logZ) p.out_logZ[pidx] = (float)(lse_msk - lse_all);
  }
  if constexpr (MODE == kModeStats) return;
  if (!p.out_token) return;

  int32_t tok = -1;
  bool writer = wave == 0;  // who stores the token: wave 0, except after a chunk draw (the wave owning the vector)
  if (S_msk != 0) {  // workgroup-uniform from here on: every wave holds the same S_msk / own / N_msk
    if constexpr (MODE == kModePhilox) {
      uint64_t T = __umul64hi(R, S_msk);  // uniform integer in [0, S_msk)
      if (own) {
        if (wave == 0) {
          if (sparse_n >= 0) {
            const uint64_t q = lane < sparse_n ? term_q(x_sp, kMagic - N_msk) : 0ull;
            const uint64_t incl = wave_scan_u64(q);
            const int lsel = first_lane_above(incl, T);
            tok = __builtin_amdgcn_readlane(j_sp, lsel < 0 ? 0 : lsel);
          } else {
            for (int c = 0; c < nch && tok < 0; ++c) tok = walk_vectors(rv, c, 0, NVC, lane, kMagic - N_msk, T);
          }
        }
      } else {
        // chunk: scan of the shifted chunk sums in vocabulary order; then the target is carried onto the chunk's
        // own scale (T << shift stays below the chunk's unshifted sum) and the chunk is searched
        int csel = -1;
        float Ncs = 0.f;
        for (int c0 = 0; c0 < nch && csel < 0; c0 += 64) {
          const int c = c0 + lane;
          uint64_t sm = 0;
          float Nc = kNegInf;
          uint32_t sh = 0;
          if (c < nch) {
            const ChunkRec r = recs[c];
            const uint64_t m = ((uint64_t)r.pAm << kGridHi) + r.pBm;
            Nc = MASK == kMaskF32 ? r.Nm : r.Nc;
            if (m) {
              const float d = N_msk - Nc;
              if (d < 64.0f) {
                sh = (uint32_t)d;
                sm = m >> sh;
              }
            }
          }
          const uint64_t incl = wave_scan_u64(sm);
          const int lsel = first_lane_above(incl, T);
          if (lsel >= 0) {
            csel = c0 + lsel;
            const uint64_t before = readlane_u64(incl - sm, lsel);
            const uint32_t shs = (uint32_t)__builtin_amdgcn_readlane((int)sh, lsel);
            Ncs = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(Nc), lsel));
            T = (T - before) << shs;
          } else {
            T -= readlane_u64(incl, 63);
          }
        }
        if (csel >= 0) {
          // the wave that loaded the target's vector ends up with the token and stores it; -1 is put down first (it
          // stays only if nobody finds the target, which consistent sums rule out) - the barrier inside the draw
          // orders the two stores
          if (wave == 0 && lane == 0) p.out_token[pidx] = -1;
          tok = draw_in_chunk(rv, csel, lane, wave, kMagic - Ncs, T, s_tot);
          writer = tok >= 0;
        }
      }
    } else {
      // ---- parity mode: exponential race against the caller's noise, first maximum of e_j / E_j (README.md:87
      //      through torch.multinomial's CPU algorithm).  e_j = ldexp(P, n_j - N_msk): any common scale gives the same
      //      comparisons; the row scale keeps every allowed term that can win (the largest allowed term is within
      //      2^-4 of it unless `own`, and then N_msk is the masked maximum's own exponent).  Vectors are dealt round
      //      robin to the four waves; ties resolve to the smallest index at every level ------------------------------
      const float magicN = kMagic - N_msk;
      const float *E = p.noise + (int64_t)pidx * p.noise_ld;
      float best = -1.0f, sec = -1.0f;  // sec: runner-up of the race (for the reported margin)
      int32_t bj = -1;
      for (int c = 0; c < nch; ++c)
        for (int i = wave; i < NVC; i += 4) {
          if (c * kChunk + i * 64 * EPV >= V) break;
          float y[EPV];
          int e0;
          rv.vec(c, i, lane, y, e0);
#pragma unroll
          for (int k = 0; k < EPV; ++k) {
            const int j = e0 + k;
            const bool ok = j < V && y[k] > kNegInf;
            const float e = chunk_term(y[k], magicN);
            const float g = ok ? e / E[ok ? j : 0] : -1.0f;
            const bool better = g > best;  // strict: the first maximum in this lane's (increasing) order stays
            sec = better ? best : fmaxf(sec, g);
            best = better ? g : best;
            bj = better ? j : bj;
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
  }
  if (lane == 0 && writer) p.out_token[pidx] = tok;
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
    chunk_sums<DT, false>(x, kMagic - Nc, nv_valid, nullptr, pA, pB, pAm, pBm);
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
// mask preparation: bit rows [n_masks, mask_ld] -> transposed lane words + sparse-id lists.
// Block (k, c) with c < nch transposes chunk c of mask k; block (k, nch) counts mask k and lists its ids.
// ---------------------------------------------------------------------------------------------------------
template <int EPV>
__global__ __launch_bounds__(256) void mask_prepare_kernel(const uint32_t *bits, int64_t mask_ld, int V, int nch,
                                                          uint64_t *mask_t, int32_t *mask_info) {
  const int k = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t *row = bits + (int64_t)k * mask_ld;
  if (c < nch) {
    uint64_t *dst = mask_t + ((int64_t)k * nch + c) * 64;
    bool on[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // word w = wave + 4j = vector (w / EPV), component (w % EPV); loads first
      const int w = wave + 4 * j, i = w / EPV, kk = w % EPV;
      const int e = c * kChunk + (i * 64 + lane) * EPV + kk;
      on[j] = e < V && ((row[e < V ? e >> 5 : 0] >> (e & 31)) & 1u);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t m = __ballot(on[j]);
      if (lane == 0) dst[wave + 4 * j] = m;
    }
    return;
  }
  // count + ordered id list (ids only matter when the count is at most kInfoWords - 1).  All words of the row are
  // requested before the first is used (one memory latency, not one per 256-word tile).
  __shared__ int s_cnt[4];
  int32_t *info = mask_info + (int64_t)k * kInfoWords;
  const int nw = (V + 31) >> 5;
  constexpr int kTile = 16;  // words per thread per sweep: 4096 words = 131072 tokens in one sweep
  int base = 0;
  for (int w0 = 0; w0 < nw; w0 += 256 * kTile) {
    uint32_t word[kTile];
    int pc = 0;
#pragma unroll
    for (int j = 0; j < kTile; ++j) {  // thread t owns the contiguous words w0 + t*kTile .. + kTile - 1
      const int w = w0 + tid * kTile + j;
      uint32_t v = w < nw ? row[w] : 0u;
      const int rem = V - w * 32;
      if (w < nw && rem < 32) v &= (1u << rem) - 1u;
      word[j] = v;
      pc += __popc(v);
    }
    int incl = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_cnt[wave] = incl;
    __syncthreads();
    int wb = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < wave) wb += s_cnt[q];
      tot += s_cnt[q];
    }
    int at = base + wb + incl - pc;
    // single-sweep masks (V <= 131072): the total is known here, dense masks skip the id walk altogether
    const bool may_list = nw > 256 * kTile || tot <= kInfoWords - 1;
    if (may_list && at < kInfoWords - 1) {
#pragma unroll
      for (int j = 0; j < kTile; ++j) {
        uint32_t v = word[j];
        while (v && at < kInfoWords - 1) {
          const int bpos = __ffs(v) - 1;
          v &= v - 1;
          if (at < kInfoWords - 1) info[1 + at] = (w0 + tid * kTile + j) * 32 + bpos;
          ++at;
        }
      }
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) info[0] = base;
}

}  // namespace glb
