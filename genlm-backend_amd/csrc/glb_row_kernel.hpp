// glb_row_kernel.hpp — the fused particle-step kernel (one workgroup per particle, logits row held
// in VGPRs, single HBM pass).  gfx950 / wave64 only.
//
// Replaces, per particle (reference = genlm/genlm-backend):
//   cache.py:96      logps  = log_softmax(logits_row)
//   README.md:84-87  masked = logps + mask; logZ = logsumexp(masked); token = multinomial(exp(masked - logZ))
//   base.py:136-141  the same draw with logits scaled by 1/temperature
//
// Layout: the row is addressed from its 16-byte-aligned-down start (`a` leading pad elements) in
// 16-byte vectors of EPV elements.  Wave w owns the contiguous vectors [w*64*NVL, (w+1)*64*NVL);
// its k-th load instruction reads vectors w*64*NVL + k*64 + lane (1 KiB contiguous per
// wave-instruction).  All NVL loads of a lane are issued before the first use, so a 1024-thread
// workgroup keeps the whole row (<= 256 KiB) in flight.  An aligned 16-byte vector that overlaps
// the row by at least one byte lies in the same page as that byte, so loading it never faults;
// elements outside [0, V) are replaced by -inf after the load.
//
// Phases: (1) row max / masked max -> exponents N; (2) fixed-point sums S_all, S_mask (GLB math,
// glb_math.hpp) and, by mode, the exponential race or the per-tile partial sums for the inverse
// CDF; (3) one lane turns the sums into lse / logZ in double and the draw is located by
// wave -> tile -> lane -> element prefix search in vocabulary order.
#pragma once
#include <type_traits>

#include "glb_math.hpp"

namespace glb {

enum { kDtF32 = 0, kDtBf16 = 1, kDtF16 = 2 };
enum { kMaskNone = 0, kMaskBits = 1, kMaskF32 = 2 };
enum { kModeStats = 0, kModePhilox = 1, kModeNoise = 2 };

struct RowParams {
  const void *logits;
  int64_t ld;
  int32_t V;
  int32_t use_scale;
  float scale;
  int32_t n_particles;
  const int32_t *row_of;
  const void *mask;
  int64_t mask_ld;
  const int32_t *mask_id;
  int32_t n_masks;
  const float *noise;
  int64_t noise_ld;
  uint64_t seed, offset;
  int64_t particle_base;
  float *out_logZ;
  float *out_lse;
  int32_t *out_token;
  float *out_logprobs;  // stats mode only: [n_particles, out_ld]
  int64_t out_ld;
  // persistent kernel: per-particle sums and exponents parked between its streaming loop and its tail
  uint64_t *row_sums;  // [n_particles][2]  S_all, S_mask
  float *row_exps;     // [n_particles][2]  N_all, N_mask
};

template <int DT>
struct ElemTraits {
  static constexpr int EPV = DT == kDtF32 ? 4 : 8;
  static constexpr int ES = DT == kDtF32 ? 4 : 2;
};

// unpack one 16-byte vector into EPV floats
template <int DT>
__device__ __forceinline__ void unpack_vec(const uint4 &r, float *x) {
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

// Funnel-extract the EPV mask bits of elements j0 .. j0+EPV-1 from a packed bit row.  Bits of
// elements outside [0, V) are don't-care (their x is -inf).  d = max(-j0, 0) only matters for the
// first vector of a misaligned row.
template <int EPV>
__device__ __forceinline__ uint32_t mask_nibble(const uint32_t *mrow, int32_t n_words, int32_t j0) {
  const int32_t jb = j0 < 0 ? 0 : j0;
  const int32_t d = jb - j0;
  const int32_t wi = jb >> 5;
  const int32_t w0 = wi < n_words ? wi : n_words - 1;
  const int32_t w1 = wi + 1 < n_words ? wi + 1 : n_words - 1;
  const uint32_t lo = mrow[w0], hi = mrow[w1];
  const uint32_t f = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(jb & 31));  // ({hi,lo} >> sh)[31:0]
  return (f << d) & ((1u << EPV) - 1u);
}

// Zero-instruction barrier: the compiler may not reuse values derived from `r` before this point
// (prevents it from carrying 4*NVL unpacked / scaled / bit-filled copies of the row across phases).
__device__ __forceinline__ void opaque(uint4 &r) {
  asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
}
__device__ __forceinline__ void opaque(uint32_t &r) { asm volatile("" : "+v"(r)); }

__device__ __forceinline__ void opaque(uint64_t &r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void opaque(float &r) { asm volatile("" : "+v"(r)); }

// q gated by an all-ones / all-zeros word, half by half (a 64-bit `q & splat(fill)` is turned into a
// 64-bit multiply by 0x100000001 by instcombine)
__device__ __forceinline__ uint64_t mask_u64(uint64_t q, uint32_t fill) {
  const uint32_t lo = (uint32_t)q & fill, hi = (uint32_t)(q >> 32) & fill;
  return ((uint64_t)hi << 32) | lo;
}

// all-ones / all-zeros word from bit `pos` of w (v_bfe_i32)
__device__ __forceinline__ uint32_t bit_fill(uint32_t w, int pos) {
  return (uint32_t)(((int32_t)(w << (31 - pos))) >> 31);
}

template <int DT, int MASK, int MODE, int NVL, int T>
__global__ __launch_bounds__(T) void row_kernel(const RowParams p) {
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr int W = T / 64;
  constexpr int MBW = (NVL * EPV + 31) / 32;
  constexpr bool kPhiloxMode = MODE == kModePhilox;
  constexpr bool kPhilox = MODE == kModePhilox;
  constexpr bool kNoise = MODE == kModeNoise;

  __shared__ float s_max[2][W];
  __shared__ uint64_t s_sum[2][W];
  __shared__ float s_bestg[W];
  __shared__ int32_t s_bestj[W];
  __shared__ float s_lse;
  // per-(tile, lane) masked partial sums for the inverse-CDF search (philox mode only)
  __shared__ uint64_t s_asum[kPhilox ? NVL * T : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // XCD-aware particle assignment: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous particle range so particles that share a logits row (dedup fan-out) share an L2.
  int pidx;
  {
    const int n = p.n_particles, b = blockIdx.x;
    const int q = n >> 3, r = n & 7, xcd = b & 7, i = b >> 3;
    pidx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int row = p.row_of ? p.row_of[pidx] : pidx;
  const int V = p.V;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int a = (int)(((uintptr_t)rowp) & 15) / ES;
  const char *base = rowp - a * ES;
  const int nv = (V + a + EPV - 1) / EPV;
  const int v0 = wave * (64 * NVL) + lane;

  // ---- issue every load of the row ------------------------------------------------------------
  uint4 raw[NVL];
#pragma unroll
  for (int k = 0; k < NVL; ++k) {
    int v = v0 + k * 64;
    v = v < nv ? v : nv - 1;
    raw[k] = *reinterpret_cast<const uint4 *>(base + (int64_t)v * 16);
  }

  // mask bits of this lane's elements (issued behind the row loads; the table is L2 resident)
  uint32_t mb[MBW];
#pragma unroll
  for (int i = 0; i < MBW; ++i) mb[i] = 0xffffffffu;
  const float *mrow_f = nullptr;
  const uint32_t *mrow_b = nullptr;
  if constexpr (MASK == kMaskBits) {
    const int mi = p.mask_id ? p.mask_id[pidx] : (p.n_masks == 1 ? 0 : pidx);
    mrow_b = (const uint32_t *)p.mask + (int64_t)mi * p.mask_ld;
    const int n_words = (V + 31) >> 5;
#pragma unroll
    for (int i = 0; i < MBW; ++i) mb[i] = 0;
#pragma unroll
    for (int k = 0; k < NVL; ++k) {
      const int v = v0 + k * 64;
      const uint32_t nib = mask_nibble<EPV>(mrow_b, n_words, v * EPV - a);
      mb[(k * EPV) >> 5] |= nib << ((k * EPV) & 31);
    }
  } else if constexpr (MASK == kMaskF32) {
    const int mi = p.mask_id ? p.mask_id[pidx] : (p.n_masks == 1 ? 0 : pidx);
    mrow_f = (const float *)p.mask + (int64_t)mi * p.mask_ld;
  }

  // ---- invalidate out-of-row elements (in the packed registers: -inf bit patterns) ------------------
#pragma unroll
  for (int k = 0; k < NVL; ++k) {
    const int tile_lo = wave * (64 * NVL) + k * 64;
    if ((tile_lo == 0 && a > 0) || tile_lo + 64 >= nv) {
      const int j0 = (v0 + k * 64) * EPV - a;
      uint32_t w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        if ((uint32_t)(j0 + c) >= (uint32_t)V) {
          if constexpr (DT == kDtF32) w[c] = 0xff800000u;
          else if constexpr (DT == kDtBf16)
            w[c >> 1] = (c & 1) ? ((w[c >> 1] & 0x0000ffffu) | 0xff800000u)
                                : ((w[c >> 1] & 0xffff0000u) | 0x0000ff80u);
          else
            w[c >> 1] = (c & 1) ? ((w[c >> 1] & 0x0000ffffu) | 0xfc000000u)
                                : ((w[c >> 1] & 0xffff0000u) | 0x0000fc00u);
        }
      }
      raw[k] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }

  // ---- phase 1: maxima ------------------------------------------------------------------------
  float m_all = kNegInf, m_msk = kNegInf;
  uint32_t v0a = (uint32_t)v0;  // per-phase opaque copies: element indices are re-derived, not kept live
  opaque(v0a);
#pragma unroll
  for (int k = 0; k < NVL; ++k) {
    float xk[EPV];
    unpack_vec<DT>(raw[k], xk);
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      const float xv = p.use_scale ? xk[c] * p.scale : xk[c];
      m_all = fmaxf(m_all, xv);
      if constexpr (MASK == kMaskBits) {
        // allowed ? x : -inf  as one v_bfi_b32 on the bit pattern
        const uint32_t fill = bit_fill(mb[(k * EPV + c) >> 5], (k * EPV + c) & 31);
        const uint32_t sel = (__float_as_uint(xv) & fill) | (0xff800000u & ~fill);
        m_msk = fmaxf(m_msk, __uint_as_float(sel));
      } else if constexpr (MASK == kMaskF32) {
        int j = ((int)v0a + k * 64) * EPV - a + c;
        j = j < 0 ? 0 : (j >= V ? V - 1 : j);
        m_msk = fmaxf(m_msk, xv + mrow_f[j]);
      }
    }
  }
  m_all = wave_max(m_all);
  if constexpr (MASK != kMaskNone) m_msk = wave_max(m_msk);
  if (lane == 0) {
    s_max[0][wave] = m_all;
    if constexpr (MASK != kMaskNone) s_max[1][wave] = m_msk;
  }
  __syncthreads();
  m_all = s_max[0][0];
#pragma unroll
  for (int w = 1; w < W; ++w) m_all = fmaxf(m_all, s_max[0][w]);
  if constexpr (MASK != kMaskNone) {
    m_msk = s_max[1][0];
#pragma unroll
    for (int w = 1; w < W; ++w) m_msk = fmaxf(m_msk, s_max[1][w]);
  } else {
    m_msk = m_all;
  }
  const float N_all = exp_n(m_all);
  const float N_msk = exp_n(m_msk);
  const float Nb_all = N_all + (float)kFixShift, Nb_msk = N_msk + (float)kFixShift;
  // Masked sums (GLB math): on the row's scale N_all - there the masked term of a bit mask is the
  // unmasked term gated by the mask bit - unless the masked maximum lies above it (additive masks only)
  // or the result keeps fewer than 37 bits (allowed mass < 2^-7 of the row; block-uniform, rare): then
  // the loop body runs a second time on the masked maximum's own scale N_msk.
  const bool over = (MASK == kMaskF32) && (N_msk > N_all);
  float Nb_m = over ? Nb_msk : Nb_all;  // scale of the masked terms in this pass
  float N_fin = over ? N_msk : N_all;   // ... and of the masked sum finally kept

  // ---- phase 2: fixed-point sums (+ race / per-tile partial sums) ----------------------------------
  uint64_t S_all = 0, S_msk = 0;
  float best_g = -1.0f;
  int32_t best_j = 0x7fffffff;
  const float *noise_row = nullptr;
  if constexpr (kNoise) noise_row = p.noise + (int64_t)pidx * p.noise_ld;
#pragma unroll 1
  for (int pass = 0;; ++pass) {
    const float Nb_q = (MASK == kMaskF32) ? Nb_all : Nb_m;
    uint64_t acc = 0, s_msk = 0;
    // defeat loop-invariant hoisting of the per-element exp splits / bit fills (100+ VGPRs), and keep the
    // compiler from carrying the per-element bit fills of phase 1 across the barrier (50+ VGPRs)
#pragma unroll
    for (int i = 0; i < MBW; ++i) opaque(mb[i]);
#pragma unroll
    for (int k = 0; k < NVL; ++k) opaque(raw[k]);
    uint32_t v0b = (uint32_t)v0;
    opaque(v0b);
    // (measured: with 4 waves per SIMD the hand-interleaved exp_fix4 of glb_math.hpp is SLOWER here - 105.7
    //  vs 95.8 us on the 1024 x 50257 masked case; the hardware already hides the dependent chains.)
#pragma unroll
    for (int k = 0; k < NVL; ++k) {
      uint64_t ak = 0;
      float xk[EPV];
      unpack_vec<DT>(raw[k], xk);
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        const float xv = p.use_scale ? xk[c] * p.scale : xk[c];
        float nf, P;
        exp_parts(xv, nf, P);
        const uint64_t q = fix_term_from_parts(nf, P, Nb_q);
        acc += q;
        float nfy = nf, Py = P, yv = xv;
        if constexpr (MASK == kMaskNone) {
          if constexpr (kPhilox) ak += q;
        } else if constexpr (MASK == kMaskBits) {
          const uint32_t fill = bit_fill(mb[(k * EPV + c) >> 5], (k * EPV + c) & 31);
          ak += mask_u64(q, fill);
          if constexpr (kNoise) yv = __uint_as_float((__float_as_uint(xv) & fill) | (0xff800000u & ~fill));
        } else {
          int j = ((int)v0b + k * 64) * EPV - a + c;
          j = j < 0 ? 0 : (j >= V ? V - 1 : j);
          yv = xv + mrow_f[j];
          exp_parts(yv, nfy, Py);
          ak += fix_term_from_parts(nfy, Py, Nb_m);
        }
        if constexpr (kNoise) {
          const int jr = ((int)v0b + k * 64) * EPV - a + c;
          const int jc = jr < 0 ? 0 : (jr >= V ? V - 1 : jr);
          const float E = noise_row[jc];
          // the race is always run against the masked maximum's exponent (branch-free: selects, not lane branches)
          const bool live = (yv > kNegInf) && pass == 0;
          const float df = live ? nfy - N_msk : 0.0f;
          const float e = (df < -100.0f) ? 0.0f : __builtin_ldexpf(Py, (int)df - 30);
          const float g = live ? e / E : -1.0f;
          const bool better = g > best_g;
          best_g = better ? g : best_g;
          best_j = better ? jr : best_j;
        }
      }
      s_msk += ak;
      if constexpr (kPhilox) s_asum[k * T + tid] = ak;
      // one vector at a time: without this the scheduler interleaves many vectors and spills
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MASK == kMaskNone) s_msk = acc;
    {
      const uint64_t t_all = wave_scan_u64(acc);
      uint64_t t_msk = t_all;
      if constexpr (MASK != kMaskNone) t_msk = wave_scan_u64(s_msk);
      if (lane == 63) {
        s_sum[0][wave] = t_all;
        s_sum[1][wave] = t_msk;
      }
    }
    if constexpr (kNoise) {
      if (pass == 0) {
        // wave argmax (max g, then min j) through LDS-free DPP would need 2 fields; 6 xor steps suffice
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float og = __shfl_xor(best_g, o, 64);
          const int32_t oj = __shfl_xor(best_j, o, 64);
          if (og > best_g || (og == best_g && oj < best_j)) {
            best_g = og;
            best_j = oj;
          }
        }
        if (lane == 0) {
          s_bestg[wave] = best_g;
          s_bestj[wave] = best_j;
        }
      }
    }
    __syncthreads();
    uint64_t Sa = 0, Sm = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      Sa += s_sum[0][w];
      Sm += s_sum[1][w];
    }
    if (pass == 0) S_all = Sa;
    S_msk = Sm;
    N_fin = (pass == 0 && !over) ? N_all : N_msk;
    if constexpr (MASK == kMaskNone) break;
    uint32_t top = (uint32_t)(Sm >> 37);  // sums stay below 2^62
    opaque(top);                          // VALU compare (uniform u64 `<` miscompile, see the draw below)
    if (pass == 1 || over || top != 0u || !(m_msk > kNegInf)) break;
    __syncthreads();  // s_sum is rewritten by the second pass
    Nb_m = Nb_msk;
  }
  const float Nb_fin = N_fin + (float)kFixShift;

  // ---- phase 3: lse / logZ ---------------------------------------------------------------------
  if (tid == 0) {
    const double lse_all = S_all ? log_fix(S_all, (int32_t)N_all - kFixFrac) : (double)kNegInf;
    const double lse_msk = S_msk ? log_fix(S_msk, (int32_t)N_fin - kFixFrac) : (double)kNegInf;
    if (p.out_lse) p.out_lse[pidx] = (float)lse_all;
    if (p.out_logZ) p.out_logZ[pidx] = (float)(lse_msk - lse_all);
    if constexpr (MODE == kModeStats) s_lse = (float)lse_all;
    if constexpr (kNoise) {
      if (p.out_token) {
        float g = s_bestg[0];
        int32_t j = s_bestj[0];
#pragma unroll
        for (int w = 1; w < W; ++w) {
          const float og = s_bestg[w];
          const int32_t oj = s_bestj[w];
          if (og > g || (og == g && oj < j)) {
            g = og;
            j = oj;
          }
        }
        p.out_token[pidx] = (S_msk != 0 && g >= 0.0f) ? j : -1;
      }
    }
    if constexpr (kPhilox) {
      if (p.out_token && S_msk == 0) p.out_token[pidx] = -1;
    }
  }

  if constexpr (MODE == kModeStats) {
    if (p.out_logprobs) {
      __syncthreads();
      const float lse = s_lse;
      float *orow = p.out_logprobs + (int64_t)pidx * p.out_ld;
      const bool vec_ok = ((((uintptr_t)(orow - a)) & 15) == 0);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int v = v0 + k * 64;
        const int j0 = v * EPV - a;
        float xk[EPV];
        unpack_vec<DT>(raw[k], xk);
        if (p.use_scale) {
#pragma unroll
          for (int c = 0; c < EPV; ++c) xk[c] = xk[c] * p.scale;
        }
        if (v < nv) {
          if (vec_ok && j0 >= 0 && j0 + EPV <= V) {
#pragma unroll
            for (int h = 0; h < EPV / 4; ++h) {
              float4 o;
              o.x = xk[4 * h + 0] - lse;
              o.y = xk[4 * h + 1] - lse;
              o.z = xk[4 * h + 2] - lse;
              o.w = xk[4 * h + 3] - lse;
              *reinterpret_cast<float4 *>(orow + j0 + 4 * h) = o;
            }
          } else {
#pragma unroll
            for (int c = 0; c < EPV; ++c)
              if ((uint32_t)(j0 + c) < (uint32_t)V) orow[j0 + c] = xk[c] - lse;
          }
        }
      }
    }
  }

  // ---- inverse-CDF draw: wave -> tile -> lane -> element, vocabulary order ---------------------------
  if constexpr (kPhilox) {
    if (p.out_token && S_msk != 0) {
      // the draw is wave-uniform work: find the wave first, only that wave continues
      uint64_t Tw = 0;
      int wsel = W - 1;
      {
        const uint64_t gp = (uint64_t)(p.particle_base + pidx);
        const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset,
                                 (uint32_t)(p.offset >> 32)};
        const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
        uint32_t rnd[4];
        philox4x32_10(ctr, key, rnd);
        const uint64_t R = ((uint64_t)rnd[1] << 32) | rnd[0];
        Tw = __umul64hi(R, S_msk);  // uniform integer in [0, S_msk)
        // Keep the search in VALU registers: hipcc (ROCm 7.2) lowers a wave-uniform `u64 < u64`
        // feeding a select to v_cmp + s_cselect without materialising SCC (wrong wave picked).
        {
          uint32_t z = 0;
          opaque(z);
          Tw += z;
        }
        uint64_t run = 0;
        bool found = false;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const uint64_t cw = s_sum[1][w];
          if (!found && Tw < run + cw) {
            found = true;
            wsel = w;
            Tw -= run;
          }
          run += cw;
        }
      }
      if (wave == wsel) {
        // binary search over this wave's tiles: sum of tiles [lo, mid) by lane-local adds + one scan
        int lo = 0, hi = NVL;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          uint64_t part = 0;
          for (int k = lo; k < mid; ++k) part += s_asum[k * T + tid];
          const uint64_t c = wave_sum_u64(part);
          if (Tw < c) {
            hi = mid;
          } else {
            Tw -= c;
            lo = mid;
          }
        }
        const int ksel = lo;
        const uint64_t asel = s_asum[ksel * T + tid];
        const uint64_t incl = wave_scan_u64(asel);
        const unsigned long long ball = __ballot(incl > Tw);
        const int lsel = __ffsll((long long)ball) - 1;
        if (lane == lsel) {
          uint64_t Tl = Tw - (incl - asel);
          const int v = v0 + ksel * 64;  // < nv because its tile sum is non-zero
          const uint4 rr = *reinterpret_cast<const uint4 *>(base + (int64_t)v * 16);
          float xs[EPV];
          unpack_vec<DT>(rr, xs);
          const int j0 = v * EPV - a;
          uint32_t nib = (1u << EPV) - 1u;
          if constexpr (MASK == kMaskBits) nib = mask_nibble<EPV>(mrow_b, (V + 31) >> 5, j0);
          int32_t tok = -1;
#pragma unroll
          for (int c = 0; c < EPV; ++c) {
            float xv = xs[c];
            if (p.use_scale) xv = xv * p.scale;
            const int j = j0 + c;
            bool ok = (uint32_t)j < (uint32_t)V;
            if constexpr (MASK == kMaskBits) ok = ok && ((nib >> c) & 1u);
            if constexpr (MASK == kMaskF32) {
              const int jc = j < 0 ? 0 : (j >= V ? V - 1 : j);
              xv = xv + mrow_f[jc];
            }
            const uint64_t q = ok ? fix_term(xv, Nb_fin) : 0ull;
            if (tok < 0) {
              if (Tl < q) tok = j;
              else Tl -= q;
            }
          }
          p.out_token[pidx] = tok;
        }
      }
    }
  }
}

}  // namespace glb
