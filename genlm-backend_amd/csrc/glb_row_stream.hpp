// glb_row_stream.hpp — fallback for rows too long for the register-resident kernels (fp32 V > 65 528,
// 16-bit V > 131 064: e.g. 256k-token vocabularies, or fp32 logits at V = 128 256).
// One 1024-thread workgroup per particle streams the row from memory three times (maximum, sums,
// and — for a draw — one more partial pass); the second and third pass hit L2 / Infinity Cache when
// the row fits.  Same GLB math, same vocabulary-order inverse CDF: bit-identical to the other kernels.
#pragma once
#include "glb_row_kernel.hpp"

namespace glb {

template <int DT, int MASK, int MODE>
__global__ __launch_bounds__(1024) void row_stream_kernel(const RowParams p) {
  constexpr int T = 1024, W = 16;
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  static_assert(MODE != kModeNoise && MASK != kMaskF32, "streaming fallback: mask none/bits, stats/philox");
  __shared__ float s_max[2][W];
  __shared__ uint64_t s_sum[2][W];
  __shared__ uint64_t s_pre[W];
  __shared__ float s_lse;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pidx = blockIdx.x;
  const int V = p.V;
  const int row = p.row_of ? p.row_of[pidx] : pidx;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int a = (int)(((uintptr_t)rowp) & 15) / ES;
  const char *base = rowp - a * ES;
  const int nv = (V + a + EPV - 1) / EPV;
  const uint32_t *mrow = nullptr;
  if constexpr (MASK == kMaskBits) {
    const int mi = p.mask_id ? p.mask_id[pidx] : (p.n_masks == 1 ? 0 : pidx);
    mrow = (const uint32_t *)p.mask + (int64_t)mi * p.mask_ld;
  }
  // wave w owns the contiguous vectors [w * per_wave, (w+1) * per_wave): vocabulary order = wave-major
  const int per_wave = ((nv + W - 1) / W + 63) / 64 * 64;
  const int v_lo = wave * per_wave, v_hi = (v_lo + per_wave < nv) ? v_lo + per_wave : nv;

  // element c of vector v: value and "allowed" flag (out-of-row elements are -inf / not allowed)
  auto elems = [&](int v, float (&x)[EPV], uint32_t &nib) {
    const uint4 rk = *reinterpret_cast<const uint4 *>(base + (int64_t)v * 16);
    unpack_vec<DT>(rk, x);
    const int j0 = v * EPV - a;
    nib = (1u << EPV) - 1u;
    if constexpr (MASK == kMaskBits) nib = mask_nibble<EPV>(mrow, (V + 31) >> 5, j0);
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      if (p.use_scale) x[c] = x[c] * p.scale;
      if ((uint32_t)(j0 + c) >= (uint32_t)V) {
        x[c] = kNegInf;
        nib &= ~(1u << c);
      }
    }
  };

  // pass 1: maxima
  float m_all = kNegInf, m_msk = kNegInf;
  for (int v = v_lo + lane; v < v_hi; v += 64) {
    float x[EPV];
    uint32_t nib;
    elems(v, x, nib);
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      m_all = fmaxf(m_all, x[c]);
      m_msk = fmaxf(m_msk, ((nib >> c) & 1u) ? x[c] : kNegInf);
    }
  }
  m_all = wave_max(m_all);
  m_msk = wave_max(m_msk);
  if (lane == 0) {
    s_max[0][wave] = m_all;
    s_max[1][wave] = m_msk;
  }
  __syncthreads();
  m_all = s_max[0][0];
  m_msk = s_max[1][0];
  for (int w = 1; w < W; ++w) {
    m_all = fmaxf(m_all, s_max[0][w]);
    m_msk = fmaxf(m_msk, s_max[1][w]);
  }
  const float N_all = exp_n(m_all), N_msk = exp_n(m_msk);
  const float Nb_all = N_all + (float)kFixShift, Nb_msk = N_msk + (float)kFixShift;

  // pass 2: fixed-point sums.  GLB math keeps the masked sum on the row's scale N_all unless that leaves it
  // fewer than 37 bits; a streaming kernel cannot cheaply come back, so when the two exponents differ it takes
  // the masked sum on both scales in the same pass and chooses afterwards.
  const bool two_scales = (N_msk != N_all);  // bit masks: N_msk <= N_all
  uint64_t s_all = 0, s_cmn = 0, s_own = 0;
  for (int v = v_lo + lane; v < v_hi; v += 64) {
    float x[EPV];
    uint32_t nib;
    elems(v, x, nib);
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      // (explicit guard: x was SET to -inf above, and the float->uint conversion of the resulting NaN is
      //  something the compiler may fold to anything; the other kernels carry -inf as opaque bit patterns)
      const uint64_t q = (x[c] > kNegInf) ? fix_term(x[c], Nb_all) : 0ull;
      s_all += q;
      const bool ok = ((nib >> c) & 1u) && (x[c] > kNegInf);
      s_cmn += ok ? q : 0ull;
      if (two_scales) s_own += ok ? fix_term(x[c], Nb_msk) : 0ull;
    }
  }
  const uint64_t t_all = wave_sum_u64(s_all), t_cmn = wave_sum_u64(s_cmn), t_own = wave_sum_u64(s_own);
  if (lane == 0) {
    s_sum[0][wave] = t_all;
    s_sum[1][wave] = t_cmn;
    s_pre[wave] = t_own;
  }
  __syncthreads();
  uint64_t S_all = 0, S_cmn = 0, S_own = 0;
  for (int w = 0; w < W; ++w) {
    S_all += s_sum[0][w];
    S_cmn += s_sum[1][w];
    S_own += s_pre[w];
  }
  uint32_t top = (uint32_t)(S_cmn >> 37);
  opaque(top);  // VALU compare (uniform u64 `<` miscompile, see v1)
  const bool own = two_scales && top == 0u && (m_msk > kNegInf);
  const uint64_t S_msk = own ? S_own : S_cmn;
  const float N_fin = own ? N_msk : N_all;
  const float Nb_fin = N_fin + (float)kFixShift;
  // the draw walks the waves' sums on the chosen scale
  __syncthreads();
  if (own && lane == 0) s_sum[1][wave] = t_own;
  __syncthreads();
  if (tid == 0) {
    const double lse_all = S_all ? log_fix(S_all, (int32_t)N_all - kFixFrac) : (double)kNegInf;
    const double lse_msk = S_msk ? log_fix(S_msk, (int32_t)N_fin - kFixFrac) : (double)kNegInf;
    if (p.out_lse) p.out_lse[pidx] = (float)lse_all;
    if (p.out_logZ) p.out_logZ[pidx] = (float)(lse_msk - lse_all);
    s_lse = (float)lse_all;
    if (MODE == kModePhilox && p.out_token && S_msk == 0) p.out_token[pidx] = -1;
  }
  if constexpr (MODE == kModeStats) {
    if (p.out_logprobs) {
      __syncthreads();
      const float lse = s_lse;
      float *orow = p.out_logprobs + (int64_t)pidx * p.out_ld;
      for (int v = v_lo + lane; v < v_hi; v += 64) {
        float x[EPV];
        uint32_t nib;
        elems(v, x, nib);
        const int j0 = v * EPV - a;
#pragma unroll
        for (int c = 0; c < EPV; ++c)
          if ((uint32_t)(j0 + c) < (uint32_t)V) orow[j0 + c] = x[c] - lse;
      }
    }
  }
  if constexpr (MODE == kModePhilox) {
    if (p.out_token && S_msk != 0) {
      const uint64_t gp = (uint64_t)(p.particle_base + pidx);
      const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
      const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
      uint32_t rnd[4];
      philox4x32_10(ctr, key, rnd);
      uint64_t Tw = __umul64hi(((uint64_t)rnd[1] << 32) | rnd[0], S_msk);
      {
        uint32_t z = 0;
        opaque(z);
        Tw += z;
      }
      int wsel = W - 1;
      {
        uint64_t run = 0;
        bool found = false;
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
      if (wave == wsel) {  // third pass over this wave's part only: tile by tile until the prefix passes T
        bool done = false;
        for (int vb = v_lo; vb < v_hi && !done; vb += 64) {
          const int v = vb + lane;
          float x[EPV];
          uint32_t nib = 0;
          uint64_t q[EPV];
          uint64_t aj = 0;
          if (v < v_hi) {
            elems(v, x, nib);
#pragma unroll
            for (int c = 0; c < EPV; ++c) {
              q[c] = ((nib >> c) & 1u) ? fix_term(x[c], Nb_fin) : 0ull;
              aj += q[c];
            }
          } else {
#pragma unroll
            for (int c = 0; c < EPV; ++c) q[c] = 0;
          }
          const uint64_t incl = wave_scan_u64(aj);
          const uint64_t tile = readlane_u64(incl, 63);
          if (Tw < tile) {
            const int lsel = __ffsll((long long)__ballot(incl > Tw)) - 1;
            if (lane == lsel) {
              uint64_t Tl = Tw - (incl - aj);
              int32_t tok = -1;
#pragma unroll
              for (int c = 0; c < EPV; ++c) {
                if (tok < 0) {
                  if (Tl < q[c]) tok = v * EPV - a + c;
                  else Tl -= q[c];
                }
              }
              p.out_token[pidx] = tok;
            }
            done = true;
          } else {
            Tw -= tile;
          }
        }
      }
    }
  }
}

}  // namespace glb
