// glb_row_kernel_v3.hpp — cooperative split-row variant of the fused particle-step kernel.
//
// v1/v2 give a whole logits row (201 KB at gpt2/fp32) to one workgroup.  With 1024 particles on 256
// CUs that is a 4-deep pipeline per CU: its fill (first row: memory only) and drain (last row:
// arithmetic only) are 2 of 5 stages, and a CU pulls at most ~24 GB/s.  v3 splits every row over a
// cluster of 4 workgroups (= 4 CUs), so each CU streams 16 quarter-rows ("items", ~50 KB) through a
// 4-stage register pipeline and the fill/drain cost shrinks to ~1/8 of the launch:
//
//   step s:   issue loads of item s          (7 x 16 B per lane, straight into VGPRs, one load per
//                                             tile of arithmetic so the memory queue never blocks)
//             local maxima of item s-1       -> 8-byte {tag, value} granules published to the cluster
//             (granules of item s-2 are in flight: issued last step, consumed next step)
//             fixed-point sums of item s-3   against the cluster-wide row exponent
//
// The only inter-workgroup exchange is the row maximum (the sums are exact integers relative to the
// shared exponent, so the four partial sums are simply added later by finish_kernel / locate_kernel).
// Exchange = recipe R2 of cdna_hip_programming.md Guideline 16: one naturally aligned 8-byte
// {tag = 1, value} granule per (particle, member, which), written by ONE relaxed agent-scope atomic
// store (sc1, write-through) and polled with relaxed agent-scope atomic loads (sc1, L1 bypass): no
// fences, correct for any workgroup -> XCD placement.  All 256 workgroups are co-resident (512
// threads, >128 VGPRs: one per CU), every spin is bounded, and finish/locate re-zero the granules.
//
// Arithmetic is GLB math: bit-identical to v1, v2 and the oracle.
#pragma once
#include "glb_row_kernel_v2.hpp"

namespace glb {

typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int kClusterSize = 4;
constexpr unsigned kSpinLimit = 1u << 18;

struct V3Params {
  RowParams rp;
  unsigned long long *xch;   // [n_particles][4 members][2] granules (zero before launch)
  uint64_t *wave_sums;       // [n_particles][4 members][8 waves][2]  (S_all, S_msk partials)
  float *row_exps;           // [n_particles][2]  N_all, N_msk
  unsigned int *timeout;     // set non-zero if a spin gave up
  int n_clusters;
};

template <int DT, int MASK, int NVI, bool SCALED>
__global__ __launch_bounds__(512) void row_kernel_v3(const V3Params q) {
  constexpr int T = 512, W = 8;
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr bool kBits = MASK == kMaskBits;
  constexpr int MW = (NVI * EPV + 31) / 32;  // mask words per item and lane
  const RowParams &p = q.rp;

  __shared__ float s_wmax[2][2][W];  // [step parity][all | masked][wave]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cluster = blockIdx.x >> 2, member = blockIdx.x & 3;
  const int V = p.V, n = p.n_particles;
  const int R = cluster < n ? (n - cluster + q.n_clusters - 1) / q.n_clusters : 0;  // items of this workgroup
  const int v_item = wave * (64 * NVI) + lane;  // lane's first vector inside an item

  struct Item {
    const char *base;       // aligned-down row start
    const uint32_t *mrow;   // mask bit row
    int a, nv, vlo, vhi;    // pad elements, vectors in row, this member's vector range [vlo, vhi)
    int pidx;
  };
  auto item_of = [&](int i) {
    Item it;
    it.pidx = cluster + i * q.n_clusters;
    const int row = p.row_of ? p.row_of[it.pidx] : it.pidx;
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    it.a = (int)(((uintptr_t)rowp) & 15) / ES;
    it.base = rowp - it.a * ES;
    it.nv = (V + it.a + EPV - 1) / EPV;
    const int Q = (it.nv + kClusterSize - 1) / kClusterSize;
    it.vlo = member * Q;
    it.vhi = it.vlo + Q < it.nv ? it.vlo + Q : it.nv;
    it.mrow = nullptr;
    if constexpr (kBits) {
      const int mi = p.mask_id ? p.mask_id[it.pidx] : (p.n_masks == 1 ? 0 : it.pidx);
      it.mrow = (const uint32_t *)p.mask + (int64_t)mi * p.mask_ld;
    }
    return it;
  };

  uint4 buf[4][NVI];
  uint32_t mbits[4][MW];
  uint32_t mwords[2][NVI][2];  // the two mask words each tile's bits straddle, fetched with the tile (items s, s-1)
  const char *ninf = (const char *)g_neg_inf_page[DT];

  // one load of item `it` (tile k) into buffer B
  typedef const __attribute__((address_space(1))) char *gptr_t;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto load_tile = [&](const Item &it, uint4 (&b)[NVI], uint32_t (&mw)[NVI][2], int k) {
    const int v = it.vlo + v_item + k * 64;
    const gptr_t src = v < it.vhi ? (gptr_t)it.base + (int64_t)v * 16 : (gptr_t)ninf + lane * 16;
    const u32x4 t = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(src);
    b[k] = make_uint4(t.x, t.y, t.z, t.w);
    if constexpr (kBits) {
      // words holding bits j0 .. j0+EPV-1 (clamped; bits of out-of-row elements are don't-care)
      const int j0 = v * EPV - it.a;
      const int jb = j0 < 0 ? 0 : j0;
      const int n_words = (V + 31) >> 5;
      const int wi = jb >> 5;
      const int w0 = wi < n_words ? wi : n_words - 1, w1 = wi + 1 < n_words ? wi + 1 : n_words - 1;
      const __attribute__((address_space(1))) uint32_t *mr = (const __attribute__((address_space(1))) uint32_t *)it.mrow;
      mw[k][0] = mr[w0];
      mw[k][1] = mr[w1];
    }
  };

  // edge patching (first / last vector of the ROW) + mask bits of the lane's NVI tiles
  auto prepare = [&](const Item &it, uint4 (&b)[NVI], const uint32_t (&mw)[NVI][2], uint32_t (&mb)[MW]) {
    auto patch = [&](uint4 &rk, int first_valid, int n_valid) {
      uint32_t w[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        if (c < first_valid || c >= n_valid) {
          if constexpr (DT == kDtF32) w[c] = 0xff800000u;
          else if constexpr (DT == kDtBf16)
            w[c >> 1] = (c & 1) ? ((w[c >> 1] & 0x0000ffffu) | 0xff800000u) : ((w[c >> 1] & 0xffff0000u) | 0x0000ff80u);
          else
            w[c >> 1] = (c & 1) ? ((w[c >> 1] & 0x0000ffffu) | 0xfc000000u) : ((w[c >> 1] & 0xffff0000u) | 0x0000fc00u);
        }
      }
      rk = make_uint4(w[0], w[1], w[2], w[3]);
    };
    const int vlast = it.nv - 1;
#pragma unroll
    for (int k = 0; k < NVI; ++k) {
      const int v = it.vlo + v_item + k * 64;
      if (v == 0 && it.a > 0) patch(b[k], it.a, vlast == 0 ? V + it.a : EPV);
      else if (v == vlast && v < it.vhi) patch(b[k], 0, V + it.a - vlast * EPV);
    }
#pragma unroll
    for (int i = 0; i < MW; ++i) mb[i] = kBits ? 0u : 0xffffffffu;
    if constexpr (kBits) {
#pragma unroll
      for (int k = 0; k < NVI; ++k) {
        const int j0 = (it.vlo + v_item + k * 64) * EPV - it.a;
        const int jb = j0 < 0 ? 0 : j0;
        const uint32_t f = __builtin_amdgcn_alignbit(mw[k][1], mw[k][0], (uint32_t)(jb & 31));
        mb[(k * EPV) >> 5] |= ((f << (jb - j0)) & ((1u << EPV) - 1u)) << ((k * EPV) & 31);
      }
    }
  };

  auto local_max = [&](const uint4 (&b)[NVI], const uint32_t (&mb)[MW], float &m_all, float &m_msk) {
    m_all = kNegInf;
    m_msk = kNegInf;
#pragma unroll
    for (int k = 0; k < NVI; ++k) {
      float xk[EPV];
      unpack_vec<DT>(b[k], xk);
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        const float xv = SCALED ? xk[c] * p.scale : xk[c];
        m_all = fmaxf(m_all, xv);
        if constexpr (kBits) {
          const uint32_t fill = bit_fill(mb[(k * EPV + c) >> 5], (k * EPV + c) & 31);
          m_msk = fmaxf(m_msk, __uint_as_float((__float_as_uint(xv) & fill) | (0xff800000u & ~fill)));
        }
      }
    }
    m_all = wave_max(m_all);
    m_msk = kBits ? wave_max(m_msk) : m_all;
  };

  // granule address of (particle, member m, which)
  auto granule = [&](int pidx, int m, int which) {
    return (gu64 *)(q.xch + ((int64_t)pidx * kClusterSize + m) * 2 + which);
  };

  // ---- software pipeline over steps ---------------------------------------------------------------------
  Item it_load, it_max, it_wait, it_sum;  // items of stage load / max / granules in flight / sums
  Item it_pre = item_of(0);               // descriptor of the next item to load, fetched a step early
  unsigned long long gran = 0;            // this lane's granule of it_wait (lanes 0..7 of every wave)

  auto step = [&](auto S_tag, int s) {
    constexpr int S = decltype(S_tag)::value;  // s & 3
    uint4(&b_load)[NVI] = buf[S];
    uint4(&b_max)[NVI] = buf[(S + 3) & 3];
    uint4(&b_sum)[NVI] = buf[(S + 1) & 3];
    const bool do_load = s < R, do_max = s >= 1 && s - 1 < R, do_wait = s >= 2 && s - 2 < R,
               do_sum = s >= 3 && s - 3 < R;
    if (do_load) it_load = it_pre;
    if (s + 1 < R) it_pre = item_of(s + 1);  // its scalar loads overlap this step

    // (a) cluster-wide exponents of the item whose sums are due: granules were requested last step
    float N_all = 0.f, N_msk = 0.f;
    if (do_sum) {
      unsigned spins = 0;
      const int m = lane & 3, which = (lane >> 2) & 1;
      bool ok = (lane >= 8) || (gran >> 32) == 1ull;
      while (!__all(ok)) {
        if (lane < 8) {
          gran = __hip_atomic_load(granule(it_sum.pidx, m, which), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (gran >> 32) == 1ull;
        }
        if (++spins > kSpinLimit) {  // never hang: flag and fall through with what we have
          if (lane == 0) atomicExch(q.timeout, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      // lanes 0-3: members' row maxima, lanes 4-7: masked maxima
      const uint32_t gv = (uint32_t)gran;
      float g_all = kNegInf, g_msk = kNegInf;
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2) {
        g_all = fmaxf(g_all, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)gv, m2)));
        g_msk = fmaxf(g_msk, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)gv, 4 + m2)));
      }
      N_all = __builtin_rintf(g_all * kLog2e);
      N_msk = __builtin_rintf(g_msk * kLog2e);
      if (member == 0 && tid == 0) {
        q.row_exps[2 * it_sum.pidx] = N_all;
        q.row_exps[2 * it_sum.pidx + 1] = N_msk;
      }
    }

    // (b) sums of item s-3, one load of item s issued per tile
    if (do_sum) {
      const float Nb_all = N_all + (float)kFixShift, Nb_msk = N_msk + (float)kFixShift;
      const bool same_n = N_all == N_msk;
      uint64_t acc = 0, accm = 0;
      const uint32_t(&mb)[MW] = mbits[(S + 1) & 3];
#pragma unroll
      for (int k = 0; k < NVI; ++k) {
        float xk[EPV];
        unpack_vec<DT>(b_sum[k], xk);
        if constexpr (SCALED) {
#pragma unroll
          for (int c = 0; c < EPV; ++c) xk[c] = xk[c] * p.scale;
        }
#pragma unroll
        for (int h = 0; h < EPV / 4; ++h) {
          uint32_t pf[4], sh[4];
          exp_fix4(xk[4 * h], xk[4 * h + 1], xk[4 * h + 2], xk[4 * h + 3], Nb_all, pf, sh);
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const uint64_t qv = ((uint64_t)pf[c4] << 32) >> sh[c4];
            acc += qv;
            if constexpr (kBits)
              accm += mask_u64(qv, bit_fill(mb[(k * EPV + 4 * h + c4) >> 5], (k * EPV + 4 * h + c4) & 31));
          }
        }
        opaque(acc);
        opaque(accm);
        if (do_load) load_tile(it_load, b_load, mwords[S & 1], k);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (kBits) {
        if (!same_n) {  // allowed set tops out in a lower binade: redo the masked sum against N_msk
          accm = 0;
#pragma unroll
          for (int k = 0; k < NVI; ++k) {
            float xk[EPV];
            unpack_vec<DT>(b_sum[k], xk);
#pragma unroll
            for (int c = 0; c < EPV; ++c) {
              const float xv = SCALED ? xk[c] * p.scale : xk[c];
              accm += mask_u64(fix_term(xv, Nb_msk), bit_fill(mb[(k * EPV + c) >> 5], (k * EPV + c) & 31));
            }
            opaque(accm);
          }
        }
      } else {
        accm = acc;
      }
      const uint64_t t_all = wave_scan_u64(acc);
      const uint64_t t_msk = kBits ? wave_scan_u64(accm) : t_all;
      if (lane == 63) {
        uint64_t *o = q.wave_sums + (((int64_t)it_sum.pidx * kClusterSize + member) * W + wave) * 2;
        o[0] = t_all;
        o[1] = t_msk;
      }
    } else if (do_load) {
#pragma unroll
      for (int k = 0; k < NVI; ++k) load_tile(it_load, b_load, mwords[S & 1], k);
    }

    // (c) local maxima of item s-1 -> granules
    if (do_max) {
      prepare(it_max, b_max, mwords[(S + 1) & 1], mbits[(S + 3) & 3]);
      float m_all, m_msk;
      local_max(b_max, mbits[(S + 3) & 3], m_all, m_msk);
      if (lane == 0) {
        s_wmax[s & 1][0][wave] = m_all;
        s_wmax[s & 1][1][wave] = m_msk;
      }
    }
    __syncthreads();
    if (do_max && tid < 2) {  // thread 0 publishes the row maximum, thread 1 the masked maximum
      float m = s_wmax[s & 1][tid][0];
#pragma unroll
      for (int w = 1; w < W; ++w) m = fmaxf(m, s_wmax[s & 1][tid][w]);
      __hip_atomic_store(granule(it_max.pidx, member, tid), (1ull << 32) | (unsigned long long)__float_as_uint(m),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // (d) request the granules of item s-2 (published one step ago by every member); consumed next step
    // rotate the stage items: next step sums item s-2, waits on s-1, takes maxima of s
    it_sum = it_wait;
    it_wait = it_max;
    it_max = it_load;
    (void)do_wait;
    gran = 0;
    if (s >= 2 && s - 2 < R && lane < 8)
      gran = __hip_atomic_load(granule(it_sum.pidx, lane & 3, (lane >> 2) & 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
  };

  for (int s = 0; s < R + 3; s += 4) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < R + 3) step(std::integral_constant<int, 1>{}, s + 1);
    if (s + 2 < R + 3) step(std::integral_constant<int, 2>{}, s + 2);
    if (s + 3 < R + 3) step(std::integral_constant<int, 3>{}, s + 3);
  }
}


// ---- second kernels of the v3 path ----------------------------------------------------------------------
// lse / logZ from the 32 per-wave partial sums of a particle; also re-zeroes its granules for the next launch
__device__ __forceinline__ void finish_row_v3(const V3Params &q, int pidx, uint64_t S_all, uint64_t S_msk) {
  const RowParams &p = q.rp;
  const float N_all = q.row_exps[2 * pidx], N_msk = q.row_exps[2 * pidx + 1];
  const double lse_all = S_all ? log_fix(S_all, (int32_t)N_all - kFixFrac) : (double)kNegInf;
  const double lse_msk = S_msk ? log_fix(S_msk, (int32_t)N_msk - kFixFrac) : (double)kNegInf;
  if (p.out_lse) p.out_lse[pidx] = (float)lse_all;
  if (p.out_logZ) p.out_logZ[pidx] = (float)(lse_msk - lse_all);
}

// one wave per particle: totals, lse/logZ, and (DRAW) the Philox inverse-CDF walk inside one chunk
template <int DT, int MASK, bool DRAW>
__global__ __launch_bounds__(256) void locate_kernel_v3(const V3Params q, int nvi) {
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  const RowParams &p = q.rp;
  const int lane = threadIdx.x & 63;
  const int pidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pidx >= p.n_particles) return;
  // lane c < 32 stands for chunk c = member * 8 + wave (vocabulary order)
  const uint64_t *ws = q.wave_sums + (int64_t)pidx * (kClusterSize * 8 * 2);
  const uint64_t c_all = lane < 32 ? ws[lane * 2] : 0ull, cs = lane < 32 ? ws[lane * 2 + 1] : 0ull;
  const uint64_t S_all = wave_sum_u64(c_all);
  const uint64_t incl_c = wave_scan_u64(cs);
  const uint64_t S = readlane_u64(incl_c, 63);
  if (lane == 0) finish_row_v3(q, pidx, S_all, S);
  if (lane < 8) q.xch[(int64_t)pidx * 8 + lane] = 0ull;  // granules back to "not published"
  if constexpr (!DRAW) return;
  if (!p.out_token) return;
  if (S == 0) {
    if (lane == 0) p.out_token[pidx] = -1;
    return;
  }
  const int V = p.V;
  const int row = p.row_of ? p.row_of[pidx] : pidx;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int a = (int)(((uintptr_t)rowp) & 15) / ES;
  const char *base = rowp - a * ES;
  const int nv = (V + a + EPV - 1) / EPV;
  const int Q = (nv + kClusterSize - 1) / kClusterSize;
  const uint32_t *mrow = nullptr;
  if constexpr (MASK == kMaskBits) {
    const int mi = p.mask_id ? p.mask_id[pidx] : (p.n_masks == 1 ? 0 : pidx);
    mrow = (const uint32_t *)p.mask + (int64_t)mi * p.mask_ld;
  }
  const uint64_t gp = (uint64_t)(p.particle_base + pidx);
  const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
  const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
  uint32_t rnd[4];
  philox4x32_10(ctr, key, rnd);
  uint64_t Tc = __umul64hi(((uint64_t)rnd[1] << 32) | rnd[0], S);
  {
    uint32_t z = 0;
    opaque(z);
    Tc += z;
  }
  const int csel = __ffsll((long long)__ballot(incl_c > Tc)) - 1;
  Tc -= readlane_u64(incl_c - cs, csel);
  const float Nb = q.row_exps[2 * pidx + 1] + (float)kFixShift;
  const int mem = csel >> 3, wv = csel & 7;
  const int vhi = (mem + 1) * Q < nv ? (mem + 1) * Q : nv;
  const int vstart = mem * Q + wv * 64 * nvi;
  uint64_t run = 0, asel = 0;
  uint4 rsel = make_uint4(0, 0, 0, 0);
  uint32_t nsel = 0;
  int j0sel = 0;
  bool found = false;
  for (int t = 0; t < nvi; ++t) {
    const int v = vstart + t * 64 + lane;
    const bool inside = v < vhi;
    const uint4 rk = *reinterpret_cast<const uint4 *>(base + (int64_t)(inside ? v : vhi - 1) * 16);
    const int j0 = v * EPV - a;
    uint32_t nib = (1u << EPV) - 1u;
    if constexpr (MASK == kMaskBits) nib = mask_nibble<EPV>(mrow, (V + 31) >> 5, j0);
    if (!inside) nib = 0;
    float xs[EPV];
    unpack_vec<DT>(rk, xs);
    uint64_t aj = 0;
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      const float xv = p.use_scale ? xs[c] * p.scale : xs[c];
      const bool ok = ((uint32_t)(j0 + c) < (uint32_t)V) && ((nib >> c) & 1u);
      aj += ok ? fix_term(xv, Nb) : 0ull;
    }
    const uint64_t cj = wave_sum_u64(aj);
    if (!found && Tc < run + cj) {
      found = true;
      Tc -= run;
      asel = aj;
      rsel = rk;
      nsel = nib;
      j0sel = j0;
    }
    run += cj;
  }
  const uint64_t incl = wave_scan_u64(asel);
  const int lsel = __ffsll((long long)__ballot(incl > Tc)) - 1;
  if (lane == lsel) {
    uint64_t Tl = Tc - (incl - asel);
    float xs[EPV];
    unpack_vec<DT>(rsel, xs);
    int32_t tok = -1;
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      const float xv = p.use_scale ? xs[c] * p.scale : xs[c];
      const bool ok = ((uint32_t)(j0sel + c) < (uint32_t)V) && ((nsel >> c) & 1u);
      const uint64_t qv = ok ? fix_term(xv, Nb) : 0ull;
      if (tok < 0) {
        if (Tl < qv) tok = j0sel + c;
        else Tl -= qv;
      }
    }
    p.out_token[pidx] = tok;
  }
}

}  // namespace glb
