// glb_row_kernel_v4.hpp — persistent fused particle-step kernel with IN-PLACE register prefetch.
//
// One 512-thread workgroup per CU streams rows through its registers (a gpt2-sized fp32 row is 25 16-byte
// vectors per lane).  The row is needed twice - maximum, then fixed-point sums - so it has to sit still
// while it is reduced; but the moment tile k of the sums pass has been consumed its registers are dead, and
// the load of tile k of the NEXT row is issued straight into them.  No staging buffer, no copy: the loads
// stay in flight through the cross-wave reductions and the next row's maximum pass (which consumes the
// tiles in issue order behind the compiler's counted `s_waitcnt vmcnt(N)`), so the CU's memory queue only
// runs dry while the very first row arrives and while the very last one is reduced.
//
// Differences from glb_row_kernel_v2.hpp (LDS-DMA staging): no LDS for the row (LDS holds two mask bit rows
// and a little scratch), no `land` copy, and the rare own-scale pass of the masked sum cannot run in the
// loop (the row's registers are already being refilled): the loop only takes the sums on the row's scale;
// rows whose masked sum came out below 2^37 are redone from memory by the whole workgroup after the loop
// (fix_row), then every row gets its lse / logZ / token (finish_row / locate_row), one wave per row.
//
// Arithmetic is GLB math exactly as everywhere else: bit-identical to the oracle.
#pragma once
#include "glb_row_kernel_v2.hpp"

namespace glb {

typedef const __attribute__((address_space(1))) char *gptr_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 gload16(gptr_t src) {
  const u32x4_t t = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_t *>(src);
  return make_uint4(t.x, t.y, t.z, t.w);
}

// Own-scale masked sum of one row, recomputed from memory by the whole workgroup (rare: allowed mass below
// 2^-7 of the row).  Same chunk layout as the streaming loop: wave w, chunk g = tiles [g*GS, (g+1)*GS).
template <int DT, int NVL, int T, bool SCALED>
__device__ __forceinline__ void fix_row(const RowParams &p, int pidx, uint32_t scr_max, uint32_t scr_sum) {
  constexpr int W = T / 64;
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr int GS = (NVL % 5 == 0) ? 5 : 4;
  constexpr int NG = NVL / GS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  const int row = p.row_of ? p.row_of[pidx] : pidx;
  const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
  const int a = (int)(((uintptr_t)rowp) & 15) / ES;
  const char *base = rowp - a * ES;
  const int nv = (V + a + EPV - 1) / EPV;
  const int mi = p.mask_id ? p.mask_id[pidx] : (p.n_masks == 1 ? 0 : pidx);
  const uint32_t *mrow = (const uint32_t *)p.mask + (int64_t)mi * p.mask_ld;
  const int n_words = (V + 31) >> 5;
  const int v0 = wave * (64 * NVL) + lane;
  const int wl = lane & 15, wr = wl < W ? wl : W - 1;

  float mk = kNegInf;
#pragma unroll 1
  for (int k = 0; k < NVL; ++k) {
    const int v = v0 + k * 64;
    if (v < nv) {
      const uint4 rk = *reinterpret_cast<const uint4 *>(base + (int64_t)v * 16);
      const int j0 = v * EPV - a;
      const uint32_t nib = mask_nibble<EPV>(mrow, n_words, j0);
      float xs[EPV];
      unpack_vec<DT>(rk, xs);
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        const float xv = SCALED ? xs[c] * p.scale : xs[c];
        const bool ok = ((uint32_t)(j0 + c) < (uint32_t)V) && ((nib >> c) & 1u);
        mk = fmaxf(mk, ok ? xv : kNegInf);
      }
    }
  }
  mk = wave_max(mk);
  if (lane == 0) lds_write_b32(scr_max + wave * 4, __float_as_uint(mk));
  lds_barrier();
  mk = row16_max_bcast(__uint_as_float(lds_read_b32_wait(scr_max + wr * 4)));
  if (!(mk > kNegInf)) {  // nothing allowed (workgroup-uniform): the sums on the row's scale are 0 already
    lds_barrier();
    return;
  }
  const float N_k = __builtin_rintf(mk * kLog2e);
  const float Nb = N_k + (float)kFixShift;
  uint64_t s_w = 0;
#pragma unroll 1
  for (int g = 0; g < NG; ++g) {
    uint64_t ag = 0;
#pragma unroll 1
    for (int t = 0; t < GS; ++t) {
      const int v = v0 + (g * GS + t) * 64;
      if (v < nv) {
        const uint4 rk = *reinterpret_cast<const uint4 *>(base + (int64_t)v * 16);
        const int j0 = v * EPV - a;
        const uint32_t nib = mask_nibble<EPV>(mrow, n_words, j0);
        float xs[EPV];
        unpack_vec<DT>(rk, xs);
#pragma unroll
        for (int c = 0; c < EPV; ++c) {
          const float xv = SCALED ? xs[c] * p.scale : xs[c];
          const bool ok = ((uint32_t)(j0 + c) < (uint32_t)V) && ((nib >> c) & 1u);
          ag += ok ? fix_term(xv, Nb) : 0ull;
        }
      }
    }
    const uint64_t tg = wave_sum_u64(ag);
    if (lane == 0) p.chunk_sums[(int64_t)pidx * (W * NG) + wave * NG + g] = tg;
    s_w += tg;
  }
  if (lane == 0) lds_write_b64(scr_sum + wave * 8, s_w);
  lds_barrier();
  const uint64_t cw = wl < W ? lds_read_b64_wait(scr_sum + wr * 8) : 0ull;
  const uint64_t S = readlane_u64(row16_scan_u64(cw), 15);
  if (tid == 0) {
    p.row_sums[2 * pidx + 1] = S;
    p.row_exps[2 * pidx + 1] = N_k;
  }
  lds_barrier();  // scratch is reused by the next row's fix
}

template <int DT, int MASK, int MODE, int NVL, int T, bool SCALED>
__global__ __launch_bounds__(T) void row_kernel_v4(const RowParams p) {
  constexpr int W = T / 64;
  static_assert(W <= 16, "cross-wave scratch is reduced inside one DPP row of 16 lanes");
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr int MBW = (NVL * EPV + 31) / 32;
  constexpr bool kPhilox = MODE == kModePhilox;
  constexpr bool kBits = MASK == kMaskBits;
  static_assert(MASK != kMaskF32 && MODE != kModeNoise, "v4 covers mask none/bits, stats/philox");
  constexpr int GS = (NVL % 5 == 0) ? 5 : 4;  // tiles per chunk of the draw's search
  constexpr int NG = NVL / GS;                // chunks per wave
  static_assert(NVL % GS == 0, "chunks must tile a wave's vectors exactly (locate_row indexes them linearly)");
  // LDS: [pad | mask row A | mask row B | scratch]; a mask row covers NVL*T*EPV bits plus alignment slack
  constexpr int MROW_V = kBits ? (NVL * T * EPV / 8 + 15) / 16 + 2 : 0;  // uint4 per buffer
  constexpr int MPT = kBits ? (MROW_V + T - 1) / T : 1;                  // mask vectors staged per thread
  constexpr int SCR_V = 24;                                              // 2x16 floats, 2x16 u64
  __shared__ uint4 s_lds[1 + 2 * MROW_V + SCR_V];
  const uint32_t scr = lds_addr(s_lds + 1 + 2 * MROW_V);
  const uint32_t scr_max = scr, scr_sum = scr + 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  const int n = p.n_particles, G = gridDim.x;
  const int v0 = wave * (64 * NVL) + lane;
  const int wl = lane & 15;            // lane wl of every DPP row stands for wave wl
  const int wr = wl < W ? wl : W - 1;  // rows of 16 lanes but only W waves: the rest duplicate the last

  auto particle_of = [&](int vb) {  // XCD-aware particle order (same as v1)
    const int q = n >> 3, r = n & 7, xcd = vb & 7, i = vb >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  };
  struct RowRef {
    gptr_t base;  // 16-byte aligned-down row start
    int a, nv;    // leading pad elements, vectors covering the row
    gptr_t mrow16;  // 16-byte aligned-down mask row
    int am, mvec;   // leading pad words, 16-byte vectors covering the mask row
    int pidx;
  };
  auto row_ref = [&](int vb) {
    RowRef r;
    r.pidx = particle_of(vb);
    const int row = p.row_of ? p.row_of[r.pidx] : r.pidx;
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    r.a = (int)(((uintptr_t)rowp) & 15) / ES;
    r.base = (gptr_t)(rowp - r.a * ES);
    r.nv = (V + r.a + EPV - 1) / EPV;
    r.mrow16 = nullptr;
    r.am = 0;
    r.mvec = 1;
    if constexpr (kBits) {
      const int mi = p.mask_id ? p.mask_id[r.pidx] : (p.n_masks == 1 ? 0 : r.pidx);
      const char *mp = (const char *)p.mask + (int64_t)mi * p.mask_ld * 4;
      r.am = (int)(((uintptr_t)mp) & 15) / 4;
      r.mrow16 = (gptr_t)(mp - r.am * 4);
      r.mvec = (((V + 31) >> 5) + r.am + 3) >> 2;
    }
    return r;
  };
  const gptr_t ninf = (gptr_t)(const char *)g_neg_inf_page[DT];
  uint32_t lane_off = (uint32_t)v0 * 16u;  // re-opaqued per row so the NVL tile addresses are not hoisted
  auto load_tile = [&](const RowRef &r, int k) -> uint4 {
    const uint32_t off = lane_off + (uint32_t)k * 1024u;
    return gload16(off < (uint32_t)r.nv * 16u ? r.base + off : ninf + lane * 16);  // past the row: -inf page
  };
  auto load_mask = [&](const RowRef &r, uint4 (&mreg)[MPT]) {
    if constexpr (kBits) {
#pragma unroll
      for (int j = 0; j < MPT; ++j) {
        int mv = tid + j * T;
        mv = mv < r.mvec ? mv : r.mvec - 1;
        mreg[j] = gload16(r.mrow16 + (int64_t)mv * 16);
      }
    }
  };
  auto stage_mask = [&](const uint4 (&mreg)[MPT], int buf) {
    if constexpr (kBits) {
#pragma unroll
      for (int j = 0; j < MPT; ++j)
        if (tid + j * T < MROW_V) s_lds[1 + buf * MROW_V + tid + j * T] = mreg[j];
    }
  };
  // this lane's mask bits from the staged bit row: tile k's EPV bits start at bit j00 + 64*EPV*k (a fixed shift
  // and a word index that advances by 2*EPV per tile); words outside the staged row only feed -inf elements
  auto build_bits = [&](const RowRef &r, int buf, uint32_t (&mb)[MBW]) {
#pragma unroll
    for (int i = 0; i < MBW; ++i) mb[i] = kBits ? 0u : 0xffffffffu;
    if constexpr (kBits) {
      const int j00 = v0 * EPV - r.a;
      const uint32_t shb = (uint32_t)(j00 & 31);
      const uint32_t *mw = reinterpret_cast<const uint32_t *>(s_lds + 1 + buf * MROW_V) + ((j00 >> 5) + r.am);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const uint32_t f = __builtin_amdgcn_alignbit(mw[k * 2 * EPV + 1], mw[k * 2 * EPV], shb);
        mb[(k * EPV) >> 5] |= (f & ((1u << EPV) - 1u)) << ((k * EPV) & 31);
      }
    }
  };
  // only two vectors of a row can be partly outside it: the first (leading pad) and the last
  auto patch = [&](uint4 &rk, int first_valid, int n_valid) {  // keep elements [first_valid, n_valid)
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

  uint4 raw[NVL];
  uint4 mreg[MPT];
  uint32_t mb[MBW];

  int vb = blockIdx.x;
  if (vb >= n) return;
#ifdef GLB_STAMPS
  int stamp_i = 0;
  if (tid == 0 && blockIdx.x < 256) g_realtime[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
  GLB_STAMP();
  RowRef cur = row_ref(vb);
  int buf = 0;
  load_mask(cur, mreg);
#pragma unroll
  for (int k = 0; k < NVL; ++k) raw[k] = load_tile(cur, k);
  stage_mask(mreg, 0);
  if constexpr (kBits) lds_barrier();
  build_bits(cur, 0, mb);
  GLB_STAMP();

  for (;;) {
    const int vb_next = vb + G;
    const bool has_next = vb_next < n;
    RowRef nxr = cur;
    if (has_next) {
      nxr = row_ref(vb_next);
      load_mask(nxr, mreg);  // lands long before the row tiles issued behind it
    }
    opaque(lane_off);

    // ---- phase 1: row maximum (tiles are consumed in the order their loads were issued) ---------------
    float m_all = kNegInf;
    {
      const int vlast = cur.nv - 1;                        // last vector of the row
      const int klast = (vlast - wave * (64 * NVL)) >> 6;  // its tile in this wave (if it is this wave's)
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        if (k == 0 && tid == 0 && cur.a > 0) patch(raw[0], cur.a, EPV);
        if (k == klast && v0 + k * 64 == vlast) patch(raw[k], (vlast == 0) ? cur.a : 0, V + cur.a - vlast * EPV);
        float xk[EPV];
        unpack_vec<DT>(raw[k], xk);
#pragma unroll
        for (int c = 0; c < EPV; ++c) m_all = fmaxf(m_all, SCALED ? xk[c] * p.scale : xk[c]);
      }
    }
    m_all = wave_max(m_all);
    if (lane == 0) lds_write_b32(scr_max + wave * 4, __float_as_uint(m_all));
    lds_barrier();
    m_all = row16_max_bcast(__uint_as_float(lds_read_b32_wait(scr_max + wr * 4)));
    const float N_all = __builtin_rintf(m_all * kLog2e);
    const float Nb_all = N_all + (float)kFixShift;
    GLB_STAMP();

    // ---- phase 2: both fixed-point sums on the row's scale; tile k of the next row is requested as soon
    //      as tile k of this one has been consumed --------------------------------------------------------
    uint64_t acc = 0, s_msk = 0;
    uint64_t ag[kPhilox ? NG : 1];  // per-lane masked sums of the chunks (GS tiles each)
    if constexpr (kPhilox) {
#pragma unroll
      for (int g = 0; g < NG; ++g) ag[g] = 0;
    }
#pragma unroll
    for (int i = 0; i < MBW; ++i) opaque(mb[i]);
#pragma unroll
    for (int k = 0; k < NVL; ++k) opaque(raw[k]);
#pragma unroll
    for (int k = 0; k < NVL; ++k) {
      uint64_t ak = 0;
      float xk[EPV];
      unpack_vec<DT>(raw[k], xk);
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
          const int c = 4 * h + c4;
          const uint64_t q = ((uint64_t)pf[c4] << 32) >> sh[c4];
          acc += q;
          if constexpr (kBits) {
            const uint32_t fill = bit_fill(mb[(k * EPV + c) >> 5], (k * EPV + c) & 31);
            ak += mask_u64(q, fill);
          } else if constexpr (kPhilox) {
            ak += q;
          }
        }
      }
      s_msk += ak;
      if constexpr (kPhilox) {
        ag[k / GS] += ak;
        if (k % GS == GS - 1 || k == NVL - 1) opaque(ag[k / GS]);
      }
      // pin both running sums here: otherwise the masked adds are reassociated and sunk below the loop,
      // which keeps every q of the row alive
      opaque(s_msk);
      opaque(acc);
      if (has_next) raw[k] = load_tile(nxr, k);
      __builtin_amdgcn_sched_barrier(0);  // one vector at a time (register pressure, load placement)
    }
    if constexpr (!kBits) s_msk = acc;
    GLB_STAMP();
    {
      const uint64_t t_all = wave_scan_u64(acc);
      uint64_t t_msk = t_all;
      if constexpr (kBits) t_msk = wave_scan_u64(s_msk);
      if (lane == 63) {
        lds_write_b64(scr_sum + wave * 8, t_all);
        lds_write_b64(scr_sum + 128 + wave * 8, t_msk);
      }
      if constexpr (kPhilox) {
        // wave totals of every chunk -> workspace; locate_row finishes the draw from them
        uint64_t *crow = p.chunk_sums + (int64_t)cur.pidx * (W * NG) + wave * NG;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const uint64_t tg = wave_scan_u64(ag[g]);
          if (lane == 63) crow[g] = tg;
        }
      }
    }
    if (has_next) stage_mask(mreg, buf ^ 1);  // the other buffer: nobody reads it until after this barrier
    lds_barrier();
    {
      const uint64_t cw_all = wl < W ? lds_read_b64_wait(scr_sum + wr * 8) : 0ull;
      const uint64_t cw_msk = wl < W ? lds_read_b64_wait(scr_sum + 128 + wr * 8) : 0ull;
      const uint64_t in_all = row16_scan_u64(cw_all), in_msk = row16_scan_u64(cw_msk);
      const uint64_t S_all = readlane_u64(in_all, 15), S_msk = readlane_u64(in_msk, 15);
      if (tid == 0) {
        p.row_sums[2 * cur.pidx] = S_all;
        p.row_sums[2 * cur.pidx + 1] = S_msk;  // below 2^37: redone on its own scale in the tail
        p.row_exps[2 * cur.pidx] = N_all;
        p.row_exps[2 * cur.pidx + 1] = N_all;
      }
    }
    GLB_STAMP();
    if (!has_next) break;
    buf ^= 1;
    cur = nxr;
    vb = vb_next;
    build_bits(cur, buf, mb);
    GLB_STAMP();
  }

  // ---- tail ------------------------------------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's workspace stores have reached L2
  lds_barrier();
  if constexpr (kBits) {
    // rows whose allowed tokens hold < 2^-7 of the mass: masked sum again, on the masked maximum's own scale
#pragma unroll 1
    for (int vr = blockIdx.x; vr < n; vr += G) {
      const int pidx = particle_of(vr);
      uint32_t top = (uint32_t)(ld_agent(p.row_sums + 2 * pidx + 1) >> 37);  // sums stay below 2^62
      opaque(top);  // VALU compare (uniform u64 `<` miscompile, see v1)
      if (top == 0u) fix_row<DT, NVL, T, SCALED>(p, pidx, scr_max, scr_sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  for (int vr = blockIdx.x + wave * G; vr < n; vr += W * G) {
    const int pidx = particle_of(vr);
    if constexpr (kPhilox) {
      if (p.out_token) locate_row<DT, MASK>(p, pidx, lane);
      else if (lane == 0) finish_row(p, pidx);
    } else {
      if (lane == 0) finish_row(p, pidx);
    }
  }
  GLB_STAMP();
#ifdef GLB_STAMPS
  if (tid == 0 && blockIdx.x < 256) g_realtime[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace glb
