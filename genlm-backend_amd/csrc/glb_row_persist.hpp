// glb_row_persist.hpp — persistent fused particle-step kernel with IN-PLACE register prefetch.
//
// One 512-thread workgroup per CU streams rows through its registers (a gpt2-sized fp32 row is 25 16-byte
// vectors per lane).  A row is needed twice - maximum, then fixed-point sums - so it has to sit still while
// it is reduced; but the moment tile k of the sums pass has been consumed its registers are dead, and the
// load of tile k of the NEXT row is issued straight into them.  No staging buffer, no copy: the loads stay
// in flight through the cross-wave reductions and the next row's maximum pass (which consumes the tiles in
// issue order behind the compiler's counted `s_waitcnt vmcnt(N)`), so the CU's memory queue only runs dry
// while the very first row arrives and while the very last one is reduced.
//
// What that needs on gfx950 / ROCm 7.2: no `__syncthreads()` inside the loop (its fence drains vmcnt, i.e.
// every prefetched tile): barriers are raw `s_barrier` behind an explicit `s_waitcnt lgkmcnt(0)`, and the
// cross-wave scratch is accessed with inline-asm `ds_*`.
//
// The loop only takes the sums on the row's scale (GLB math: one exponential serves both sums).  Rows whose
// masked sum came out below 2^37 - allowed mass below 2^-7 of the row, rare - are redone from memory on
// the masked maximum's own scale by the whole workgroup after the loop (fix_row; the row's registers were
// already being refilled when the sum became known).  Then every row gets its lse / logZ (finish_row) and
// its token (locate): from per-chunk wave totals the loop left in the workspace, a wave - or a pair of
// waves, when the workgroup streamed at most four rows - picks the chunk, re-reads that chunk's 4-5 KiB of
// the row (L2 / Infinity Cache), recomputes the tile sums against the stored exponent and walks
// tile -> lane -> element in vocabulary order.  Same integers as the one-workgroup-per-particle kernel.
//
// (An earlier version staged the next row in LDS by LDS-DMA, `global_load_lds_dwordx4`, and copied it to
//  registers at the row switch: 24.5k cycles per row against 22.3k here, with 90 more VGPRs; see DESIGN.md.)
#pragma once
#include "glb_row_kernel.hpp"

namespace glb {

typedef const __attribute__((address_space(1))) char *gptr_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// 1 KiB of -inf per element type (f32 | bf16 | f16 bit patterns): vectors past the end of a row are fetched
// from here instead of being patched after the load
struct NegInfPage {
  uint32_t w[3][256];
  constexpr NegInfPage() : w() {
    for (int i = 0; i < 256; ++i) {
      w[0][i] = 0xff800000u;
      w[1][i] = 0xff80ff80u;
      w[2][i] = 0xfc00fc00u;
    }
  }
};
__device__ const NegInfPage g_neg_inf_page{};

__device__ __forceinline__ uint4 gload16(gptr_t src) {  // global_load_dwordx4 (never flat)
  const u32x4_t t = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_t *>(src);
  return make_uint4(t.x, t.y, t.z, t.w);
}

// Uniform read of one input index through the scalar cache.  hipcc will not use s_load here by itself (the
// kernel also stores to global memory, so it cannot prove the array unclobbered) and its vector load of a
// uniform address comes with `s_waitcnt vmcnt(0)`: at the top of a row that drains every prefetched tile.
__device__ __forceinline__ int32_t sload_i32(const int32_t *base, int idx) {
  int32_t v;
  const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane(idx) * 4u;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(off) : "memory");
  return v;
}

// ---- LDS scratch by inline asm + raw barrier (nothing here touches vmcnt) ---------------------------------
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void lds_write_b32(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b64(uint32_t addr, uint64_t v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_read_b32_wait(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint64_t lds_read_b64_wait(uint32_t addr) {
  uint64_t v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// all LDS writes of this wave done, then workgroup barrier
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// max over the first 16 lanes of each row of 16 (DPP), result broadcast from lane 15
__device__ __forceinline__ float row16_max_bcast(float v) {
  v = dpp_max_step<0x111, 0xf>(v);
  v = dpp_max_step<0x112, 0xf>(v);
  v = dpp_max_step<0x114, 0xf>(v);
  v = dpp_max_step<0x118, 0xf>(v);
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 15));
}
// inclusive prefix sum within each row of 16 lanes
__device__ __forceinline__ uint64_t row16_scan_u64(uint64_t v) {
  v += dpp_u64_or0<0x111, 0xf>(v);
  v += dpp_u64_or0<0x112, 0xf>(v);
  v += dpp_u64_or0<0x114, 0xf>(v);
  v += dpp_u64_or0<0x118, 0xf>(v);
  return v;
}

// The workspace words read in the tail were written by other waves of this workgroup with plain stores
// (complete: vmcnt(0) + barrier); agent-scope loads keep the reads out of any stale vector-L1 line.
__device__ __forceinline__ uint64_t ld_agent(const uint64_t *q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float *q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lse / logZ of one particle from its fixed-point sums (double-precision log: one lane)
__device__ __forceinline__ void finish_row(const RowParams &p, int pidx) {
  const uint64_t S_all = ld_agent(p.row_sums + 2 * pidx), S_msk = ld_agent(p.row_sums + 2 * pidx + 1);
  const float N_all = ld_agent(p.row_exps + 2 * pidx), N_msk = ld_agent(p.row_exps + 2 * pidx + 1);
  const double lse_all = S_all ? log_fix(S_all, (int32_t)N_all - kFixFrac) : (double)kNegInf;
  const double lse_msk = S_msk ? log_fix(S_msk, (int32_t)N_msk - kFixFrac) : (double)kNegInf;
  if (p.out_lse) p.out_lse[pidx] = (float)lse_all;
  if (p.out_logZ) p.out_logZ[pidx] = (float)(lse_msk - lse_all);
}

// tuning aid (tools/dbg/stamps.hip): shader-clock stamps of the first and last wave at the phase boundaries
#ifdef GLB_STAMPS
__device__ unsigned long long g_stamps[256 * 2 * 64];
__device__ unsigned long long g_realtime[256 * 2];
#define GLB_STAMP()                                                                                        \
  do {                                                                                                     \
    if (lane == 0 && (wave == 0 || wave == W - 1) && stamp_i < 64 && blockIdx.x < 256)                     \
      g_stamps[(blockIdx.x * 2 + (wave != 0)) * 64 + stamp_i] = __builtin_amdgcn_s_memtime();              \
    ++stamp_i;                                                                                             \
  } while (0)
#else
#define GLB_STAMP() do { } while (0)
#endif

// (Tried and removed: a second register buffer holding the first 8-12 tiles of row r+2, requested right after the
//  sums pass of row r so that the memory queue is also fed during the reductions.  hipcc's counted waits for the
//  in-place tiles do not look past those younger loads - the maximum pass ended up waiting for the look-ahead
//  tiles too - and the kernel got slower: 60.5 / 96.8 us for 8 / 12 tiles against 57.8.)
template <int DT, int MASK, int MODE, int NVL, int T, bool SCALED>
__global__ __launch_bounds__(T) void row_kernel_persist(const RowParams p) {
  constexpr int W = T / 64;
  static_assert(W <= 16 && W % 2 == 0, "cross-wave scratch is reduced inside one DPP row of 16 lanes");
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr int MBW = (NVL * EPV + 31) / 32;
  constexpr bool kPhilox = MODE == kModePhilox;
  constexpr bool kBits = MASK == kMaskBits;
  static_assert(MASK != kMaskF32 && MODE != kModeNoise, "persistent kernel: mask none/bits, stats/philox");
  constexpr int GS = (NVL % 5 == 0) ? 5 : 4;  // tiles per chunk of the draw's search
  constexpr int NG = NVL / GS;                // chunks per wave
  static_assert(NVL % GS == 0, "chunks must tile a wave's vectors exactly (the draw indexes them linearly)");
  // LDS: [pad | mask row A | mask row B | scratch]; a mask row covers NVL*T*EPV bits plus alignment slack
  constexpr int MROW_V = kBits ? (NVL * T * EPV / 8 + 15) / 16 + 2 : 0;  // uint4 per buffer
  constexpr int MPT = kBits ? (MROW_V + T - 1) / T : 1;                  // mask vectors staged per thread
  constexpr int SCR_V = 24;                                              // 2x16 floats, 2x16 u64
  // the tail's inputs of the first RING rows of this workgroup stay in LDS as well (chunk totals, masked
  // exponent): reading them back from the workspace costs an L2 round trip per dependent step
  constexpr int RING = 16;
  constexpr int RING_V = kPhilox ? (RING * (W * NG + 1) * 8 + 15) / 16 : 0;
  __shared__ uint4 s_lds[1 + 2 * MROW_V + SCR_V + RING_V];
  const uint32_t scr = lds_addr(s_lds + 1 + 2 * MROW_V);
  const uint32_t scr_max = scr, scr_sum = scr + 128;
  const uint32_t ring = scr + SCR_V * 16;            // [RING][W*NG] u64 chunk totals
  const uint32_t ring_n = ring + RING * W * NG * 8;  // [RING] f32 masked exponent (8-byte slots)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  const int n = p.n_particles, G = gridDim.x;
  const int v0 = wave * (64 * NVL) + lane;
  const int wl = lane & 15;            // lane wl of every DPP row stands for wave wl
  const int wr = wl < W ? wl : W - 1;  // rows of 16 lanes but only W waves: the rest duplicate the last

  auto particle_of = [&](int vb) {  // XCD-aware particle order (same as the one-workgroup-per-particle kernel)
    const int q = n >> 3, r = n & 7, xcd = vb & 7, i = vb >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  };
  struct RowRef {
    gptr_t base;    // 16-byte aligned-down row start
    int a, nv;      // leading pad elements, vectors covering the row
    gptr_t mrow16;  // 16-byte aligned-down mask row
    int am, mvec;   // leading pad words, 16-byte vectors covering the mask row
    int pidx;
  };
  auto row_ref = [&](int vb) {
    RowRef r;
    r.pidx = particle_of(vb);
    const int row = p.row_of ? sload_i32(p.row_of, r.pidx) : r.pidx;
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    r.a = (int)(((uintptr_t)rowp) & 15) / ES;
    r.base = (gptr_t)(rowp - r.a * ES);
    r.nv = (V + r.a + EPV - 1) / EPV;
    r.mrow16 = nullptr;
    r.am = 0;
    r.mvec = 1;
    if constexpr (kBits) {
      const int mi = p.mask_id ? sload_i32(p.mask_id, r.pidx) : (p.n_masks == 1 ? 0 : r.pidx);
      const char *mp = (const char *)p.mask + (int64_t)mi * p.mask_ld * 4;
      r.am = (int)(((uintptr_t)mp) & 15) / 4;
      r.mrow16 = (gptr_t)(mp - r.am * 4);
      r.mvec = (((V + 31) >> 5) + r.am + 3) >> 2;
    }
    return r;
  };
  const gptr_t ninf = (gptr_t)(const char *)g_neg_inf_page.w[DT];
  uint32_t lane_off = (uint32_t)v0 * 16u;  // re-opaqued per row so the NVL tile addresses are not hoisted
  auto load_tile = [&](const RowRef &r, int k) -> uint4 {
    const uint32_t off = lane_off + (uint32_t)k * 1024u;
    return gload16(off < (uint32_t)r.nv * 16u ? r.base + off : ninf + lane * 16);  // past the row: -inf page
  };
  auto load_mask = [&](const RowRef &r, uint4(&mreg)[MPT]) {
    if constexpr (kBits) {
#pragma unroll
      for (int j = 0; j < MPT; ++j) {
        int mv = tid + j * T;
        mv = mv < r.mvec ? mv : r.mvec - 1;
        mreg[j] = gload16(r.mrow16 + (int64_t)mv * 16);
      }
    }
  };
  auto stage_mask = [&](const uint4(&mreg)[MPT], int buf) {
    if constexpr (kBits) {
#pragma unroll
      for (int j = 0; j < MPT; ++j)
        if (tid + j * T < MROW_V) s_lds[1 + buf * MROW_V + tid + j * T] = mreg[j];
    }
  };
  // this lane's mask bits from the staged bit row: tile k's EPV bits start at bit j00 + 64*EPV*k (a fixed shift
  // and a word index that advances by 2*EPV per tile); words outside the staged row only feed -inf elements
  auto build_bits = [&](const RowRef &r, int buf, uint32_t(&mb)[MBW]) {
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
  int ri = 0;             // index of the current row among this workgroup's rows
  uint64_t low_rows = 0;  // bit i: row i (< 64) needs its masked sum redone on its own scale (workgroup-uniform)
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
    const float N_all = exp_n(m_all);
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
          }
        }
      }
      if constexpr (kBits) {
        s_msk += ak;
        if constexpr (kPhilox) ag[k / GS] += ak;
      }
      if constexpr (kPhilox) {
        if (k % GS == GS - 1 || k == NVL - 1) {
          if constexpr (!kBits) {  // unmasked: a chunk's sum is the difference of the running sum at its ends
            ag[k / GS] = acc - s_msk;
            s_msk = acc;
          }
          opaque(ag[k / GS]);
        }
      }
      // pin the running sums here: otherwise the adds are reassociated and sunk below the loop, which keeps
      // every q of the row alive
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
        // wave totals of every chunk -> workspace; the tail finishes the draw from them
        uint64_t *crow = p.chunk_sums + (int64_t)cur.pidx * (W * NG) + wave * NG;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const uint64_t tg = wave_scan_u64(ag[g]);
          if (lane == 63) {
            crow[g] = tg;
            if (ri < RING) lds_write_b64(ring + (uint32_t)((ri * W + wave) * NG + g) * 8u, tg);
          }
        }
      }
    }
    if (has_next) stage_mask(mreg, buf ^ 1);  // the other buffer: nobody reads it until after this barrier
    lds_barrier();
    {
      const uint64_t cw_all = wl < W ? lds_read_b64_wait(scr_sum + wr * 8) : 0ull;
      const uint64_t cw_msk = wl < W ? lds_read_b64_wait(scr_sum + 128 + wr * 8) : 0ull;
      const uint64_t in_all = row16_scan_u64(cw_all), in_msk = row16_scan_u64(cw_msk);
      // (readlane OUTSIDE the divergent store: inside `if (tid == 0)` hipcc 7.2 returned 0 for the scanned value)
      const uint64_t S_all = readlane_u64(in_all, 15), S_msk = readlane_u64(in_msk, 15);
      if (tid == 0) {
        p.row_sums[2 * cur.pidx] = S_all;
        p.row_sums[2 * cur.pidx + 1] = S_msk;  // below 2^37: redone on its own scale in the tail
        p.row_exps[2 * cur.pidx] = N_all;
        p.row_exps[2 * cur.pidx + 1] = N_all;
        if constexpr (kPhilox) {
          if (ri < RING) lds_write_b32(ring_n + (uint32_t)ri * 8u, __float_as_uint(N_all));
        }
      }
      if constexpr (kBits) {
        uint32_t top = (uint32_t)(S_msk >> 37);  // sums stay below 2^62
        opaque(top);  // VALU compare (uniform u64 `<` miscompile, see the one-workgroup-per-particle kernel)
        if (top == 0u && ri < 64) low_rows |= 1ull << ri;
      }
    }
    GLB_STAMP();
    ++ri;
    if (!has_next) break;
    buf ^= 1;
    cur = nxr;
    vb = vb_next;
    build_bits(cur, buf, mb);
    GLB_STAMP();
  }

  // =================================== tail ===============================================================
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's workspace stores have reached L2
  lds_barrier();
  GLB_STAMP();

  // per-row geometry again (the loop's registers are gone)
  struct TailRow {
    const char *base;
    int a, nv;
    const uint32_t *mrow;
  };
  auto tail_row = [&](int pidx) {
    TailRow r;
    const int row = p.row_of ? p.row_of[pidx] : pidx;
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    r.a = (int)(((uintptr_t)rowp) & 15) / ES;
    r.base = rowp - r.a * ES;
    r.nv = (V + r.a + EPV - 1) / EPV;
    r.mrow = nullptr;
    if constexpr (kBits) {
      const int mi = p.mask_id ? p.mask_id[pidx] : (p.n_masks == 1 ? 0 : pidx);
      r.mrow = (const uint32_t *)p.mask + (int64_t)mi * p.mask_ld;
    }
    return r;
  };
  const int n_words = (V + 31) >> 5;
  // allowed terms of one vector against exponent Nb: per-lane sum (and the terms themselves on request)
  auto vec_terms = [&](const TailRow &r, const uint4 &rk, uint32_t nib, int j0, float Nb, uint64_t(&q)[EPV]) {
    float xs[EPV];
    unpack_vec<DT>(rk, xs);
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < EPV; ++c) {
      const float xv = SCALED ? xs[c] * p.scale : xs[c];
      const bool ok = ((uint32_t)(j0 + c) < (uint32_t)V) && ((nib >> c) & 1u);
      q[c] = ok ? fix_term(xv, Nb) : 0ull;
      s += q[c];
    }
    return s;
  };

  const int n_mine = (n - 1 - (int)blockIdx.x) / G + 1;  // rows blockIdx.x, +G, +2G, ...
  bool any_low = false;
  if constexpr (kBits) {
    uint32_t lo32 = (uint32_t)low_rows | (uint32_t)(low_rows >> 32);
    opaque(lo32);
    any_low = lo32 != 0u || n_mine > 64;  // workgroup-uniform
  }
  if (kBits && any_low) {
    // ---- rows whose allowed tokens hold < 2^-7 of the mass: masked sum again, on the masked maximum's own
    //      scale, from memory, by the whole workgroup (same chunk layout as the loop) --------------------------
#pragma unroll 1
    for (int i = 0, vr = blockIdx.x; vr < n; ++i, vr += G) {
      if (i < 64) {
        if (!((low_rows >> i) & 1ull)) continue;
      }
      const int pidx = particle_of(vr);
      if (i >= 64) {  // beyond the bitmap (> 64 rows per workgroup): ask the workspace
        uint32_t top = (uint32_t)(ld_agent(p.row_sums + 2 * pidx + 1) >> 37);
        opaque(top);
        if (top != 0u) continue;
      }
      const TailRow r = tail_row(pidx);
      float mk = kNegInf;
#pragma unroll 1
      for (int k = 0; k < NVL; ++k) {
        const int v = v0 + k * 64;
        if (v < r.nv) {
          const uint4 rk = *reinterpret_cast<const uint4 *>(r.base + (int64_t)v * 16);
          const int j0 = v * EPV - r.a;
          const uint32_t nib = mask_nibble<EPV>(r.mrow, n_words, j0);
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
      lds_barrier();  // scratch is reused by the next flagged row
      if (!(mk > kNegInf)) continue;  // nothing allowed (workgroup-uniform): the sums on the row's scale are 0
      const float N_k = exp_n(mk);
      const float Nb = N_k + (float)kFixShift;
      uint64_t s_w = 0;
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        uint64_t a_g = 0;
#pragma unroll 1
        for (int t = 0; t < GS; ++t) {
          const int v = v0 + (g * GS + t) * 64;
          if (v < r.nv) {
            const uint4 rk = *reinterpret_cast<const uint4 *>(r.base + (int64_t)v * 16);
            const int j0 = v * EPV - r.a;
            uint64_t q[EPV];
            a_g += vec_terms(r, rk, mask_nibble<EPV>(r.mrow, n_words, j0), j0, Nb, q);
          }
        }
        const uint64_t tg = wave_sum_u64(a_g);
        if (lane == 0) {
          p.chunk_sums[(int64_t)pidx * (W * NG) + wave * NG + g] = tg;
          if constexpr (kPhilox) {
            if (i < RING) lds_write_b64(ring + (uint32_t)((i * W + wave) * NG + g) * 8u, tg);
          }
        }
        s_w += tg;
      }
      if (lane == 0) lds_write_b64(scr_sum + wave * 8, s_w);
      lds_barrier();
      const uint64_t cw = wl < W ? lds_read_b64_wait(scr_sum + wr * 8) : 0ull;
      const uint64_t S = readlane_u64(row16_scan_u64(cw), 15);
      if (tid == 0) {
        p.row_sums[2 * pidx + 1] = S;
        p.row_exps[2 * pidx + 1] = N_k;
        if constexpr (kPhilox) {
          if (i < RING) lds_write_b32(ring_n + (uint32_t)i * 8u, __float_as_uint(N_k));
        }
      }
      lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  GLB_STAMP();

  // ---- lse / logZ and the token of every row this workgroup streamed -----------------------------------------
  if (!(kPhilox && p.out_token)) {
    for (int i = tid; i < n_mine; i += T) finish_row(p, particle_of(blockIdx.x + i * G));
  } else {
    // With at most W/2 rows a PAIR of waves shares a row's draw: both pick the chunk, the even wave takes its
    // first tiles, the odd wave the rest (and the double-precision logs); one LDS word tells the odd wave where
    // its tiles start in the chunk's running sum.  Otherwise one wave per row.
    const bool paired = n_mine <= W / 2;
    const int slot = paired ? (wave >> 1) : wave, half = paired ? (wave & 1) : 0;
    const int step = paired ? W / 2 : W;
    const int tiles = p.chunk_vecs >> 6;  // <= 5
    const int t_split = paired ? (tiles + 1) / 2 : tiles;
    constexpr int kMaxTiles = 5;
    for (int i0 = 0; i0 < n_mine; i0 += step) {  // workgroup-uniform trip count (barrier inside)
      const int i = i0 + slot;
      const bool live = i < n_mine;
      const int pidx = live ? particle_of(blockIdx.x + i * G) : 0;
      uint64_t Tc = 0, run = 0, asel = 0, qsel[EPV];
      uint64_t cj[kMaxTiles], aj[kMaxTiles];
      uint4 rks[kMaxTiles];
      uint32_t nibs[kMaxTiles];
      int j0sel = 0;
      bool found = false, empty = true;
      int t_lo = 0, t_hi = 0;
      float Nb = 0.f;
      TailRow r{};
      int csel = 0;
      if (live) {
        r = tail_row(pidx);
        uint64_t cs = 0;
        if (lane < W * NG)
          cs = i < RING ? lds_read_b64_wait(ring + (uint32_t)(i * W * NG + lane) * 8u)
                        : ld_agent(p.chunk_sums + (int64_t)pidx * (W * NG) + lane);
        const uint64_t incl_c = wave_scan_u64(cs);
        const uint64_t S = readlane_u64(incl_c, 63);
        empty = (S == 0);
        if (!empty) {
          const uint64_t gp = (uint64_t)(p.particle_base + pidx);
          const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
          const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
          uint32_t rnd[4];
          philox4x32_10(ctr, key, rnd);
          Tc = __umul64hi(((uint64_t)rnd[1] << 32) | rnd[0], S);  // uniform integer in [0, S)
          {
            uint32_t z = 0;
            opaque(z);  // VALU compare (uniform u64 `<` miscompile)
            Tc += z;
          }
          csel = __ffsll((long long)__ballot(incl_c > Tc)) - 1;
          Tc -= readlane_u64(incl_c - cs, csel);
          Nb = (i < RING ? __uint_as_float(lds_read_b32_wait(ring_n + (uint32_t)i * 8u))
                         : ld_agent(p.row_exps + 2 * pidx + 1)) + (float)kFixShift;
          t_lo = half ? t_split : 0;
          t_hi = half ? tiles : t_split;
          // all of this wave's tile loads go out before the first is consumed
#pragma unroll
          for (int j = 0; j < kMaxTiles; ++j) {
            const int v = csel * p.chunk_vecs + j * 64 + lane;
            const int vc = v < r.nv ? v : r.nv - 1;
            const bool mine = j >= t_lo && j < t_hi;
            rks[j] = mine ? *reinterpret_cast<const uint4 *>(r.base + (int64_t)vc * 16) : make_uint4(0, 0, 0, 0);
            nibs[j] = (1u << EPV) - 1u;
            if constexpr (kBits) nibs[j] = mine ? mask_nibble<EPV>(r.mrow, n_words, v * EPV - r.a) : 0u;
          }
#pragma unroll
          for (int j = 0; j < kMaxTiles; ++j) {
            aj[j] = 0;
            cj[j] = 0;
            if (j >= t_lo && j < t_hi) {
              const int v = csel * p.chunk_vecs + j * 64 + lane;
              uint64_t q[EPV];
              aj[j] = vec_terms(r, rks[j], nibs[j], v * EPV - r.a, Nb, q);
              cj[j] = wave_sum_u64(aj[j]);
            }
          }
        }
      }
      GLB_STAMP();
      if (paired) {  // even wave -> odd wave: total of the even wave's tiles
        if (live && !empty && half == 0 && lane == 0) {
          uint64_t r0 = 0;
#pragma unroll
          for (int j = 0; j < kMaxTiles; ++j) r0 += cj[j];
          lds_write_b64(scr_sum + slot * 8, r0);
        }
        lds_barrier();
        if (live && !empty && half == 1) run = lds_read_b64_wait(scr_sum + slot * 8);
        lds_barrier();  // the slot is rewritten in the next round
      }
      GLB_STAMP();
      if (live) {
        if (half == (paired ? 1 : 0) && lane == 0) finish_row(p, pidx);
        if (empty) {
          if (half == 0 && lane == 0) p.out_token[pidx] = -1;
        } else {
          int jsel = 0;
#pragma unroll
          for (int j = 0; j < kMaxTiles; ++j) {
            if (j >= t_lo && j < t_hi) {
              if (!found && Tc >= run && Tc - run < cj[j]) {
                found = true;
                Tc -= run;
                asel = aj[j];
                jsel = j;
              }
              run += cj[j];
            }
          }
          if (found) {  // wave-uniform
            const uint64_t incl = wave_scan_u64(asel);
            const int lsel = __ffsll((long long)__ballot(incl > Tc)) - 1;
            if (lane == lsel) {
              uint64_t Tl = Tc - (incl - asel);
              uint4 rsel = rks[0];
              uint32_t nsel = nibs[0];
#pragma unroll
              for (int j = 1; j < kMaxTiles; ++j)
                if (j == jsel) {
                  rsel = rks[j];
                  nsel = nibs[j];
                }
              const int v = csel * p.chunk_vecs + jsel * 64 + lane;
              j0sel = v * EPV - r.a;
              vec_terms(r, rsel, nsel, j0sel, Nb, qsel);
              int32_t tok = -1;
#pragma unroll
              for (int c = 0; c < EPV; ++c) {
                if (tok < 0) {
                  if (Tl < qsel[c]) tok = j0sel + c;
                  else Tl -= qsel[c];
                }
              }
              p.out_token[pidx] = tok;
            }
          }
        }
      }
    }
  }
  GLB_STAMP();
#ifdef GLB_STAMPS
  if (tid == 0 && blockIdx.x < 256) g_realtime[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace glb
