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
// The loop only takes the sums on the row's scale (GLB math: one exponential serves both sums).  The draw rides
// along: the sums pass parks every lane's masked sum per tile (or pair of tiles) in LDS; once the row's total is
// known every wave derives the same Philox target, the wave that owns it binary-searches its tiles in LDS,
// re-requests the one vector (or two) per lane it lands in, and finishes - recompute, lane, element, in
// vocabulary order - after the next row's maximum pass, when those vectors have arrived behind the prefetched
// tiles.  Same integers as the one-workgroup-per-particle kernel.  Rows whose masked sum came out below 2^37 -
// allowed mass below 2^-7 of the row, rare - are redone from memory on the masked maximum's own scale by the
// whole workgroup after the loop, together with their draw (the row's registers were already being refilled when
// the sum became known).  Last, one lane per row turns the sums into lse / logZ (double-precision log).
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
//  in-place tiles only count loads that are issued on every path as "younger", so with conditional requests the
//  maximum pass waited for the look-ahead tiles too (60.5 / 96.8 us for 8 / 12 tiles against 57.8); made
//  unconditional (rows that do not exist read the -inf page) the waits came out right and it was still slower,
//  66 us against 58: the CU's memory queue is not starved during the reductions, the extra requests only delay
//  the next row's own tiles.)
template <int DT, int MASK, int MODE, int NVL, int T, bool SCALED>
__global__ __launch_bounds__(T) void row_kernel_persist(const RowParams p) {
  constexpr int W = T / 64;
  static_assert(W <= 16, "cross-wave scratch is reduced inside one DPP row of 16 lanes");
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr int MBW = (NVL * EPV + 31) / 32;
  constexpr bool kPhilox = MODE == kModePhilox;
  constexpr bool kBits = MASK == kMaskBits;
  static_assert(MASK != kMaskF32 && MODE != kModeNoise, "persistent kernel: mask none/bits, stats/philox");
  // The draw works from per-lane masked sums of groups of GT tiles, parked in LDS by the sums pass: the owning
  // wave binary-searches the groups, then re-reads and recomputes only the GT tiles of one group.
  constexpr int GT = NVL > 25 ? 2 : 1;
  static_assert(NVL % GT == 0 && NVL / GT <= 4 * W, "tile groups must tile a wave's vectors exactly");
  constexpr int NGR = NVL / GT;
  // LDS: [pad | mask row A | mask row B | scratch | group sums]; a mask row covers NVL*T*EPV bits plus slack
  constexpr int MROW_V = kBits ? (NVL * T * EPV / 8 + 15) / 16 + 2 : 0;  // uint4 per buffer
  constexpr int MPT = kBits ? (MROW_V + T - 1) / T : 1;                  // mask vectors staged per thread
  constexpr int MAXR = 64;                                               // rows whose input indices are tabled in LDS
  constexpr int SCR_V = 24 + 16 + MAXR / 2;                              // 2x16 floats, 2x16 u64, 32 u64 group totals, index table
  constexpr int GSUM_V = kPhilox ? NGR * T / 2 : 0;                      // [NGR][T] u64
  static_assert((1 + 2 * MROW_V + SCR_V + GSUM_V) * 16 <= 160 * 1024, "LDS budget");
  __shared__ uint4 s_lds[1 + 2 * MROW_V + SCR_V + GSUM_V];
  const uint32_t scr = lds_addr(s_lds + 1 + 2 * MROW_V);
  const uint32_t scr_max = scr, scr_sum = scr + 128, scr_tot = scr + 384, scr_idx = scr + 640;
  uint64_t *const s_gsum = reinterpret_cast<uint64_t *>(s_lds + 1 + 2 * MROW_V + SCR_V);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  const int n = p.n_particles, G = gridDim.x;
  const int v0 = wave * (64 * NVL) + lane;
  const int wl = lane & 15;            // lane wl of every DPP row stands for wave wl
  const int wr = wl < W ? wl : W - 1;  // rows of 16 lanes but only W waves: the rest duplicate the last
  const int n_words = (V + 31) >> 5;

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
  // (row_of, mask_id) of the workgroup's first MAXR rows are fetched once, up front, into LDS: a scalar load at
  // the top of every row stalls the wave for its full latency
  auto row_ref = [&](int vb, int i) {
    RowRef r;
    r.pidx = particle_of(vb);
    int row, mi;
    if (i > 0 && i < MAXR) {
      const uint64_t e = lds_read_b64_wait(scr_idx + (uint32_t)i * 8u);
      row = __builtin_amdgcn_readfirstlane((int)(uint32_t)e);
      mi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(e >> 32));
    } else {
      row = p.row_of ? sload_i32(p.row_of, r.pidx) : r.pidx;
      mi = !kBits ? 0 : (p.mask_id ? sload_i32(p.mask_id, r.pidx) : (p.n_masks == 1 ? 0 : r.pidx));
    }
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    r.a = (int)(((uintptr_t)rowp) & 15) / ES;
    r.base = (gptr_t)(rowp - r.a * ES);
    r.nv = (V + r.a + EPV - 1) / EPV;
    r.mrow16 = nullptr;
    r.am = 0;
    r.mvec = 1;
    if constexpr (kBits) {
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
  // allowed terms of one vector (first element j0, mask bits nib) against exponent Nb: per-lane sum and terms
  auto vec_terms = [&](const uint4 &rk, uint32_t nib, int j0, float Nb, uint64_t(&q)[EPV]) {
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
  auto philox_bits = [&](int pidx) {  // the particle's 64 random bits (independent of the row's sums)
    const uint64_t gp = (uint64_t)(p.particle_base + pidx);
    const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
    const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
    uint32_t rnd[4];
    philox4x32_10(ctr, key, rnd);
    return ((uint64_t)rnd[1] << 32) | rnd[0];
  };
  auto target_of = [&](uint64_t R, uint64_t S) {  // uniform integer in [0, S), kept in VGPRs
    uint64_t Tc = __umul64hi(R, S);
    uint32_t z = 0;
    opaque(z);  // VALU compares (hipcc lowers a wave-uniform u64 `<` feeding a select to v_cmp + s_cselect w/o SCC)
    return Tc + z;
  };

  uint4 raw[NVL];
  uint4 mreg[MPT];
  uint32_t mb[MBW];

  // The draw of row r is spread over the next two rows so that none of it sits in front of a barrier:
  //   (1) after row r's totals are known every wave derives the Philox target and the owning wave, and the
  //       waves share out the owning wave's tile groups: group totals (64-lane sums of the LDS entries) -> LDS;
  //   (2) after the next barrier the owning wave scans the <= 32 group totals, picks the group and re-requests its
  //       GT vectors per lane (+ mask bits) - they queue behind nothing but row r+1's first tiles;
  //   (3) in row r+2's maximum pass, which mostly waits for memory, it recomputes those vectors and walks
  //       tile -> lane -> element in vocabulary order.
  struct DrawJob {  // wave-uniform
    bool live;
    int owner, pidx, a, nv;
    gptr_t base;
    const uint32_t *mrow;
    float Nb;
    uint64_t target;
  };
  DrawJob job{};
  job.live = false;
  bool pend = false;  // wave-uniform: stage (2) done, (3) outstanding
  uint4 pend_rk[GT];
  uint32_t pend_nib[GT];
  uint64_t pend_T = 0;
  float pend_Nb = 0.f;
  int pend_j0 = 0, pend_pidx = 0;
  auto pick_group = [&]() {  // stage (2)
    if (!job.live) return;
    job.live = false;
    if (wave != job.owner) return;
    const uint64_t t = lane < NGR ? lds_read_b64_wait(scr_tot + (uint32_t)lane * 8u) : 0ull;
    const uint64_t incl = wave_scan_u64(t);
    uint64_t Tc = job.target;
    const int gsel = __ffsll((long long)__ballot(incl > Tc)) - 1;
    Tc -= readlane_u64(incl - t, gsel);
    pend = true;
    pend_T = Tc;
    pend_Nb = job.Nb;
    pend_pidx = job.pidx;
    pend_j0 = (v0 + gsel * GT * 64) * EPV - job.a;
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      const int v = v0 + (gsel * GT + g) * 64;
      pend_rk[g] = gload16(job.base + (int64_t)(v < job.nv ? v : job.nv - 1) * 16);
      pend_nib[g] = (1u << EPV) - 1u;
      if constexpr (kBits) pend_nib[g] = mask_nibble<EPV>(job.mrow, n_words, v * EPV - job.a);
    }
  };
  auto finish_draw = [&]() {
    if (!pend) return;
    pend = false;
    uint64_t Tc = pend_T;
    uint64_t aj[GT], cj[GT], q[GT][EPV];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      aj[g] = vec_terms(pend_rk[g], pend_nib[g], pend_j0 + g * 64 * EPV, pend_Nb, q[g]);
      cj[g] = GT > 1 ? wave_sum_u64(aj[g]) : 0ull;
    }
    int gsel = 0;
    if constexpr (GT > 1) {
      if (!(Tc < cj[0])) {
        Tc -= cj[0];
        gsel = 1;
      }
    }
    const uint64_t asel = gsel ? aj[GT - 1] : aj[0];
    const uint64_t incl = wave_scan_u64(asel);
    const int lsel = __ffsll((long long)__ballot(incl > Tc)) - 1;
    if (lane == lsel) {
      uint64_t Tl = Tc - (incl - asel);
      const int j0 = pend_j0 + gsel * 64 * EPV;
      int32_t tok = -1;
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        const uint64_t qc = gsel ? q[GT - 1][c] : q[0][c];
        if (tok < 0) {
          if (Tl < qc) tok = j0 + c;
          else Tl -= qc;
        }
      }
      p.out_token[pend_pidx] = tok;
    }
  };

  int vb = blockIdx.x;
  if (vb >= n) return;
#ifdef GLB_STAMPS
  int stamp_i = 0;
  if (tid == 0 && blockIdx.x < 256) g_realtime[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
  GLB_STAMP();
  RowRef cur = row_ref(vb, 0);
  int buf = 0;
  int ri = 0;             // index of the current row among this workgroup's rows
  uint64_t low_rows = 0;  // bit i: row i (< 64) needs its masked sum redone on its own scale (workgroup-uniform)
  uint64_t tab_e = 0;  // this thread's entry of the index table (requested ahead of the tiles, written behind them)
  const bool tab_mine = tid > 0 && tid < MAXR && tid < (n - 1 - (int)blockIdx.x) / G + 1;
  if (tab_mine) {
    const int pi = particle_of(blockIdx.x + tid * G);
    const uint32_t row = (uint32_t)(p.row_of ? p.row_of[pi] : pi);
    const uint32_t mi = !kBits ? 0u : (uint32_t)(p.mask_id ? p.mask_id[pi] : (p.n_masks == 1 ? 0 : pi));
    tab_e = ((uint64_t)mi << 32) | row;
  }
  load_mask(cur, mreg);
#pragma unroll
  for (int k = 0; k < NVL; ++k) raw[k] = load_tile(cur, k);
  if (tab_mine) lds_write_b64(scr_idx + (uint32_t)tid * 8u, tab_e);
  stage_mask(mreg, 0);
  lds_barrier();
  build_bits(cur, 0, mb);
  GLB_STAMP();

  for (;;) {
    const int vb_next = vb + G;
    const bool has_next = vb_next < n;
    RowRef nxr = cur;
    if (has_next) {
      nxr = row_ref(vb_next, ri + 1);
      load_mask(nxr, mreg);  // lands long before the row tiles issued behind it
    }
    opaque(lane_off);
    // stage (3) of the row before the previous one: its vectors were requested ahead of all of this row's tiles,
    // so they are here, and the arithmetic runs while the tail of the prefetch is still arriving
    if constexpr (kPhilox) finish_draw();

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
    uint64_t R_cur = 0;
    if constexpr (kPhilox) {
      pick_group();  // stage (2) of the previous row (its group totals are behind the barrier)
      if (p.out_token) R_cur = philox_bits(cur.pidx);  // here, where the SIMD's other wave fills the issue slots
    }

    // ---- phase 2: both fixed-point sums on the row's scale; tile k of the next row is requested as soon
    //      as tile k of this one has been consumed --------------------------------------------------------
    uint64_t acc = 0, s_msk = 0, grp = 0, prev = 0;
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
        grp += ak;
      }
      if constexpr (kPhilox) {
        if (k % GT == GT - 1) {  // this lane's masked sum of the tile group -> LDS, for the draw
          if constexpr (!kBits) {  // unmasked: difference of the running sum at the group's ends
            grp = acc - prev;
            prev = acc;
          }
          s_gsum[(k / GT) * T + tid] = grp;
          grp = 0;
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
      }
      bool low = false;
      if constexpr (kBits) {
        uint32_t top = (uint32_t)(S_msk >> 37);  // sums stay below 2^62
        opaque(top);  // VALU compare (uniform u64 `<` miscompile, see the one-workgroup-per-particle kernel)
        low = top == 0u;
        if (low && ri < 64) low_rows |= 1ull << ri;
      }
      if constexpr (kPhilox) {
        // ---- the draw: every wave derives the same target; the wave that owns it searches its tile groups in
        //      LDS, re-requests the GT vectors of the group it lands in and carries on (finish_draw) ----------
        if (p.out_token && !low) {  // low-mass rows (incl. nothing allowed) are drawn in the tail
          uint64_t Tc = target_of(R_cur, S_msk);
          const int owner = __ffsll((long long)(__ballot(in_msk > Tc) & 0xffffull)) - 1;  // lanes 0..15 <-> waves
          Tc -= readlane_u64(in_msk - cw_msk, owner);
          {  // group totals of the owning wave's entries: 16 lanes x 4 entries per group, 4 groups per wave
            const int g = wave * 4 + (lane >> 4);
            uint64_t v = 0;
            if (g < NGR) {
              const uint64_t *src = s_gsum + g * T + owner * 64 + (lane & 15) * 4;
              v = (src[0] + src[1]) + (src[2] + src[3]);
            }
            const uint64_t tot = row16_scan_u64(v);
            if (g < NGR && (lane & 15) == 15) lds_write_b64(scr_tot + (uint32_t)g * 8u, tot);
          }
          job.live = true;
          job.owner = owner;
          job.pidx = cur.pidx;
          job.a = cur.a;
          job.nv = cur.nv;
          job.base = cur.base;
          job.mrow = kBits ? reinterpret_cast<const uint32_t *>((const char *)cur.mrow16) + cur.am : nullptr;
          job.Nb = Nb_all;
          job.target = Tc;
        }
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
  if constexpr (kPhilox) {
    finish_draw();  // stage (3) of the last row but one
    lds_barrier();  // the last row's group totals
    pick_group();   // its stage (2) ...
    finish_draw();  // ... and (3)
  }
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

  const int n_mine = (n - 1 - (int)blockIdx.x) / G + 1;  // rows blockIdx.x, +G, +2G, ...
  bool any_low = false;
  if constexpr (kBits) {
    uint32_t lo32 = (uint32_t)low_rows | (uint32_t)(low_rows >> 32);
    opaque(lo32);
    any_low = lo32 != 0u || n_mine > 64;  // workgroup-uniform
  }
  if (kBits && any_low) {
    // ---- rows whose allowed tokens hold < 2^-7 of the mass: masked sum again, on the masked maximum's own
    //      scale, from memory, by the whole workgroup; then their draw, by the owning wave walking its tiles ----
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
      if (!(mk > kNegInf)) {  // nothing allowed (workgroup-uniform): the sums on the row's scale are 0 already
        if (kPhilox && p.out_token && tid == 0) p.out_token[pidx] = -1;
        continue;
      }
      const float N_k = exp_n(mk);
      const float Nb = N_k + (float)kFixShift;
      uint64_t a_w = 0;
#pragma unroll 1
      for (int k = 0; k < NVL; ++k) {
        const int v = v0 + k * 64;
        if (v < r.nv) {
          const uint4 rk = *reinterpret_cast<const uint4 *>(r.base + (int64_t)v * 16);
          const int j0 = v * EPV - r.a;
          uint64_t q[EPV];
          a_w += vec_terms(rk, mask_nibble<EPV>(r.mrow, n_words, j0), j0, Nb, q);
        }
      }
      const uint64_t s_w = wave_sum_u64(a_w);
      if (lane == 0) lds_write_b64(scr_sum + wave * 8, s_w);
      lds_barrier();
      const uint64_t cw = wl < W ? lds_read_b64_wait(scr_sum + wr * 8) : 0ull;
      const uint64_t in_w = row16_scan_u64(cw);
      const uint64_t S = readlane_u64(in_w, 15);
      if (tid == 0) {
        p.row_sums[2 * pidx + 1] = S;
        p.row_exps[2 * pidx + 1] = N_k;
      }
      if (kPhilox && p.out_token) {
        if (S == 0) {
          if (tid == 0) p.out_token[pidx] = -1;
        } else {
          uint64_t Tc = target_of(philox_bits(pidx), S);
          const int owner = __ffsll((long long)(__ballot(in_w > Tc) & 0xffffull)) - 1;
          Tc -= readlane_u64(in_w - cw, owner);
          if (wave == owner) {  // tile by tile until the running sum passes the target
            bool done = false;
#pragma unroll 1
            for (int k = 0; k < NVL && !done; ++k) {
              const int v = v0 + k * 64;
              const int j0 = v * EPV - r.a;
              uint64_t q[EPV], aj = 0;
#pragma unroll
              for (int c = 0; c < EPV; ++c) q[c] = 0;
              if (v < r.nv) {
                const uint4 rk = *reinterpret_cast<const uint4 *>(r.base + (int64_t)v * 16);
                aj = vec_terms(rk, mask_nibble<EPV>(r.mrow, n_words, j0), j0, Nb, q);
              }
              const uint64_t incl = wave_scan_u64(aj);
              const uint64_t tile = readlane_u64(incl, 63);
              if (Tc < tile) {
                const int lsel = __ffsll((long long)__ballot(incl > Tc)) - 1;
                if (lane == lsel) {
                  uint64_t Tl = Tc - (incl - aj);
                  int32_t tok = -1;
#pragma unroll
                  for (int c = 0; c < EPV; ++c) {
                    if (tok < 0) {
                      if (Tl < q[c]) tok = j0 + c;
                      else Tl -= q[c];
                    }
                  }
                  p.out_token[pidx] = tok;
                }
                done = true;
              } else {
                Tc -= tile;
              }
            }
          }
        }
      }
      lds_barrier();  // scratch is reused by the next flagged row
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  GLB_STAMP();

  // ---- lse / logZ of every row this workgroup streamed (double-precision logs: one lane per row) ------------
  for (int i = tid; i < n_mine; i += T) finish_row(p, particle_of(blockIdx.x + i * G));
  GLB_STAMP();
#ifdef GLB_STAMPS
  if (tid == 0 && blockIdx.x < 256) g_realtime[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace glb
