// glb_row_kernel_v2.hpp — persistent, software-pipelined variant of the fused particle-step kernel.
//
// v1 (glb_row_kernel.hpp) runs one 1024-thread workgroup per particle with the row in VGPRs; only
// one such workgroup fits a CU, so a row's HBM load never overlaps another row's arithmetic and the
// kernel stops at ~27 % of the HBM roofline.  v2 keeps ONE workgroup per CU for the whole launch and
// streams rows through it: while row i is reduced from registers, row i+1 is already in flight —
// NL of its NVL vectors per lane by LDS-DMA (`global_load_lds_dwordx4`: no VGPRs, each wave fills its
// own 1 KiB-per-instruction slots and later reads back only what it wrote, so no barrier guards the
// staging area), ND by ordinary loads into spare registers, the mask bit row by LDS-DMA too.
//
// What had to change for that (gfx950 / ROCm 7.2, see cdna_hip_programming.md §5 "Pipelining across
// barriers"): no `__syncthreads()` (its fence drains vmcnt), no use of an ordinary load's result and
// no compiler-visible LDS read while a DMA is in flight.  Barriers are raw `s_barrier`, the
// cross-wave scratch is accessed with inline-asm `ds_*`, the one `s_waitcnt vmcnt(0)` per row is
// explicit, and the inverse-CDF draw is finished after the streaming loop (locate_row, one wave per row) from
// per-chunk wave totals the loop leaves in a workspace: it re-reads one 4-5 KiB chunk of one row per particle.
//
// Arithmetic is GLB math exactly as in v1: results are bit-identical to v1 and to the oracle.
#pragma once
#include "glb_row_kernel.hpp"

namespace glb {

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// 1 KiB of -inf per element type: vectors past the end of a row are fetched from here instead of being
// patched after the load (f32 | bf16 | f16 bit patterns)
__device__ const uint32_t g_neg_inf_page[3][256] = {
    {0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u},
    {0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u},
    {0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u}};

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
// all LDS writes of this wave done, then workgroup barrier — never touches vmcnt
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

// ---- per-particle epilogue (run by the tail of the persistent kernel, one wave per particle) ----------
// The workspace words were written by other waves of this workgroup with plain stores (complete: vmcnt(0) +
// barrier); agent-scope loads keep the reads out of any stale vector-L1 line.
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

// Second half of the Philox draw, one wave per particle.  From the chunk totals the streaming loop left in the
// workspace it picks the chunk (64-lane scan), re-reads that chunk's 4-5 KiB of the row (L2 / Infinity Cache),
// recomputes the per-lane tile sums against the stored exponent and walks tile -> lane -> element in
// vocabulary order.  Same integers as the one-workgroup-per-particle kernel, same token.
template <int DT, int MASK>
__device__ __forceinline__ void locate_row(const RowParams &p, int pidx, int lane) {
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
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
  const uint64_t cs = lane < p.n_chunks ? ld_agent(p.chunk_sums + (int64_t)pidx * p.n_chunks + lane) : 0ull;
  const uint64_t incl_c = wave_scan_u64(cs);
  const uint64_t S = readlane_u64(incl_c, 63);
  if (lane == 0) finish_row(p, pidx);
  if (S == 0) {
    if (lane == 0) p.out_token[pidx] = -1;
    return;
  }
  const uint64_t gp = (uint64_t)(p.particle_base + pidx);
  const uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)p.offset, (uint32_t)(p.offset >> 32)};
  const uint32_t key[2] = {(uint32_t)p.seed, (uint32_t)(p.seed >> 32)};
  uint32_t rnd[4];
  philox4x32_10(ctr, key, rnd);
  uint64_t Tc = __umul64hi(((uint64_t)rnd[1] << 32) | rnd[0], S);
  {
    uint32_t z = 0;
    opaque(z);  // VALU compare (uniform u64 `<` miscompile, see v1)
    Tc += z;
  }
  const int csel = __ffsll((long long)__ballot(incl_c > Tc)) - 1;
  Tc -= readlane_u64(incl_c - cs, csel);
  const float Nb = ld_agent(p.row_exps + 2 * pidx + 1) + (float)kFixShift;
  const int tiles = p.chunk_vecs >> 6;  // <= 5
  uint64_t run = 0, asel = 0;
  uint4 rsel = make_uint4(0, 0, 0, 0);
  uint32_t nsel = 0;
  int j0sel = 0;
  bool found = false;
  // all of the chunk's loads go out before the first is consumed
  constexpr int kMaxTiles = 5;
  uint4 rks[kMaxTiles];
  uint32_t nibs[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    const int v = csel * p.chunk_vecs + j * 64 + lane;
    const int vc = v < nv ? v : nv - 1;
    rks[j] = j < tiles ? *reinterpret_cast<const uint4 *>(base + (int64_t)vc * 16) : make_uint4(0, 0, 0, 0);
    nibs[j] = (1u << EPV) - 1u;
    if constexpr (MASK == kMaskBits) nibs[j] = j < tiles ? mask_nibble<EPV>(mrow, (V + 31) >> 5, v * EPV - a) : 0u;
  }
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    if (j < tiles) {
      const int v = csel * p.chunk_vecs + j * 64 + lane;
      const uint4 rk = rks[j];
      const int j0 = v * EPV - a;
      const uint32_t nib = nibs[j];
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
      const uint64_t q = ok ? fix_term(xv, Nb) : 0ull;
      if (tok < 0) {
        if (Tl < q) tok = j0sel + c;
        else Tl -= q;
      }
    }
    p.out_token[pidx] = tok;
  }
}

// tuning aid (tools/dbg/stamps.hip): shader-clock stamps of the first and last wave at the phase boundaries
#ifdef GLB_STAMPS
__device__ unsigned long long g_stamps[256 * 2 * 64];
__device__ unsigned long long g_realtime[256 * 2];
__device__ int g_skew_cycles = 0;   // experiment: odd workgroups start this many cycles late
#define GLB_STAMP()                                                                                        \
  do {                                                                                                     \
    if (lane == 0 && (wave == 0 || wave == W - 1) && stamp_i < 64 && blockIdx.x < 256)                     \
      g_stamps[(blockIdx.x * 2 + (wave != 0)) * 64 + stamp_i] = __builtin_amdgcn_s_memtime();              \
    ++stamp_i;                                                                                             \
  } while (0)
#else
#define GLB_STAMP() do { } while (0)
#endif

template <int DT, int MASK, int MODE, int NVL, int NL, int ND, int T, bool SCALED>
__global__ __launch_bounds__(T) void row_kernel_v2(const RowParams p) {
  constexpr int W = T / 64;
  static_assert(NVL <= 35 && W <= 16, "cross-wave scratch is reduced inside one DPP row of 16 lanes");
  constexpr int EPV = ElemTraits<DT>::EPV;
  constexpr int ES = ElemTraits<DT>::ES;
  constexpr int NX = NVL - NL - ND;  // vectors loaded late (after the row is consumed), exposed
  constexpr int MBW = (NVL * EPV + 31) / 32;
  constexpr bool kPhilox = MODE == kModePhilox;
  constexpr bool kBits = MASK == kMaskBits;
  static_assert(NX >= 0 && MASK != kMaskF32 && MODE != kModeNoise, "v2 covers mask none/bits, stats/philox");
  // one LDS object (a second one makes hipcc drain vmcnt before LDS reads): [stage | mask row | scratch]
  constexpr int STAGE_V = NL * T;                                   // uint4 slots
  constexpr int MASK_V = kBits ? (((NVL * T * EPV / 8 + 15) / 16 + 2) + 63) / 64 * 64 : 0;  // whole 1 KiB chunks
  constexpr int GS = (NVL % 5 == 0) ? 5 : 4;                        // tiles per chunk of the draw's search
  constexpr int NG = NVL / GS;                                      // chunks per wave
  static_assert(NVL % GS == 0, "chunks must tile a wave's vectors exactly (locate_kernel indexes them linearly)");
  constexpr int SCR_V = 24;                                         // 2x16 floats, 2x16 u64
  __shared__ uint4 s_lds[STAGE_V + MASK_V + SCR_V];
  uint4 *s_stage = s_lds;
  const uint32_t *s_mask = reinterpret_cast<const uint32_t *>(s_lds + STAGE_V);
  const uint32_t scr = lds_addr(s_lds + STAGE_V + MASK_V);  // byte address of the scratch area
  const uint32_t scr_max = scr, scr_sum = scr + 128;  // [2][16] f32, [2][16] u64

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  const int n = p.n_particles, G = gridDim.x;
  const int v0 = wave * (64 * NVL) + lane;

  // row bookkeeping for "virtual block" vb (same XCD-aware particle order as v1)
  auto particle_of = [&](int vb) {
    const int q = n >> 3, r = n & 7, xcd = vb & 7, i = vb >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  };
  struct RowRef {
    const char *base;  // 16-byte aligned-down row start
    int a, nv;         // leading pad elements, vectors covering the row
    const uint32_t *mrow16;  // 16-byte aligned-down mask row
    int am, mvec;            // leading pad words, 16-byte vectors covering the mask row
    int pidx;
  };
  auto row_ref = [&](int vb) {
    RowRef r;
    r.pidx = particle_of(vb);
    const int row = p.row_of ? p.row_of[r.pidx] : r.pidx;
    const char *rowp = (const char *)p.logits + (int64_t)row * p.ld * ES;
    r.a = (int)(((uintptr_t)rowp) & 15) / ES;
    r.base = rowp - r.a * ES;
    r.nv = (V + r.a + EPV - 1) / EPV;
    r.mrow16 = nullptr;
    r.am = 0;
    r.mvec = 0;
    if constexpr (kBits) {
      const int mi = p.mask_id ? p.mask_id[r.pidx] : (p.n_masks == 1 ? 0 : r.pidx);
      const char *mp = (const char *)p.mask + (int64_t)mi * p.mask_ld * 4;
      r.am = (int)(((uintptr_t)mp) & 15) / 4;
      r.mrow16 = (const uint32_t *)(mp - r.am * 4);
      r.mvec = (((V + 31) >> 5) + r.am + 3) >> 2;
    }
    return r;
  };

  const char *ninf = (const char *)g_neg_inf_page[DT];
  uint4 raw[NVL];
  uint4 nxt[ND > 0 ? ND : 1];
  uint32_t mb[MBW];

  // Prefetch of a row, one slot at a time.  Issuing all of a row's loads at once blocks the issuing
  // waves for most of the transfer (the CU's memory queue holds far less than 200 KB of requests:
  // measured 6-10k cycles at issue, 0.4k at the final vmcnt wait), so the slots are issued from inside
  // phase 2, one per tile of arithmetic: slot k < NL by LDS-DMA into this wave's staging slot, the next
  // ND into `nxt`, plus this wave's chunks of the mask bit row.
  uint32_t lane_off = (uint32_t)v0 * 16u;  // byte offset of this lane's first vector; re-opaqued per row so
                                           // the 2*NVL slot addresses are not hoisted out of the row loop
  auto prefetch_slot = [&](const RowRef &r, int k, bool with_mask) {
    const uint32_t off = lane_off + (uint32_t)k * 1024u;
    // both candidates are GLOBAL pointers (address space 1): a generic select would turn the ND loads into
    // flat_load, in front of which hipcc drains vmcnt (every in-flight DMA) and lgkmcnt
    typedef const __attribute__((address_space(1))) char *gptr_t;
    const gptr_t src = off < (uint32_t)r.nv * 16u ? (gptr_t)r.base + off : (gptr_t)ninf + lane * 16;  // past the row: -inf page
    if (k < NL) {
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                       (lds_ptr_t)(s_stage + (wave * NL + k) * 64), 16, 0, 0);
    } else if (k < NL + ND) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 t = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(src);
      nxt[k - NL] = make_uint4(t.x, t.y, t.z, t.w);
    }
    if constexpr (kBits) {
      constexpr int MCH = (MASK_V / 64 + W - 1) / W;  // mask chunks per wave
      if (with_mask && k < MCH) {
        const int m = wave + k * W;  // chunk m = mask vectors [64m, 64m+64)
        if (m * 64 < r.mvec) {
          int mv = m * 64 + lane;
          mv = mv < r.mvec ? mv : r.mvec - 1;
          __builtin_amdgcn_global_load_lds(
              (const void __attribute__((address_space(1))) *)((const char *)r.mrow16 + (int64_t)mv * 16),
              (lds_ptr_t)(s_lds + STAGE_V + m * 64), 16, 0, 0);
        }
      }
    }
  };
  auto prefetch_mask = [&](const RowRef &r, int k) {
    if constexpr (kBits) {
      const int m = wave + k * W;
      if (m * 64 < r.mvec) {
        int mv = m * 64 + lane;
        mv = mv < r.mvec ? mv : r.mvec - 1;
        __builtin_amdgcn_global_load_lds(
            (const void __attribute__((address_space(1))) *)((const char *)r.mrow16 + (int64_t)mv * 16),
            (lds_ptr_t)(s_lds + STAGE_V + m * 64), 16, 0, 0);
      }
    }
  };
  auto prefetch = [&](const RowRef &r) {
#pragma unroll
    for (int k = 0; k < NVL; ++k) prefetch_slot(r, k, true);
  };

  // after the explicit vmcnt(0) (+ barrier for the mask row): staged data -> registers, late vectors,
  // mask nibbles, invalidation of out-of-row elements
  auto land = [&](const RowRef &r) {
#pragma unroll
    for (int k = 0; k < NL; ++k) raw[k] = s_stage[(wave * NL + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < ND; ++k) raw[NL + k] = nxt[k];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int v = v0 + (NL + ND + k) * 64;
      raw[NL + ND + k] = *reinterpret_cast<const uint4 *>(v < r.nv ? r.base + (int64_t)v * 16 : ninf + lane * 16);
    }
    // late vectors must have landed before the next prefetch is issued: hipcc would otherwise wait
    // vmcnt(0) at their first use and drain the DMA with them
    if constexpr (NX > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int j00 = v0 * EPV - r.a;  // element index of this lane's first element (may be -a..-1)
#pragma unroll
    for (int i = 0; i < MBW; ++i) mb[i] = kBits ? 0u : 0xffffffffu;
    if constexpr (kBits) {
      // tile k's EPV bits start at bit j00 + 64*EPV*k of the staged mask row: a fixed shift and a word
      // index that advances by 2*EPV per tile.  Words outside the staged row (index -1, or past the end)
      // are harmless LDS reads: they only feed bits of elements that are -inf anyway.
      const uint32_t shb = (uint32_t)(j00 & 31);
      const uint32_t *mw = s_mask + ((j00 >> 5) + r.am);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const uint32_t f = __builtin_amdgcn_alignbit(mw[k * 2 * EPV + 1], mw[k * 2 * EPV], shb);
        mb[(k * EPV) >> 5] |= (f & ((1u << EPV) - 1u)) << ((k * EPV) & 31);
      }
    }
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
    if (tid == 0 && r.a > 0) patch(raw[0], r.a, EPV);
    const int vlast = r.nv - 1;                      // last vector of the row
    const int klast = (vlast - wave * (64 * NVL)) >> 6;  // its tile in this wave (if it is this wave's)
#pragma unroll
    for (int k = 0; k < NVL; ++k)
      if (k == klast && v0 + k * 64 == vlast) patch(raw[k], (vlast == 0) ? r.a : 0, V + r.a - vlast * EPV);
  };

  // ---- prologue: first row straight to registers (nothing to overlap with yet) ---------------------
  int vb = blockIdx.x;
  if (vb >= n) return;
#ifdef GLB_STAMPS
  int stamp_i = 0;
  if (tid == 0 && blockIdx.x < 256) g_realtime[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef GLB_STAMPS
  if ((blockIdx.x >> 3) & 1) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)g_skew_cycles) __builtin_amdgcn_s_sleep(8);
  }
#endif
  GLB_STAMP();  // 0: start
  RowRef cur = row_ref(vb);
  prefetch(cur);
  GLB_STAMP();  // 1: first row's loads issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (kBits) lds_barrier();
  GLB_STAMP();  // 2: ... arrived
  land(cur);
  GLB_STAMP();  // 3: landed        then per row: +1 phase 1, +2 phase 2 body, +3 reduce, [+4 wait, +5 landed]

  for (;;) {
    const int vb_next = vb + G;
    const bool has_next = vb_next < n;
    RowRef nxr = cur;
    if (has_next) nxr = row_ref(vb_next);
    opaque(lane_off);
    // head start: the memory queue is empty right after `land`, so the first KB slots go out in one burst
    // (private staging slots only; the shared mask row waits for the barrier after phase 1)
    // (not with bit masks: that combination produced wrong sums on hardware - an unexplained race - so the
    //  masked kernels keep issuing every slot from inside phase 2)
    constexpr int KB = (NVL >= 12 && !kBits) ? 6 : 0;
    if (has_next) {
#pragma unroll
      for (int k = 0; k < KB; ++k) prefetch_slot(nxr, k, false);
    }

    // ---- phase 1: row maximum (the masked maximum is only needed on the rare low-mass path below) -----
    float m_all = kNegInf;
#pragma unroll
    for (int k = 0; k < NVL; ++k) {
      float xk[EPV];
      unpack_vec<DT>(raw[k], xk);
#pragma unroll
      for (int c = 0; c < EPV; ++c) m_all = fmaxf(m_all, SCALED ? xk[c] * p.scale : xk[c]);
    }
    m_all = wave_max(m_all);
    if (lane == 0) lds_write_b32(scr_max + wave * 4, __float_as_uint(m_all));
    lds_barrier();
    const int wl = lane & 15;            // lane wl of every DPP row stands for wave wl
    const int wr = wl < W ? wl : W - 1;  // rows of 16 lanes but only W waves: the rest duplicate the last
    m_all = row16_max_bcast(__uint_as_float(lds_read_b32_wait(scr_max + wr * 4)));
    const float N_all = __builtin_rintf(m_all * kLog2e);
    float N_msk = N_all;
    GLB_STAMP();

    // ---- phase 2: fixed-point sums -----------------------------------------------------------------
    // Pass 0 takes both sums on the row's scale N_all: one exponential per element, the masked term is
    // the unmasked term gated by the mask bit.  Only if that leaves the masked sum fewer than 37 bits
    // (allowed mass < 2^-7 of the row: workgroup-uniform, rare) pass 1 redoes it on the masked maximum's scale.
    uint64_t S_all = 0, S_msk = 0;
    uint64_t ag[kPhilox ? NG : 1];  // per-lane masked sums of the chunks (GS tiles each)
    float Nb_cur = N_all + (float)kFixShift;
#pragma unroll 1
    for (int pass = 0;; ++pass) {
      uint64_t acc = 0, s_msk = 0;
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
          exp_fix4(xk[4 * h], xk[4 * h + 1], xk[4 * h + 2], xk[4 * h + 3], Nb_cur, pf, sh);
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
        // pin both running sums here: otherwise the masked adds are reassociated and sunk below the
        // loop, which keeps every q of the row alive (200 VGPRs)
        opaque(s_msk);
        opaque(acc);
        // next row's slot k goes out now: this wave's staging slot k was emptied by `land`, and every wave
        // is past `land` (barrier after phase 1), so the shared mask row may be refilled too
        if (has_next && pass == 0) {
          if (k >= KB) prefetch_slot(nxr, k, false);
          if constexpr (kBits) {
            constexpr int MCH2 = (MASK_V / 64 + W - 1) / W;
            if (k < MCH2) prefetch_mask(nxr, k);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one vector at a time (register pressure)
      }
      if constexpr (!kBits) s_msk = acc;
      if (pass == 0) GLB_STAMP();
      {
        const uint64_t t_all = wave_scan_u64(acc);
        uint64_t t_msk = t_all;
        if constexpr (kBits) t_msk = wave_scan_u64(s_msk);
        if (lane == 63) {
          lds_write_b64(scr_sum + wave * 8, t_all);
          lds_write_b64(scr_sum + 128 + wave * 8, t_msk);
        }
        if constexpr (kPhilox) {
          // wave totals of every chunk -> workspace; locate_kernel finishes the draw from them
          uint64_t *crow = p.chunk_sums + (int64_t)cur.pidx * (W * NG) + wave * NG;
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const uint64_t tg = wave_scan_u64(ag[g]);
            if (lane == 63) crow[g] = tg;
          }
        }
      }
      lds_barrier();
      // lane l (mod 16) holds wave l's totals; a 16-lane scan gives the totals
      const uint64_t cw_all = wl < W ? lds_read_b64_wait(scr_sum + wr * 8) : 0ull;
      const uint64_t cw_msk = wl < W ? lds_read_b64_wait(scr_sum + 128 + wr * 8) : 0ull;
      const uint64_t in_all = row16_scan_u64(cw_all), in_msk = row16_scan_u64(cw_msk);
      if (pass == 0) S_all = readlane_u64(in_all, 15);
      S_msk = readlane_u64(in_msk, 15);
      if constexpr (!kBits) break;
      uint32_t top = (uint32_t)(S_msk >> 37);  // sums stay below 2^62
      opaque(top);  // VALU compare (uniform u64 `<` miscompile, see v1)
      const bool low = (top == 0u);
      if (pass == 1 || !low) break;
      // ---- rare: the allowed tokens hold < 2^-7 of the row's mass; redo the masked sum on their own scale ----
      float m_msk = kNegInf;
#pragma unroll
      for (int i = 0; i < MBW; ++i) opaque(mb[i]);
#pragma unroll
      for (int k = 0; k < NVL; ++k) opaque(raw[k]);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        float xk[EPV];
        unpack_vec<DT>(raw[k], xk);
#pragma unroll
        for (int c = 0; c < EPV; ++c) {
          const float xv = SCALED ? xk[c] * p.scale : xk[c];
          const uint32_t fill = bit_fill(mb[(k * EPV + c) >> 5], (k * EPV + c) & 31);
          m_msk = fmaxf(m_msk, __uint_as_float((__float_as_uint(xv) & fill) | (0xff800000u & ~fill)));
        }
        opaque(m_msk);
        __builtin_amdgcn_sched_barrier(0);
      }
      m_msk = wave_max(m_msk);
      if (lane == 0) lds_write_b32(scr_max + 64 + wave * 4, __float_as_uint(m_msk));
      lds_barrier();
      m_msk = row16_max_bcast(__uint_as_float(lds_read_b32_wait(scr_max + 64 + wr * 4)));
      if (!(m_msk > kNegInf)) break;  // nothing allowed: S_msk is 0
      N_msk = __builtin_rintf(m_msk * kLog2e);
      Nb_cur = N_msk + (float)kFixShift;
    }

    // ---- phase 3 is not here: the sums and exponents go to the workspace and finish_kernel /
    //      locate_kernel turn them into lse / logZ (double-precision log) and the token.  In-kernel it
    //      cost 1.2k cycles per row on one lane and its hoisted double constants spilled.
    if (tid == 0) {
      p.row_sums[2 * cur.pidx] = S_all;
      p.row_sums[2 * cur.pidx + 1] = S_msk;
      p.row_exps[2 * cur.pidx] = N_all;
      p.row_exps[2 * cur.pidx + 1] = N_msk;
    }

    GLB_STAMP();
    if (!has_next) break;
    // ---- next row: everything prefetched has landed ---------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (kBits) lds_barrier();  // the mask row was DMA'd by other waves
    GLB_STAMP();
    cur = nxr;
    vb = vb_next;
    land(cur);
    GLB_STAMP();
  }
  // ---- tail: lse / logZ (and the token) of every row this workgroup streamed, one wave per row ----------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's workspace stores have reached L2
  lds_barrier();
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
