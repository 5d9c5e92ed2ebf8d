// glb_api.hip — C ABI of libglb_hip.so (include/glb.h): argument validation, launch-geometry
// selection, the small bookkeeping kernels of the hot path, and the host-side RNG helpers.
// gfx950 only.  No entry point allocates, frees or synchronises.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/glb.h"
#include "glb_chunk.hpp"
#include "glb_common.hpp"

namespace glb {
// launchers exported by the three glb_chunk_tu.hip translation units (one per element type)
#define GLB_DECL(dt)                                                                                              \
  hipError_t launch_stats_##dt(const StepParams &p, int mask_kind, bool scaled, int expc, hipStream_t s);          \
  hipError_t launch_fused_step_##dt(const StepParams &p, int mask_kind, int mode, bool scaled, int expc,          \
                                    hipStream_t s);                                                               \
  hipError_t launch_finish_##dt(const StepParams &p, int mask_kind, int mode, int expc, hipStream_t s);           \
  hipError_t launch_logprob_rows_##dt(const void *logits, int64_t ld, int V, float scale, const float *lse,       \
                                      void *out, bool out16, int64_t out_ld, int n_rows, hipStream_t s);           \
  hipError_t launch_logprob_waves_##dt(const void *logits, int64_t ld, int V, int nch, float scale, void *out,    \
                                       int64_t out_ld, float *out_lse, int n_rows, uint64_t *recs, uint32_t epoch,  \
                                       uint32_t *err, uint64_t spin_ticks, bool out16, int variant, hipStream_t s);
GLB_DECL(0) GLB_DECL(1) GLB_DECL(2)
#undef GLB_DECL
thread_local hipEvent_t g_step_ev_start = nullptr, g_step_ev_stop = nullptr;  // glb_logprob_mask_sample_timed
int g_lds_pad[3] = {0, 0, 0};  // dynamic LDS bytes per one-wave workgroup of the fused step, by element type (occupancy cap)
}  // namespace glb

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  return fail(GLB_EHIP, "%s: %s", what, hipGetErrorString(e));
}
}  // namespace

namespace glb {  // the error slot for the library's other translation units (glb_common.hpp)
int api_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int api_hip_fail(hipError_t e, const char *what) { return hip_fail(e, what); }
}  // namespace glb

namespace {

// expc: the term's arithmetic contract (glb::kExpPoly / glb::kExpHw, glb_math.hpp)
hipError_t launch_stats(int dtype, const glb::StepParams &p, int mask_kind, bool scaled, int expc, hipStream_t s) {
  switch (dtype) {
    case 0: return glb::launch_stats_0(p, mask_kind, scaled, expc, s);
    case 1: return glb::launch_stats_1(p, mask_kind, scaled, expc, s);
    case 2: return glb::launch_stats_2(p, mask_kind, scaled, expc, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_fused_step(int dtype, const glb::StepParams &p, int mask_kind, int mode, bool scaled, int expc, hipStream_t s) {
  switch (dtype) {
    case 0: return glb::launch_fused_step_0(p, mask_kind, mode, scaled, expc, s);
    case 1: return glb::launch_fused_step_1(p, mask_kind, mode, scaled, expc, s);
    case 2: return glb::launch_fused_step_2(p, mask_kind, mode, scaled, expc, s);
  }
  return hipErrorInvalidValue;
}

// Step workspaces that glb_workspace_init has zeroed: the one-launch step tags its records with a per-workspace epoch
// (1, 2, ...), so a stale record can never pass for a fresh one.  Everything else about a workspace stays the caller's.
struct WsEntry {
  uint32_t epoch;
  size_t bytes;
};
// the last 64 bytes of a registered workspace hold its error word: waves of a one-launch call that gave up waiting for
// their records add 1 to it (glb_workspace_check reads and clears it).  Callers size workspaces with the *_bytes
// functions, which leave 256 bytes beyond what the calls use.
constexpr size_t kWsErrTail = 64;
std::atomic<uint64_t> g_spin_ticks{glb::kSpinTicks};
std::mutex g_ws_mu;
std::unordered_map<const void *, WsEntry> g_ws;

int device_cus() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 64;
    }
    return n;
  }();
  return cus;
}

// finishing waves a one-launch step may have: half of the wave slots the device holds at the kernels' occupancy (4
// waves per SIMD; 3 with float masks), so that they can never keep the stats waves they wait for from a slot
// The next epoch of a registered workspace for a launch that tags its records: 1 and *epoch set when `workspace` was
// initialised with at least `bytes` and `s` is not being captured (the epoch is a launch argument: a replayed graph would
// reuse it); 0 when the caller has to take its untagged form; -1 on a HIP error.
int ws_next_epoch(void *workspace, size_t bytes, size_t used, hipStream_t s, uint32_t *epoch, uint32_t **err) {
  *err = nullptr;  // bytes: the tagged records at the workspace's start; used: everything the call touches
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (cs != hipStreamCaptureStatusNone) return 0;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto it = g_ws.find(workspace);
  if (it == g_ws.end() || it->second.bytes < bytes) return 0;
  if (++it->second.epoch == 0u) {  // 2^32 calls on this workspace: start the tags over
    if (hipMemsetAsync(workspace, 0, bytes, s) != hipSuccess) return -1;
    it->second.epoch = 1u;
  }
  *epoch = it->second.epoch;
  if (it->second.bytes >= used + kWsErrTail) *err = (uint32_t *)((char *)workspace + it->second.bytes - kWsErrTail);
  return 1;
}

// wave slots of the device at `waves_per_simd` (one-wave workgroups)
int wave_slots(int waves_per_simd) { return device_cus() * 4 * waves_per_simd; }

// which log-softmax kernel: -1 = the best the call allows (independent waves when the workspace carries tags, else a
// workgroup per row or three launches).  The diagnostic build takes it from the environment for same-box comparisons:
// 0 = never the independent waves.
int lsm_mode() {
#ifdef GLB_STAMPS
  if (const char *e = getenv("GLB_LSM_GRID")) return atoi(e);
#endif
  return -1;
}

int lsm_variant() {
#ifdef GLB_STAMPS
  if (const char *e = getenv("GLB_LSM_VARIANT")) return atoi(e);
#endif
  return 0;
}

int fin_wave_cap(bool float_mask) { return device_cus() * (float_mask ? 6 : 8); }



hipError_t launch_finish(int dtype, const glb::StepParams &p, int mask_kind, int mode, int expc, hipStream_t s) {
  switch (dtype) {
    case 0: return glb::launch_finish_0(p, mask_kind, mode, expc, s);
    case 1: return glb::launch_finish_1(p, mask_kind, mode, expc, s);
    case 2: return glb::launch_finish_2(p, mask_kind, mode, expc, s);
  }
  return hipErrorInvalidValue;
}

inline int64_t n_chunks(int64_t vocab) { return (vocab + glb::kChunk - 1) / glb::kChunk; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// prepared masks: transposed lane words, then one "allows anything" word per (mask, chunk)
inline size_t prepared_words_bytes(int64_t n_masks, int64_t vocab) {
  return align256((size_t)n_masks * (size_t)n_chunks(vocab) * 64 * sizeof(uint64_t));
}
inline size_t prepared_bytes(int64_t n_masks, int64_t vocab) {
  return prepared_words_bytes(n_masks, vocab) + align256((size_t)n_masks * (size_t)n_chunks(vocab) * sizeof(uint64_t));
}

// step workspace: chunk records (64 bytes per unit and chunk); then prepared masks
inline size_t step_recs_bytes(int64_t units, int64_t vocab) {
  return align256((size_t)units * (size_t)n_chunks(vocab) * glb::kRecWords * sizeof(uint64_t));
}
inline size_t step_fixed_bytes(int64_t units, int64_t vocab) { return step_recs_bytes(units, vocab); }

// start: an event the launch carries as its start stamp (glb_logprob_mask_sample_timed: a call that prepares its own masks
// is timed from this launch on)
hipError_t launch_mask_prepare(const uint32_t *bits, int64_t n_masks, int64_t vocab, int64_t mask_ld, int dtype,
                               void *out, hipStream_t s, hipEvent_t start = nullptr, const int32_t *which = nullptr,
                               int64_t n_which = 0) {
  const int nch = (int)n_chunks(vocab);
  uint64_t *mt = (uint64_t *)out;
  uint64_t *many = (uint64_t *)((char *)out + prepared_words_bytes(n_masks, vocab));
  const dim3 grid((unsigned)nch, (unsigned)(which ? n_which : n_masks)), block(64);
  if (dtype == GLB_F32)
    hipExtLaunchKernelGGL((glb::mask_prepare_kernel<4>), grid, block, 0, s, start, nullptr, 0, bits, mask_ld, (int)vocab, nch, mt, many,
                          which, (int)n_masks);
  else
    hipExtLaunchKernelGGL((glb::mask_prepare_kernel<8>), grid, block, 0, s, start, nullptr, 0, bits, mask_ld, (int)vocab, nch, mt, many,
                          which, (int)n_masks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// one thread per output word: bit = (mask == 0); flags any value that is neither 0 nor -inf
__global__ void mask_to_bits_kernel(const float *mask, int64_t n_masks, int64_t V, int64_t mask_ld,
                                    uint32_t *bits, int64_t bits_ld, int32_t *nonbinary) {
  const int64_t words = bits_ld;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_masks * words) return;
  const int64_t k = gid / words, w = gid % words;
  const float *row = mask + k * mask_ld;
  uint32_t out = 0;
  bool nb = false;
  for (int b = 0; b < 32; ++b) {
    const int64_t j = w * 32 + b;
    if (j < V) {
      const float v = row[j];
      if (v == 0.0f) out |= 1u << b;
      else if (!(v == glb::kNegInf)) nb = true;
    }
  }
  bits[k * bits_ld + w] = out;
  if (nb && nonbinary) *nonbinary = 1;
}

__device__ inline bool same_ctx(const int32_t *tok, const int64_t *st, const int32_t *len, int64_t i,
                                int64_t j) {
  const int64_t li = len[i];
  if (len[j] != li) return false;
  const int32_t *a = tok + st[i], *b = tok + st[j];
  for (int64_t t = 0; t < li; ++t)
    if (a[t] != b[t]) return false;
  return true;
}

__device__ inline uint64_t ctx_hash_step(uint64_t h, uint32_t v) {
  h ^= v;
  h *= 0x100000001b3ull;
  return h ^ (h >> 29);
}

// Hash of every context, one thread each, as many workgroups as it takes.  What bounds context hashing is cache lines
// touched per load instruction (every lane reads another context: 64 lines per instruction whatever the width), so
// four tokens come in one 16-byte load (element-aligned only, like the logits rows) and - the point of a kernel of
// its own - the lines are spread over many CUs' L1s instead of one.
__global__ __launch_bounds__(256) void hash_contexts_kernel(const int32_t *tok, const int64_t *st, const int32_t *len,
                                                             int32_t n, uint64_t *out_hash) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t li = len[i];
  const int32_t *t = tok + st[i];
  uint64_t h = 0xcbf29ce484222325ull;  // (the length is compared on its own: a context's hash extends token by token)
  int j = 0;
  for (; j + 4 <= li; j += 4) {
    const glb::u32x4_t w = *reinterpret_cast<const glb::u32x4_t *>(t + j);
    h = ctx_hash_step(ctx_hash_step(ctx_hash_step(ctx_hash_step(h, w.x), w.y), w.z), w.w);
  }
  for (; j < li; ++j) h = ctx_hash_step(h, (uint32_t)t[j]);
  out_hash[i] = h;
}

// token comparison with 16-byte loads (used after the hashes and lengths were found equal: true duplicates)
__device__ inline bool same_tokens(const int32_t *a, const int32_t *b, int32_t li) {
  int j = 0;
  for (; j + 4 <= li; j += 4) {
    const glb::u32x4_t x = *reinterpret_cast<const glb::u32x4_t *>(a + j), y = *reinterpret_cast<const glb::u32x4_t *>(b + j);
    if (((x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w)) != 0u) return false;
  }
  for (; j < li; ++j)
    if (a[j] != b[j]) return false;
  return true;
}

// ---- populations above 8192 contexts: the table lives in the workspace and every phase but the numbering runs on as
// many workgroups as it takes.  Who wins a slot depends on timing; the result does not (a group's representative is the
// minimum index that reached its slot, ids follow first appearance).
__global__ __launch_bounds__(256) void group_init_kernel(int32_t cap, int32_t *table, int32_t *minidx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    table[i] = -1;
    minidx[i] = 0x7fffffff;
  }
}

__global__ __launch_bounds__(256) void group_insert_kernel(const int32_t *tok, const int64_t *st, const int32_t *len,
                                                            int32_t n, int32_t cap, const uint64_t *hashes,
                                                            int32_t *table, int32_t *minidx, int32_t *slot_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = hashes[i];
  const int32_t li = len[i];
  const int64_t si = st[i];
  const int32_t *mine = tok + si;
  // A wave whose contexts are all the same one (SIS step 0, a freshly resampled population) sends its first lane
  // alone: 16 384 CAS + atomicMin on ONE address are otherwise 16 384 serialised round trips (0.4 ms).
  const uint32_t h0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)h);
  const uint32_t h0h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(h >> 32));
  const int32_t l0 = __builtin_amdgcn_readfirstlane(li);
  const uint32_t s0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)si);
  const uint32_t s0h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)si >> 32));
  const int64_t si0 = (int64_t)(((uint64_t)s0h << 32) | s0l);
  const bool like_first = h == (((uint64_t)h0h << 32) | h0l) && li == l0 && (si == si0 || same_tokens(mine, tok + si0, li));
  const uint64_t act = __ballot(true);
  const bool all_same = __ballot(like_first) == act;
  const int leader = __ffsll((long long)act) - 1, lane = threadIdx.x & 63;
  int32_t slot = (int32_t)(h & (uint64_t)(cap - 1));
  if (!all_same || lane == leader) {
    for (;;) {
      const int32_t owner = atomicCAS(&table[slot], -1, i);
      if (owner == -1 || owner == i) break;
      if (hashes[owner] == h && len[owner] == li && same_tokens(mine, tok + st[owner], li)) break;
      slot = (slot + 1) & (cap - 1);
    }
    atomicMin(&minidx[slot], i);  // the leader has the wave's smallest index
  }
  if (all_same) slot = __builtin_amdgcn_readlane(slot, leader);
  slot_of[i] = slot;
}

// group ids in first-appearance order: one workgroup scans the representative flags tile by tile
__global__ __launch_bounds__(1024) void group_number_kernel(int32_t n, const int32_t *minidx, const int32_t *slot_of,
                                                             int32_t *gid_of, int32_t *out_group_of, int32_t *out_rep,
                                                             int32_t *out_n_groups) {
  const int T = 1024, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int32_t s_wave[16];
  __shared__ int32_t s_carry;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += T) {
    const int i = base + tid;
    const int32_t flag = (i < n && minidx[slot_of[i]] == i) ? 1 : 0;
    int32_t incl = flag;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int32_t v = s_wave[w];
      if (w < wave) wbase += v;
      total += v;
    }
    const int32_t carry = s_carry;
    if (flag) {
      const int32_t g = carry + wbase + incl - 1;
      gid_of[i] = g;
      out_rep[g] = i;
    }
    __syncthreads();
    if (tid == 0) s_carry = carry + total;
    __syncthreads();
  }
  if (tid == 0) *out_n_groups = s_carry;
  for (int i = tid; i < n; i += T) out_group_of[i] = gid_of[minidx[slot_of[i]]];
}

// Exact dedup with the hash table in LDS (n <= 8192 contexts: every launch of the BASELINE configurations): groups
// numbered by first appearance, representative = smallest index; no global-memory atomics and no table in the
// workspace to clear: each thread keeps its contexts' slots in registers, and the only global traffic besides the
// tokens (or their hashes) is the output.  One workgroup: the whole job is a handful of memory latencies.
// HASHED: the hashes come from hash_contexts_kernel (populations above 1024: the table phase then touches tokens only
// to confirm true duplicates); otherwise the kernel hashes its contexts itself (one launch for small populations).
template <int PER, bool HASHED>
__global__ __launch_bounds__(1024) void group_contexts_lds_kernel(const int32_t *tok, const int64_t *st,
                                                                   const int32_t *len, int32_t n, int32_t cap,
                                                                   int32_t *out_group_of, int32_t *out_rep,
                                                                   int32_t *out_n_groups, const uint64_t *hash_in) {
  extern __shared__ __attribute__((aligned(16))) int32_t s_dyn[];
  int32_t *s_table = s_dyn, *s_min = s_dyn + cap;
  // !HASHED (up to 2048 contexts, one launch): every context's hash, start and length also sit in LDS, so that a
  // collision or a duplicate is recognised without going back to global memory for anything but the tokens
  uint64_t *s_hash = reinterpret_cast<uint64_t *>(s_dyn + 2 * cap);
  int64_t *s_st = reinterpret_cast<int64_t *>(s_hash + 1024 * PER);
  int32_t *s_len = reinterpret_cast<int32_t *>(s_st + 1024 * PER);
  __shared__ int32_t s_wave[16];
  const int T = 1024, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < cap; i += T) {
    s_table[i] = -1;
    s_min[i] = 0x7fffffff;
  }
  // thread t owns the PER consecutive contexts t*PER ..: one block scan then numbers the groups in index order
  int32_t slot[PER], li[PER];
  int64_t si[PER];
  uint64_t h[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    li[k] = i < n ? len[i] : 0;
    si[k] = i < n ? st[i] : 0;
    h[k] = 0xcbf29ce484222325ull;
  }
  if constexpr (HASHED) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid * PER + k;
      if (i < n) h[k] = hash_in[i];
    }
  } else {
    // four tokens per 16-byte load (element-aligned only, like the logits rows): what bounds this phase is cache lines
    // touched per load instruction on this one CU's L1 - every lane reads another context - and a 16-byte load touches
    // no more lines than a 4-byte one
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int32_t *t = tok + si[k];
      int j = 0;
      for (; j + 4 <= li[k]; j += 4) {
        const glb::u32x4_t w = *reinterpret_cast<const glb::u32x4_t *>(t + j);
        h[k] = ctx_hash_step(ctx_hash_step(ctx_hash_step(ctx_hash_step(h[k], w.x), w.y), w.z), w.w);
      }
      for (; j < li[k]; ++j) h[k] = ctx_hash_step(h[k], (uint32_t)t[j]);
      const int i = tid * PER + k;
      s_hash[i] = h[k];
      s_st[i] = si[k];
      s_len[i] = li[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    slot[k] = 0;
    // A wave whose contexts are all the same one (SIS step 0, a freshly resampled population) sends its first lane
    // alone: 64 CAS + atomicMin on ONE LDS word are otherwise 64 serialised turns per wave.
    bool all_same = false;
    int leader = 0;
    {
      const uint64_t act = __ballot(i < n);
      if (act != 0ull) {
        leader = __ffsll((long long)act) - 1;
        const uint32_t h0l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h[k], leader);
        const uint32_t h0h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h[k] >> 32), leader);
        const int32_t l0 = __builtin_amdgcn_readlane(li[k], leader);
        const uint32_t s0l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)si[k], leader);
        const uint32_t s0h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)si[k] >> 32), leader);
        const int64_t si0 = (int64_t)(((uint64_t)s0h << 32) | s0l);
        const bool like = i < n && h[k] == (((uint64_t)h0h << 32) | h0l) && li[k] == l0 &&
                          (si[k] == si0 || same_tokens(tok + si[k], tok + si0, li[k]));
        all_same = __ballot(like) == act;
      }
    }
    if (i < n && (!all_same || lane == leader)) {
      int32_t s = (int32_t)(h[k] & (uint64_t)(cap - 1));
      for (;;) {
        const int32_t owner = atomicCAS(&s_table[s], -1, i);
        if (owner == -1 || owner == i) break;
        if constexpr (HASHED) {
          if (hash_in[owner] == h[k] && len[owner] == li[k] && same_tokens(tok + si[k], tok + st[owner], li[k])) break;
        } else {
          if (s_hash[owner] == h[k] && s_len[owner] == li[k] && same_tokens(tok + si[k], tok + s_st[owner], li[k])) break;
        }
        s = (s + 1) & (cap - 1);
      }
      slot[k] = s;
      atomicMin(&s_min[s], i);  // (the leader has the wave's smallest index of this k)
    }
    if (all_same) slot[k] = __builtin_amdgcn_readlane(slot[k], leader);
  }
  __syncthreads();
  // group ids in first-appearance order: exclusive scan of the representative flags in index order
  int32_t flag[PER], mine = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    flag[k] = (i < n && s_min[slot[k]] == i) ? 1 : 0;
    mine += flag[k];
  }
  int32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int32_t g = incl - mine, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int32_t v = s_wave[w];
    if (w < wave) g += v;
    total += v;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (flag[k]) {
      s_table[slot[k]] = g;  // the slot's owner is not needed any more: it now names the group
      out_rep[g] = tid * PER + k;
      ++g;
    }
  }
  if (tid == 0) *out_n_groups = total;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    if (i < n) out_group_of[i] = s_table[slot[k]];
  }
}

__global__ void match_prefixes_kernel(const int32_t *tok, const int64_t *st, const int32_t *len,
                                      int64_t n, const int32_t *ptok, const int64_t *pst,
                                      const int32_t *plen, int64_t np, int32_t *out_prefix,
                                      int32_t *out_base) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t li = len[i];
  const int32_t *c = tok + st[i];
  int32_t best = -1;
  int64_t bl = 0;
  for (int64_t k = 0; k < np; ++k) {
    const int64_t lk = plen[k];
    if (lk >= li || lk <= bl) continue;
    const int32_t *q = ptok + pst[k];
    bool eq = true;
    for (int64_t t = 0; t < lk; ++t)
      if (q[t] != c[t]) {
        eq = false;
        break;
      }
    if (eq) {
      best = (int32_t)k;
      bl = lk;
    }
  }
  out_prefix[i] = best;
  out_base[i] = (int32_t)bl;
}

// one thread per attention-mask column (the widest output); ids / positions share the tail columns
__global__ void gather_padded_kernel(const int32_t *tok, const int64_t *st, const int32_t *lens,
                                     const int32_t *sel, int64_t n_sel, const int32_t *base, int64_t pad_id,
                                     int64_t p_max, int64_t l_max, int64_t *ids, int64_t *am,
                                     int64_t *pos, int32_t *last) {
  const int64_t width = p_max + l_max;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_sel * width) return;
  const int64_t u = gid / width, p = gid % width;
  const int64_t s = sel ? sel[u] : u;
  const int64_t b = base ? base[s] : 0;
  int64_t len = (int64_t)lens[s] - b;
  len = len < 0 ? 0 : (len > l_max ? l_max : len);
  if (p < p_max) {
    am[gid] = p < b ? 1 : 0;
  } else {
    const int64_t t = p - p_max;
    const bool in = t < len;
    am[gid] = in ? 1 : 0;
    ids[u * l_max + t] = in ? (int64_t)tok[st[s] + b + t] : pad_id;
    pos[u * l_max + t] = in ? b + t : 0;
    if (t == 0 && last) last[u] = (int32_t)(len - 1);
  }
}

// 16-byte (or elem-wide fallback) copies of prefix KV slabs into the zero-padded batch tensor
template <typename VT>
__global__ void gather_kv_kernel(const void *const *slabs, const int32_t *slab_len,
                                 const int32_t *prefix_of, int64_t n_rows, int64_t heads,
                                 int64_t row_vecs, int64_t p_max, VT *out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = n_rows * heads * p_max * row_vecs;
  if (gid >= total) return;
  const int64_t x = gid % row_vecs;
  const int64_t pp = (gid / row_vecs) % p_max;
  const int64_t h = (gid / (row_vecs * p_max)) % heads;
  const int64_t u = gid / (row_vecs * p_max * heads);
  const int32_t k = prefix_of[u];
  VT v{};
  if (k >= 0) {
    const int64_t pl = slab_len[k];
    if (pp < pl) v = reinterpret_cast<const VT *>(slabs[k])[(h * pl + pp) * row_vecs + x];
  }
  out[gid] = v;
}

__global__ void particles_advance_kernel(int32_t *ctx, int64_t ctx_ld, int32_t *len,
                                         int32_t *active, float *lw, const float *logZ,
                                         const int32_t *tok, int64_t n, int32_t eos,
                                         int32_t max_len, uint64_t *hashes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !active[i]) return;
  const int32_t t = tok[i];
  if (t == -2) return;  // a failed launch (include/glb.h: out_token), never a result: the particle stays as it was
  lw[i] += logZ[i];
  if (t == eos || t < 0) {
    active[i] = 0;
  } else {
    const int32_t l = len[i];
    ctx[i * ctx_ld + l] = t;
    len[i] = l + 1;
    if (hashes) hashes[i] = ctx_hash_step(hashes[i], (uint32_t)t);  // the context's hash grows with it
    if (l + 1 >= max_len) active[i] = 0;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char *glb_version(void) { return "glb-hip 0.2.0 (gfx950)"; }
int glb_abi_version(void) { return GLB_ABI_VERSION; }

int glb_last_error(char *buf, size_t n) {
  if (!buf || n == 0) return GLB_EINVAL;
  snprintf(buf, n, "%s", g_err.c_str());
  return GLB_OK;
}

int glb_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return c;
}

size_t glb_step_workspace_bytes(int64_t n_particles, int64_t n_rows, int64_t vocab, int64_t n_masks) {
  if (n_particles <= 0 || vocab <= 0) return 0;
  const int64_t units = n_particles > n_rows ? n_particles : n_rows;
  return step_fixed_bytes(units, vocab) + (n_masks > 0 ? prepared_bytes(n_masks, vocab) : 0) + 256;
}

size_t glb_mask_prepared_bytes(int64_t n_masks, int64_t vocab) {
  if (n_masks <= 0 || vocab <= 0) return 0;
  return prepared_bytes(n_masks, vocab);
}

int glb_mask_prepare(const uint32_t *mask_bits, int64_t n_masks, int64_t vocab, int64_t mask_ld, int32_t dtype,
                     void *out_prepared, size_t out_bytes, void *stream) {
  if (!mask_bits || !out_prepared) return fail(GLB_EINVAL, "null pointer");
  if (n_masks <= 0 || vocab <= 0 || mask_ld < (vocab + 31) / 32) return fail(GLB_EINVAL, "bad sizes");
  if (dtype < GLB_F32 || dtype > GLB_F16) return fail(GLB_EINVAL, "bad dtype %d", dtype);
  if (vocab > 0x7fffff00ll || n_masks > 65535) return fail(GLB_EINVAL, "vocab / n_masks out of range");
  if (out_bytes < prepared_bytes(n_masks, vocab))
    return fail(GLB_ENOSPC, "prepared buffer %zu < %zu bytes", out_bytes, prepared_bytes(n_masks, vocab));
  if (((uintptr_t)out_prepared) % 16 || ((uintptr_t)mask_bits) % 4) return fail(GLB_EINVAL, "misaligned pointer");
  const hipError_t e = launch_mask_prepare(mask_bits, n_masks, vocab, mask_ld, dtype, out_prepared, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "mask_prepare launch");
  return GLB_OK;
}

int glb_mask_prepare_rows(const uint32_t *mask_bits, int64_t n_masks, int64_t vocab, int64_t mask_ld, int32_t dtype,
                          const int32_t *rows, int64_t n_rows, void *prepared, size_t prepared_bytes_, void *stream) {
  if (!mask_bits || !prepared || (n_rows > 0 && !rows)) return fail(GLB_EINVAL, "null pointer");
  if (n_masks <= 0 || vocab <= 0 || mask_ld < (vocab + 31) / 32 || n_rows < 0) return fail(GLB_EINVAL, "bad sizes");
  if (dtype < GLB_F32 || dtype > GLB_F16) return fail(GLB_EINVAL, "bad dtype %d", dtype);
  if (vocab > 0x7fffff00ll || n_masks > 65535 || n_rows > 65535) return fail(GLB_EINVAL, "vocab / n_masks / n_rows out of range");
  if (prepared_bytes_ < prepared_bytes(n_masks, vocab))
    return fail(GLB_ENOSPC, "prepared buffer %zu < %zu bytes", prepared_bytes_, prepared_bytes(n_masks, vocab));
  if (((uintptr_t)prepared) % 16 || ((uintptr_t)mask_bits) % 4) return fail(GLB_EINVAL, "misaligned pointer");
  if (n_rows == 0) return GLB_OK;
  const hipError_t e = launch_mask_prepare(mask_bits, n_masks, vocab, mask_ld, dtype, prepared, (hipStream_t)stream, nullptr, rows, n_rows);
  if (e != hipSuccess) return hip_fail(e, "mask_prepare launch");
  return GLB_OK;
}

int glb_logprob_mask_sample(const glb_step_args *a, void *stream) {
  if (!a) return fail(GLB_EINVAL, "glb_logprob_mask_sample: null args");
  if (a->struct_size != sizeof(glb_step_args))
    return fail(GLB_EINVAL, "glb_step_args.struct_size %u != %zu (ABI mismatch)", a->struct_size,
                sizeof(glb_step_args));
  if (!a->logits) return fail(GLB_EINVAL, "logits is null");
  if (a->dtype < GLB_F32 || a->dtype > GLB_F16) return fail(GLB_EINVAL, "bad dtype %d", a->dtype);
  if (a->n_rows <= 0 || a->vocab <= 0 || a->n_particles <= 0)
    return fail(GLB_EINVAL, "n_rows=%lld vocab=%lld n_particles=%lld must be positive",
                (long long)a->n_rows, (long long)a->vocab, (long long)a->n_particles);
  if (a->ld < a->vocab) return fail(GLB_EINVAL, "ld %lld < vocab %lld", (long long)a->ld, (long long)a->vocab);
  if (a->vocab > 0x7fffff00ll || a->n_particles > 0x7fffffffll || a->n_rows > 0x7fffffffll)
    return fail(GLB_EINVAL, "vocab / n_particles / n_rows exceed 31 bits");
  const int es = a->dtype == GLB_F32 ? 4 : 2;
  if (((uintptr_t)a->logits) % es) return fail(GLB_EINVAL, "logits pointer not element aligned");
  if (!a->row_of && a->n_particles != a->n_rows)
    return fail(GLB_EINVAL, "row_of is null but n_particles != n_rows");
  if (a->mask_kind < GLB_MASK_NONE || a->mask_kind > GLB_MASK_PREPARED)
    return fail(GLB_EINVAL, "bad mask_kind %d", a->mask_kind);
  const bool by_row = a->mask_kind == GLB_MASK_NONE || a->row_mask_id != nullptr;  // reduce once per logits row
  const int64_t n_units = by_row ? a->n_rows : a->n_particles;
  if (a->mask_kind != GLB_MASK_NONE) {
    if (!a->mask || a->n_masks <= 0) return fail(GLB_EINVAL, "mask table missing");
    if (a->mask_kind != GLB_MASK_PREPARED) {
      const int64_t need = a->mask_kind == GLB_MASK_BITS ? (a->vocab + 31) / 32 : a->vocab;
      if (a->mask_ld < need) return fail(GLB_EINVAL, "mask_ld %lld < %lld", (long long)a->mask_ld, (long long)need);
    }
    if (a->row_mask_id && a->mask_id) return fail(GLB_EINVAL, "give mask_id (per particle) or row_mask_id (per row), not both");
    if (!a->row_mask_id && !a->mask_id && a->n_masks != 1 && a->n_masks != n_units)
      return fail(GLB_EINVAL, "no mask ids given but n_masks is neither 1 nor the number of %s",
                  by_row ? "rows" : "particles");
    if (((uintptr_t)a->mask) % (a->mask_kind == GLB_MASK_PREPARED ? 16 : 4)) return fail(GLB_EINVAL, "mask pointer misaligned");
    if (a->n_masks > 0x7fffffffll) return fail(GLB_EINVAL, "n_masks exceeds 31 bits");
  }
  if (a->rng_mode < GLB_RNG_NONE || a->rng_mode > GLB_RNG_NOISE)
    return fail(GLB_EINVAL, "bad rng_mode %d", a->rng_mode);
  if (a->flags & ~GLB_STEP_HW_EXP) return fail(GLB_EINVAL, "flags %d: only GLB_STEP_HW_EXP is defined", a->flags);
  const int expc = (a->flags & GLB_STEP_HW_EXP) ? glb::kExpHw : glb::kExpPoly;
  if (a->rng_mode == GLB_RNG_NOISE && (!a->noise || (a->noise_ld < a->vocab && a->noise_ld != 0)))
    return fail(GLB_EINVAL, "noise tensor missing or noise_ld < vocab (0 = one row shared by every particle)");
  if (a->rng_mode != GLB_RNG_NONE && !a->out_token)
    return fail(GLB_EINVAL, "rng_mode set but out_token is null");
  if (!(a->logit_scale == a->logit_scale)) return fail(GLB_EINVAL, "logit_scale is NaN");
  if (!a->workspace) return fail(GLB_EINVAL, "workspace is null (glb_step_workspace_bytes)");
  if (((uintptr_t)a->workspace) % 32) return fail(GLB_EINVAL, "workspace not 32-byte aligned");
  const bool own_prep = a->mask_kind == GLB_MASK_BITS;
  const size_t fixed_bytes = step_fixed_bytes(n_units, a->vocab);
  const size_t need_ws = fixed_bytes + (own_prep ? prepared_bytes(a->n_masks, a->vocab) : 0);
  if (a->workspace_bytes < need_ws)
    return fail(GLB_ENOSPC, "workspace %zu < %zu bytes", a->workspace_bytes, need_ws);

  hipStream_t s = (hipStream_t)stream;
  glb::StepParams p{};
  p.logits = a->logits;
  p.ld = a->ld;
  p.V = (int32_t)a->vocab;
  p.nch = (int32_t)n_chunks(a->vocab);
  p.scale = a->logit_scale;
  p.n_particles = (int32_t)a->n_particles;
  p.n_pairs = (int32_t)n_units;
  p.n_masks = (int32_t)a->n_masks;
  if (by_row) {
    p.pair_row = nullptr;
    p.pair_mask = a->row_mask_id;
    p.pair_of = a->row_of;
  } else {
    p.pair_row = a->row_of;
    p.pair_mask = a->mask_id;
    p.pair_of = nullptr;
  }
  p.recs = (uint64_t *)a->workspace;
  p.noise = a->noise;
  p.noise_ld = a->noise_ld;
  p.seed = a->seed;
  p.offset = a->offset;
  p.particle_base = a->particle_base;
  p.out_logZ = a->out_logZ;
  p.out_lse = a->out_lse;
  p.out_token = a->out_token;
#ifdef GLB_STAMPS
  p.out_margin = a->out_margin;
#else
  p.out_margin = a->rng_mode == GLB_RNG_NOISE ? a->out_margin : nullptr;
#endif
  const bool scaled = a->logit_scale != 1.0f;
  // One launch (stats waves, then finishing waves that sweep the tagged records) when the workspace was initialised,
  // the stream is not being captured (the epoch is a launch argument), the draw is not the parity race and the
  // launch is big enough to be dealt one wave per chunk (smaller ones: four waves per chunk, then the finish launch);
  // two launches otherwise.
  const int64_t items = n_units * (int64_t)p.nch;
  const bool fmask = a->mask_kind == GLB_MASK_F32;
  bool fused = a->rng_mode != GLB_RNG_NOISE && items > 512 && a->n_particles <= 16 * (int64_t)fin_wave_cap(fmask);
  p.spin_ticks = g_spin_ticks.load(std::memory_order_relaxed);
#ifdef GLB_STAMPS  // diagnostic build: occupancy cap of the one-launch step from the environment (bytes of unused LDS)
  if (const char *e = getenv("GLB_LDS_PAD")) glb::g_lds_pad[a->dtype] = atoi(e);
  if (const char *e = getenv("GLB_DBG_MODE")) p.diag.dbg_mode = atoi(e);
#endif
  if (fused) {
    const int rc = ws_next_epoch(a->workspace, fixed_bytes, need_ws, s, &p.epoch, &p.err);
    if (rc < 0) return hip_fail(hipGetLastError(), "workspace re-zero");
    fused = rc > 0;
  }
  int kmask = glb::kMaskNone;
  if (a->mask_kind == GLB_MASK_F32) {
    kmask = glb::kMaskF32;
    p.mask_f = (const float *)a->mask;
    p.mask_ld = a->mask_ld;
  } else if (own_prep && fused) {
    // raw bit rows and the one-launch step: the stats waves read the rows themselves (kMaskRaw) - no mask_prepare launch,
    // no lane words written to the workspace and read back
    kmask = glb::kMaskRaw;
    p.mask_bits = (const uint32_t *)a->mask;
    p.mask_bits_ld = a->mask_ld;
  } else if (a->mask_kind != GLB_MASK_NONE) {
    kmask = glb::kMaskBits;
    const char *prep = (const char *)a->mask;
    if (own_prep) {  // transposed form into the workspace, on the same stream
      char *dst = (char *)a->workspace + fixed_bytes;
      const hipError_t e = launch_mask_prepare((const uint32_t *)a->mask, a->n_masks, a->vocab, a->mask_ld, a->dtype, dst, s,
                                               glb::g_step_ev_start);
      glb::g_step_ev_start = nullptr;  // (a timed call starts with this launch)
      if (e != hipSuccess) return hip_fail(e, "mask_prepare launch");
      prep = dst;
    }
    p.mask_t = (const uint64_t *)prep;
    p.mask_any = (const uint64_t *)(prep + prepared_words_bytes(a->n_masks, a->vocab));
  }
  if (fused) {
    p.stats_blocks = (int32_t)items;
    const int64_t cap = fin_wave_cap(fmask);
    p.fin_blocks = (int32_t)(a->n_particles < cap ? a->n_particles : cap);
#ifdef GLB_STAMPS  // diagnostic build: the placements that were measured and not kept (glb_chunk.hpp: il_lag, short_last)
    if (const char *e = getenv("GLB_FIN_LAG")) {
      const int pct = atoi(e);
      if (pct >= 0 && p.pair_of == nullptr && a->n_particles <= cap) {
        const int wps = fmask ? 3 : (a->dtype == GLB_F32 ? 4 : GLB_STATS_WAVES_16);
        const int64_t lag = ((int64_t)pct * wave_slots(wps) / 100 + p.nch - 1) / p.nch;
        p.diag.il_lag = (int32_t)(lag < n_units ? lag : n_units);
      }
    }
    if (const char *e = getenv("GLB_SHORT_LAST"))
      p.diag.short_last = (atoi(e) && p.diag.il_lag < 0 && p.nch >= 2 && a->vocab - (int64_t)(p.nch - 1) * glb::kChunk <= glb::kChunk / 2) ? 1 : 0;
#endif
    const hipError_t e = launch_fused_step(a->dtype, p, kmask, a->rng_mode, scaled, expc, s);
    if (e != hipSuccess) return hip_fail(e, "fused_step launch");
    return GLB_OK;
  }
  p.epoch = 0u;
  hipError_t e = launch_stats(a->dtype, p, kmask, scaled, expc, s);
  if (e != hipSuccess) return hip_fail(e, "chunk_stats launch");
  e = launch_finish(a->dtype, p, kmask, a->rng_mode, expc, s);
  if (e != hipSuccess) return hip_fail(e, "finish launch");
  return GLB_OK;
}

int glb_logprob_mask_sample_timed(const glb_step_args *a, void *stream, void *start_event, void *stop_event) {
  glb::g_step_ev_start = (hipEvent_t)start_event;
  glb::g_step_ev_stop = (hipEvent_t)stop_event;
  const int rc = glb_logprob_mask_sample(a, stream);
  glb::g_step_ev_start = glb::g_step_ev_stop = nullptr;
  return rc;
}

int glb_workspace_init(void *workspace, size_t workspace_bytes, void *stream) {
  if (!workspace || workspace_bytes == 0) return fail(GLB_EINVAL, "workspace is null or empty");
  if (((uintptr_t)workspace) % 32) return fail(GLB_EINVAL, "workspace not 32-byte aligned");
  const hipError_t e = hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "workspace zero");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  g_ws[workspace] = WsEntry{0u, workspace_bytes};
  return GLB_OK;
}

int glb_workspace_check(void *workspace, void *stream) {
  if (!workspace) return fail(GLB_EINVAL, "workspace is null");
  uint32_t *err = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(workspace);
    if (it == g_ws.end() || it->second.bytes < kWsErrTail) return fail(GLB_EINVAL, "workspace was not registered (glb_workspace_init)");
    err = (uint32_t *)((char *)workspace + it->second.bytes - kWsErrTail);
  }
  uint32_t n = 0;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(&n, err, sizeof n, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return hip_fail(e, "workspace check");
  if (n == 0) return GLB_OK;
  e = hipMemsetAsync(err, 0, sizeof n, s);
  if (e != hipSuccess) return hip_fail(e, "workspace check reset");
  return fail(GLB_EHIP, "%u wave(s) of a one-launch call on this workspace gave up waiting for their row's records "
              "(token -2 / NaN outputs): the launch did not complete, its results must not be used", n);
}

const uint32_t *glb_workspace_error_word(void *workspace) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto it = g_ws.find(workspace);
  if (it == g_ws.end() || it->second.bytes < kWsErrTail) return nullptr;
  return (const uint32_t *)((char *)workspace + it->second.bytes - kWsErrTail);
}

int glb_set_spin_limit(uint64_t microseconds) {
  const uint64_t ticks = microseconds == GLB_SPIN_NONE ? 0ull : (microseconds > (1ull << 40) ? (1ull << 47) : microseconds * 100ull);
  g_spin_ticks.store(microseconds ? ticks : glb::kSpinTicks, std::memory_order_relaxed);
  return GLB_OK;
}

int glb_workspace_release(void *workspace) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  g_ws.erase(workspace);
  return GLB_OK;
}

size_t glb_log_softmax_workspace_bytes(int64_t n_rows, int64_t vocab) {
  if (n_rows <= 0 || vocab <= 0) return 0;
  return step_recs_bytes(n_rows, vocab) + align256((size_t)n_rows * 4) + 256;
}

int glb_log_softmax_rows(const void *logits, int32_t dtype, int64_t n_rows, int64_t vocab,
                         int64_t ld, float logit_scale, void *out, int32_t out_dtype, int64_t out_ld, float *out_lse,
                         void *workspace, size_t workspace_bytes, void *stream) {
  if (!logits) return fail(GLB_EINVAL, "logits is null");
  if (dtype < GLB_F32 || dtype > GLB_F16) return fail(GLB_EINVAL, "bad dtype %d", dtype);
  if (out_dtype != GLB_F32 && out_dtype != dtype)
    return fail(GLB_EINVAL, "out_dtype %d: log-probabilities leave as float32 or in the logits' own type (%d)", out_dtype, dtype);
  if (n_rows <= 0 || vocab <= 0) return fail(GLB_EINVAL, "n_rows / vocab must be positive");
  if (ld < vocab || (out && out_ld < vocab)) return fail(GLB_EINVAL, "ld / out_ld smaller than vocab");
  if (!out && !out_lse) return fail(GLB_EINVAL, "no output requested");
  if (vocab > 0x7fffff00ll || n_rows > 0x7fffffffll) return fail(GLB_EINVAL, "size exceeds 31 bits");
  const bool out16 = out_dtype != GLB_F32;
  if (out && ((uintptr_t)out) % (out16 ? 2 : 4)) return fail(GLB_EINVAL, "out pointer not element aligned");
  if (!(logit_scale == logit_scale)) return fail(GLB_EINVAL, "logit_scale is NaN");
  if (!workspace || ((uintptr_t)workspace) % 32) return fail(GLB_EINVAL, "workspace null or not 32-byte aligned");
  if (workspace_bytes < glb_log_softmax_workspace_bytes(n_rows, vocab))
    return fail(GLB_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, glb_log_softmax_workspace_bytes(n_rows, vocab));
  hipStream_t s = (hipStream_t)stream;
  const int nch = (int)n_chunks(vocab);
  const int mode = lsm_mode();
  const size_t recs_bytes = step_recs_bytes(n_rows, vocab);
  // one launch of independent waves that keep their chunks in registers and meet through tagged records: needs a
  // workspace glb_workspace_init has seen (the tags) and a row of at most 64 chunks (one lane per record)
  if (mode < 0 && out && nch <= 64 && n_rows * (int64_t)nch <= 0x7fffffffll) {
    uint32_t epoch = 0, *err = nullptr;
    const int rc = ws_next_epoch(workspace, recs_bytes, recs_bytes + align256((size_t)n_rows * 4), s, &epoch, &err);
    if (rc < 0) return hip_fail(hipGetLastError(), "workspace re-zero");
    if (rc > 0) {
      const uint64_t spin = g_spin_ticks.load(std::memory_order_relaxed);
      hipError_t e;
      switch (dtype) {
        case 0: e = glb::launch_logprob_waves_0(logits, ld, (int)vocab, nch, logit_scale, out, out_ld, out_lse, (int)n_rows, (uint64_t *)workspace, epoch, err, spin, out16, lsm_variant(), s); break;
        case 1: e = glb::launch_logprob_waves_1(logits, ld, (int)vocab, nch, logit_scale, out, out_ld, out_lse, (int)n_rows, (uint64_t *)workspace, epoch, err, spin, out16, lsm_variant(), s); break;
        default: e = glb::launch_logprob_waves_2(logits, ld, (int)vocab, nch, logit_scale, out, out_ld, out_lse, (int)n_rows, (uint64_t *)workspace, epoch, err, spin, out16, lsm_variant(), s); break;
      }
      if (e != hipSuccess) return hip_fail(e, "logprob_rows_waves launch");
      return GLB_OK;
    }
  }
  // any row length, any workspace, stream capture: chunk statistics, one wave per row for lse, one elementwise launch
  float *lse = out_lse ? out_lse : (float *)((char *)workspace + recs_bytes);
  glb::StepParams p{};
  p.logits = logits;
  p.ld = ld;
  p.V = (int32_t)vocab;
  p.nch = (int32_t)n_chunks(vocab);
  p.scale = logit_scale;
  p.n_particles = (int32_t)n_rows;
  p.n_pairs = (int32_t)n_rows;
  p.recs = (uint64_t *)workspace;
  p.out_lse = lse;
  hipError_t e = launch_stats(dtype, p, glb::kMaskNone, logit_scale != 1.0f, glb::kExpPoly, s);
  if (e != hipSuccess) return hip_fail(e, "chunk_stats launch");
  e = launch_finish(dtype, p, glb::kMaskNone, glb::kModeStats, glb::kExpPoly, s);
  if (e != hipSuccess) return hip_fail(e, "finish launch");
  if (out) {
    switch (dtype) {
      case 0: e = glb::launch_logprob_rows_0(logits, ld, (int)vocab, logit_scale, lse, out, out16, out_ld, (int)n_rows, s); break;
      case 1: e = glb::launch_logprob_rows_1(logits, ld, (int)vocab, logit_scale, lse, out, out16, out_ld, (int)n_rows, s); break;
      default: e = glb::launch_logprob_rows_2(logits, ld, (int)vocab, logit_scale, lse, out, out16, out_ld, (int)n_rows, s); break;
    }
    if (e != hipSuccess) return hip_fail(e, "logprob_rows launch");
  }
  return GLB_OK;
}

int glb_mask_f32_to_bits(const float *mask, int64_t n_masks, int64_t vocab, int64_t mask_ld,
                         uint32_t *out_bits, int64_t bits_ld, int32_t *out_nonbinary, void *stream) {
  if (!mask || !out_bits) return fail(GLB_EINVAL, "null pointer");
  if (n_masks <= 0 || vocab <= 0 || mask_ld < vocab || bits_ld < (vocab + 31) / 32)
    return fail(GLB_EINVAL, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (out_nonbinary) {
    const hipError_t e = hipMemsetAsync(out_nonbinary, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return hip_fail(e, "memset");
  }
  const int64_t total = n_masks * bits_ld;
  hipLaunchKernelGGL(mask_to_bits_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, mask,
                     n_masks, vocab, mask_ld, out_bits, bits_ld, out_nonbinary);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "mask_to_bits launch");
  return GLB_OK;
}

static int64_t group_cap(int64_t n) {
  int64_t cap = 64;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

size_t glb_group_contexts_workspace(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)(2 * group_cap(n) + 2 * n) * sizeof(int32_t) + (size_t)n * sizeof(uint64_t) + 8;
}

int glb_hash_contexts(const int32_t *tokens, const int64_t *starts, const int32_t *lengths, int64_t n,
                      uint64_t *out_hashes, void *stream) {
  if (!tokens || !starts || !lengths || !out_hashes) return fail(GLB_EINVAL, "null pointer");
  if (n <= 0 || n > (1 << 28)) return fail(GLB_EINVAL, "n out of range");
  hipLaunchKernelGGL(hash_contexts_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, (hipStream_t)stream, tokens, starts,
                     lengths, (int32_t)n, out_hashes);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "hash_contexts launch");
  return GLB_OK;
}

int glb_group_contexts(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                       int64_t n, const uint64_t *ctx_hashes, int32_t *out_group_of, int32_t *out_rep,
                       int32_t *out_n_groups, void *workspace, size_t workspace_bytes, void *stream) {
  if (!tokens || !starts || !lengths || !out_group_of || !out_rep || !out_n_groups || !workspace)
    return fail(GLB_EINVAL, "null pointer");
  if (n <= 0 || n > (1 << 28)) return fail(GLB_EINVAL, "n out of range");
  if (workspace_bytes < glb_group_contexts_workspace(n))
    return fail(GLB_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, glb_group_contexts_workspace(n));
  const int64_t cap = group_cap(n);
  if (n <= 8192) {  // table in LDS (2 * cap ints <= 128 KiB), slots in registers
    const size_t lds = (size_t)cap * 2 * sizeof(int32_t);
    hipStream_t s = (hipStream_t)stream;
    if (n < 256 && !ctx_hashes) {  // one launch: the workgroup hashes its own contexts (16-byte loads), meta data in LDS (36 KiB)
      const size_t lds1 = lds + 1024 * (sizeof(uint64_t) + sizeof(int64_t) + sizeof(int32_t));
      hipLaunchKernelGGL((group_contexts_lds_kernel<1, false>), dim3(1), dim3(1024), lds1, s, tokens, starts, lengths,
                         (int32_t)n, (int32_t)cap, out_group_of, out_rep, out_n_groups, (const uint64_t *)nullptr);
    } else {  // hashes - the caller's (kept up to date token by token: glb_particles_advance), or made here by many small
              // workgroups (one wave each: as many L1s as there are waves) - then the table
      const uint64_t *hashes = ctx_hashes;
      if (!hashes) {
        uint64_t *hw = (uint64_t *)((char *)workspace + (size_t)(2 * cap + 2 * n) * sizeof(int32_t));
        hipLaunchKernelGGL(hash_contexts_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, s, tokens, starts, lengths,
                           (int32_t)n, hw);
        hashes = hw;
      }
      if (n <= 1024) {
        hipLaunchKernelGGL((group_contexts_lds_kernel<1, true>), dim3(1), dim3(1024), lds, s, tokens, starts, lengths,
                           (int32_t)n, (int32_t)cap, out_group_of, out_rep, out_n_groups, hashes);
      } else if (n > 4096) {
        static std::atomic<uint64_t> big_lds{0};  // (more than 64 KiB of dynamic LDS: allowed once per device)
        const hipError_t big_lds_rc = glb::allow_dynamic_lds((const void *)group_contexts_lds_kernel<8, true>, 128 * 1024, big_lds);
        if (big_lds_rc != hipSuccess) return hip_fail(big_lds_rc, "hipFuncSetAttribute(group_contexts)");
        hipLaunchKernelGGL((group_contexts_lds_kernel<8, true>), dim3(1), dim3(1024), lds, s, tokens, starts, lengths,
                           (int32_t)n, (int32_t)cap, out_group_of, out_rep, out_n_groups, hashes);
      } else if (n <= 2048) {
        hipLaunchKernelGGL((group_contexts_lds_kernel<2, true>), dim3(1), dim3(1024), lds, s, tokens, starts, lengths,
                           (int32_t)n, (int32_t)cap, out_group_of, out_rep, out_n_groups, hashes);
      } else {
        hipLaunchKernelGGL((group_contexts_lds_kernel<4, true>), dim3(1), dim3(1024), lds, s, tokens, starts, lengths,
                           (int32_t)n, (int32_t)cap, out_group_of, out_rep, out_n_groups, hashes);
      }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "group_contexts launch");
    return GLB_OK;
  }
  int32_t *table = (int32_t *)workspace, *minidx = table + cap, *slot_of = minidx + cap, *gid_of = slot_of + n;
  const uint64_t *hashes = ctx_hashes;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(group_init_kernel, dim3(blocks_for(cap, 256)), dim3(256), 0, s, (int32_t)cap, table, minidx);
  if (!hashes) {
    uint64_t *hw = (uint64_t *)(gid_of + n);
    hipLaunchKernelGGL(hash_contexts_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, tokens, starts, lengths, (int32_t)n,
                       hw);
    hashes = hw;
  }
  hipLaunchKernelGGL(group_insert_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, tokens, starts, lengths, (int32_t)n,
                     (int32_t)cap, hashes, table, minidx, slot_of);
  hipLaunchKernelGGL(group_number_kernel, dim3(1), dim3(1024), 0, s, (int32_t)n, minidx, slot_of, gid_of, out_group_of,
                     out_rep, out_n_groups);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "group_contexts launch");
  return GLB_OK;
}

int glb_match_prefixes(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                       int64_t n, const int32_t *prefix_tokens, const int64_t *prefix_starts,
                       const int32_t *prefix_lengths, int64_t n_prefixes, int32_t *out_prefix,
                       int32_t *out_base, void *stream) {
  if (!tokens || !starts || !lengths || !out_prefix || !out_base) return fail(GLB_EINVAL, "null pointer");
  if (n <= 0 || n_prefixes < 0) return fail(GLB_EINVAL, "bad sizes");
  if (n_prefixes > 0 && (!prefix_tokens || !prefix_starts || !prefix_lengths))
    return fail(GLB_EINVAL, "null prefix table");
  hipLaunchKernelGGL(match_prefixes_kernel, dim3(blocks_for(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, tokens, starts, lengths, n, prefix_tokens, prefix_starts,
                     prefix_lengths, n_prefixes, out_prefix, out_base);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "match_prefixes launch");
  return GLB_OK;
}

int glb_gather_padded(const int32_t *tokens, const int64_t *starts, const int32_t *lengths,
                      const int32_t *sel, int64_t n_sel, const int32_t *base, int64_t pad_id, int64_t p_max,
                      int64_t l_max, int64_t *out_input_ids, int64_t *out_attention_mask,
                      int64_t *out_position_ids, int32_t *out_last_index, void *stream) {
  if (!tokens || !starts || !lengths || !out_input_ids || !out_attention_mask || !out_position_ids)
    return fail(GLB_EINVAL, "null pointer");
  if (n_sel <= 0 || l_max <= 0 || p_max < 0) return fail(GLB_EINVAL, "bad sizes");
  const int64_t total = n_sel * (p_max + l_max);
  hipLaunchKernelGGL(gather_padded_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, tokens, starts, lengths, sel, n_sel, base, pad_id, p_max, l_max,
                     out_input_ids, out_attention_mask, out_position_ids, out_last_index);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "gather_padded launch");
  return GLB_OK;
}

int glb_gather_kv_padded(const void *const *slabs, const int32_t *slab_len, int64_t n_prefixes,
                         const int32_t *prefix_of, int64_t n_rows, int64_t heads, int64_t head_dim,
                         int64_t p_max, int32_t elem_bytes, void *out, void *stream) {
  if (!slabs || !slab_len || !prefix_of || !out) return fail(GLB_EINVAL, "null pointer");
  if (n_prefixes <= 0 || n_rows <= 0 || heads <= 0 || head_dim <= 0 || p_max <= 0)
    return fail(GLB_EINVAL, "bad sizes");
  if (elem_bytes != 2 && elem_bytes != 4) return fail(GLB_EINVAL, "elem_bytes must be 2 or 4");
  const int64_t rowb = head_dim * elem_bytes;
  hipStream_t s = (hipStream_t)stream;
  if (rowb % 16 == 0 && ((uintptr_t)out) % 16 == 0) {
    const int64_t rv = rowb / 16, total = n_rows * heads * p_max * rv;
    hipLaunchKernelGGL(gather_kv_kernel<uint4>, dim3(blocks_for(total, 256)), dim3(256), 0, s,
                       slabs, slab_len, prefix_of, n_rows, heads, rv, p_max, (uint4 *)out);
  } else if (elem_bytes == 4) {
    const int64_t total = n_rows * heads * p_max * head_dim;
    hipLaunchKernelGGL(gather_kv_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s,
                       slabs, slab_len, prefix_of, n_rows, heads, head_dim, p_max, (uint32_t *)out);
  } else {
    const int64_t total = n_rows * heads * p_max * head_dim;
    hipLaunchKernelGGL(gather_kv_kernel<uint16_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s,
                       slabs, slab_len, prefix_of, n_rows, heads, head_dim, p_max, (uint16_t *)out);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "gather_kv launch");
  return GLB_OK;
}

int glb_particles_advance(int32_t *contexts, int64_t ctx_ld, int32_t *lengths, int32_t *active,
                          float *log_weights, const float *logZ, const int32_t *token, int64_t n,
                          int32_t eos_id, int32_t max_len, uint64_t *hashes, void *stream) {
  if (!contexts || !lengths || !active || !log_weights || !logZ || !token)
    return fail(GLB_EINVAL, "null pointer");
  if (n <= 0 || ctx_ld <= 0 || max_len > ctx_ld) return fail(GLB_EINVAL, "bad sizes");
  hipLaunchKernelGGL(particles_advance_kernel, dim3(blocks_for(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, contexts, ctx_ld, lengths, active, log_weights, logZ,
                     token, n, eos_id, max_len, hashes);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "particles_advance launch");
  return GLB_OK;
}

void glb_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  glb::philox4x32_10(ctr, key, out);
}

}  // extern "C"
