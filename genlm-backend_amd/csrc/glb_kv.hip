// glb_kv.hip — KV rows shared between contexts, decided on the device (include/glb.h: glb_match_rows, glb_kv_plan).
//
// The reference keeps per-token KV on trie nodes (cache.py:103-191, mlx.py:177-318) or re-encodes every context
// (hf.py:202-288).  Here KV lives in slab rows [R, heads, cap, head_dim]; which context's prefix sits in which row is a
// block table.  Round 3 decided that table on the host from a D2H copy of 3 + 3 N ints per step (np.unique / setdiff1d
// / argsort) and shipped six small index tensors back; these two kernels keep the whole decision on the device and
// leave the host an eight-word head to read (how many rows of which kind: the shapes of the forwards it launches).
// Integer work only; restated in oracle/oracle.py (kv_plan / match_rows) and compared bit for bit.
#include <hip/hip_runtime.h>

#include <climits>

#include "../../include/glb.h"
#include "glb_common.hpp"

namespace {

__device__ inline uint64_t ctx_hash_step(uint64_t h, uint32_t v) {  // (= glb_api.hip: the context hash of glb_hash_contexts)
  h ^= v;
  h *= 0x100000001b3ull;
  return h ^ (h >> 29);
}

// ---------------------------------------------------------------------------------------------------------------------
// glb_match_rows: one wave per dedup group.  The wave hashes its context (all lanes alike, token by token: the value
// before the last token is the parent's hash), then the lanes walk the table rows r = lane, lane + 64, ...: a row whose
// (hash, length) equal the context's - or the parent's - has its tokens compared, every candidate, so equal hashes of
// unequal contexts cost a comparison and nothing else.  The smallest matching row wins.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void match_rows_kernel(const int32_t *tok, const int64_t *st, const int32_t *len,
                                                        const int32_t *rep, const int32_t *n_groups, const int32_t *row_tok,
                                                        const int32_t *row_len, const uint64_t *row_hash, int32_t R,
                                                        int32_t cap, int32_t *out_old, uint64_t *out_hash) {
  const int u = blockIdx.x, lane = threadIdx.x;
  if (u >= *n_groups) return;
  const int i = rep[u];
  const int32_t L = len[i];
  const int32_t *t = tok + st[i];
  uint64_t h = 0xcbf29ce484222325ull, h_par = h;
  for (int j = 0; j < L; ++j) {
    h_par = h;
    h = ctx_hash_step(h, (uint32_t)t[j]);
  }
  if (lane == 0 && out_hash) out_hash[u] = h;
  int best_exact = INT_MAX, best_par = INT_MAX;
  if (L <= cap && L > 0) {
    for (int r = lane; r < R; r += 64) {
      const int32_t rl = row_len[r];
      if (rl <= 0) continue;  // the row holds nothing
      const uint64_t rh = row_hash[r];
      const bool ce = rl == L && rh == h, cp = rl == L - 1 && rh == h_par;
      if (!ce && !cp) continue;
      const int32_t *q = row_tok + (int64_t)r * cap;
      bool same = true;
      for (int j = 0; j < rl && same; ++j) same = q[j] == t[j];
      if (same) {
        if (ce) best_exact = r < best_exact ? r : best_exact;
        else best_par = r < best_par ? r : best_par;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(best_exact, o, 64), b = __shfl_xor(best_par, o, 64);
    best_exact = a < best_exact ? a : best_exact;
    best_par = b < best_par ? b : best_par;
  }
  if (lane == 0) out_old[u] = best_exact != INT_MAX ? best_exact : (best_par != INT_MAX ? best_par : -1);
}

// ---------------------------------------------------------------------------------------------------------------------
// glb_kv_plan: one workgroup.  Exclusive scans over the groups / rows run in strides of the workgroup with a carry.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kT = 1024;

struct Scan {
  int *s_wave;  // [17] in LDS
  // exclusive prefix of v over the workgroup's threads, block total in `total` (two barriers)
  __device__ __forceinline__ int excl(int v, int &total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (wave == 0) {
      int w = lane < kT / 64 ? s_wave[lane] : 0, wi = w;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int up = __shfl_up(wi, o, 64);
        if (lane >= o) wi += up;
      }
      if (lane < kT / 64) s_wave[lane] = wi - w;
      if (lane == kT / 64 - 1) s_wave[16] = wi;
    }
    __syncthreads();
    const int base = s_wave[wave];
    total = s_wave[16];
    __syncthreads();  // (s_wave is reused by the next call)
    return base + incl - v;
  }
};

struct PlanArgs {
  int32_t n, R, cap, lds_rows;
  const int32_t *group_of, *rep, *n_groups, *old_src, *old_sel, *lengths;
  int64_t *stamps;  // nullable
  int64_t call_no;
  int32_t *grp_row, *inv, *rows_a, *reps_a, *pos_a, *reps_b, *rows_b, *copy_src, *copy_len, *ctx_of_row, *pos_full, *row_of_new,
      *head;
  // the table of what every row holds (nullable: row_tok == null)
  int32_t *row_tok, *row_len;
  uint64_t *row_hash;
  const uint64_t *grp_hash;
  const int32_t *tok;
  const int64_t *st;
  // scratch: keeper[R], free_by_rank[R], flags[n]
  int32_t *keeper, *free_by_rank, *flags;
};

__global__ __launch_bounds__(kT) void kv_plan_kernel(const PlanArgs a) {
  __shared__ int s_wave[17];
  __shared__ int s_lmax, s_unkept, s_copied;
  Scan scan{s_wave};
  const int tid = threadIdx.x, n = a.n, R = a.R;
  const int U = *a.n_groups < n ? (*a.n_groups < 0 ? 0 : *a.n_groups) : n;  // (never past the scratch sized for n groups)
  if (tid == 0) s_lmax = 0, s_unkept = 0, s_copied = 0;
  for (int r = tid; r < R; r += kT) {
    a.keeper[r] = INT_MAX;
    a.copy_src[r] = -1;
    a.copy_len[r] = 0;
    a.ctx_of_row[r] = -1;
    a.pos_full[r] = 0;
  }
  __syncthreads();
  // ---- the row a group's prefix sits in; the first group (by id) of every live row keeps it
  for (int u = tid; u < U; u += kT) {
    int o = a.old_src[a.old_sel ? a.old_sel[u] : u];
    // (a row index outside the table reads as "no row"; so does the row of a context that has outgrown a row's cap
    // positions: it is encoded from its tokens and its row goes back to the free ones - never a position past the row)
    o = o >= 0 && o < R && a.lengths[a.rep[u]] <= a.cap ? o : -1;
    a.flags[u] = o;
    if (o >= 0) atomicMin(&a.keeper[o], u);
  }
  __syncthreads();
  // ---- free rows in the order they are handed out: by (stamp, row) - longest unused first - or by row
  int n_free = 0;
  if (a.stamps == nullptr) {
    int carry = 0;
    for (int r0 = 0; r0 < R; r0 += kT) {
      const int r = r0 + tid;
      const int f = r < R && a.keeper[r] == INT_MAX ? 1 : 0;
      int tot;
      const int ex = scan.excl(f, tot);
      if (f) a.free_by_rank[carry + ex] = r;
      carry += tot;
    }
    n_free = carry;
  } else {
    // rank of a free row among the free rows by (stamp, row): counted against every other row.  The keys - how many calls
    // ago the row was used, -1 for a row that is not free - sit in LDS when they fit (lds_rows: the launch's dynamic LDS),
    // so the R x n_free comparisons are LDS broadcasts, not global loads (1280 rows: 222 us from global memory)
    extern __shared__ int s_age[];
    const bool in_lds = a.lds_rows >= R;
    if (in_lds) {
      for (int r = tid; r < R; r += kT) {
        const int64_t age = a.call_no - a.stamps[r];
        s_age[r] = a.keeper[r] != INT_MAX ? -1 : (int)(age < 0 ? 0 : (age > 0x7ffffffell ? 0x7ffffffell : age));
      }
      __syncthreads();
    }
    for (int r = tid; r < R; r += kT) {
      if (a.keeper[r] != INT_MAX) continue;
      int rank = 0;
      if (in_lds) {
        const int mine = s_age[r];
        for (int q = 0; q < R; ++q) {
          const int o = s_age[q];
          rank += (o > mine || (o == mine && q < r)) ? 1 : 0;  // (older first; a row that is not free has age -1 < mine)
        }
      } else {
        const int64_t s = a.stamps[r];
        for (int q = 0; q < R; ++q)
          if (a.keeper[q] == INT_MAX) {
            const int64_t sq = a.stamps[q];
            rank += (sq < s || (sq == s && q < r)) ? 1 : 0;
          }
      }
      a.free_by_rank[rank] = r;
    }
    int carry = 0;
    for (int r0 = 0; r0 < R; r0 += kT) {
      const int r = r0 + tid;
      int tot;
      scan.excl(r < R && a.keeper[r] == INT_MAX ? 1 : 0, tot);
      carry += tot;
    }
    n_free = carry;
    __syncthreads();
  }
  // ---- who needs a row: groups that grew out of a row somebody else keeps (a copy is cheaper than an encoding: first),
  //      then groups without a row whose context fits one
  int n_copy_cand = 0;
  {
    int carry = 0;
    for (int u0 = 0; u0 < U; u0 += kT) {
      const int u = u0 + tid;
      int c = 0;
      if (u < U) {
        const int o = a.flags[u];
        c = o >= 0 && a.keeper[o] != u ? 1 : 0;
      }
      int tot;
      scan.excl(c, tot);
      carry += tot;
    }
    n_copy_cand = carry;
  }
  int carry_c = 0, carry_f = 0;
  for (int u0 = 0; u0 < U; u0 += kT) {
    const int u = u0 + tid;
    int o = -1, L = 0, cand = 0, fresh = 0, keep = 0;
    if (u < U) {
      o = a.flags[u];
      L = a.lengths[a.rep[u]];
      keep = o >= 0 && a.keeper[o] == u;
      cand = o >= 0 && !keep;
      fresh = o < 0 && L <= a.cap;
    }
    int tot_c, tot_f;
    const int ex_c = scan.excl(cand, tot_c), ex_f = scan.excl(fresh, tot_f);
    if (u < U) {
      int row = -1;
      if (keep) row = o;
      else {
        const int need = cand ? carry_c + ex_c : (fresh ? n_copy_cand + carry_f + ex_f : INT_MAX);
        if (need < n_free) row = a.free_by_rank[need];
      }
      a.grp_row[u] = row;
    }
    carry_c += tot_c;
    carry_f += tot_f;
  }
  __syncthreads();
  // ---- the forward's rows: groups with a prefix in a row (kept or copied) first, the ones to encode behind them
  int n_a = 0;
  {
    int carry = 0;
    for (int u0 = 0; u0 < U; u0 += kT) {
      const int u = u0 + tid;
      int in_a = 0;
      if (u < U) {
        const int o = a.flags[u];
        in_a = o >= 0 && a.grp_row[u] >= 0 ? 1 : 0;
      }
      int tot;
      scan.excl(in_a, tot);
      carry += tot;
    }
    n_a = carry;
  }
  int carry_a = 0;
  for (int u0 = 0; u0 < U; u0 += kT) {
    const int u = u0 + tid;
    int in_a = 0, o = -1, row = -1, L = 0, ctx = 0;
    if (u < U) {
      o = a.flags[u];
      row = a.grp_row[u];
      ctx = a.rep[u];
      L = a.lengths[ctx];
      in_a = o >= 0 && row >= 0 ? 1 : 0;
    }
    int tot;
    const int ex = scan.excl(in_a, tot);
    if (u < U) {
      if (in_a) {
        const int k = carry_a + ex;
        a.inv[u] = k;
        a.rows_a[k] = row;
        a.reps_a[k] = ctx;
        a.pos_a[k] = L - 1;
        a.ctx_of_row[row] = ctx;
        a.pos_full[row] = L - 1;
        if (row != o) {  // a copy of the prefix into a free row
          a.copy_src[row] = o;
          a.copy_len[row] = L - 1;
          atomicAdd(&s_copied, 1);
        }
      } else {
        const int k = u - (carry_a + ex);  // rank among the groups to encode
        a.inv[u] = n_a + k;
        a.reps_b[k] = ctx;
        a.rows_b[k] = row;
        atomicMax(&s_lmax, L);
        if (row < 0) atomicAdd(&s_unkept, 1);
        else a.ctx_of_row[row] = -2;  // a row that is about to be filled from an encoding
      }
      if (row >= 0) {
        if (a.stamps) a.stamps[row] = a.call_no;
        if (a.row_tok) {  // the table of what the rows hold now
          a.row_len[row] = L;
          a.row_hash[row] = a.grp_hash[u];
          const int32_t *t = a.tok + a.st[ctx];
          int32_t *q = a.row_tok + (int64_t)row * a.cap;
          for (int j = 0; j < a.cap; ++j) q[j] = j < L ? t[j] : 0;
        }
      }
    }
    carry_a += tot;
  }
  __syncthreads();
  // ---- every context's row from now on
  if (a.row_of_new)
    for (int i = tid; i < n; i += kT) a.row_of_new[i] = a.grp_row[a.group_of[i]];
  if (tid == 0) {
    a.head[0] = U;
    a.head[1] = n_a;
    a.head[2] = U - n_a;
    a.head[3] = s_copied;
    a.head[4] = s_unkept;
    a.head[5] = s_lmax;
    a.head[6] = n_free;
    a.head[7] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// glb_slab_attention: the attention of a one-token forward over KV slab rows [R, heads_kv, cap, head_dim] where they lie.
// PyTorch's SDPA takes 87 us per GPT-2-small layer for 1024 rows x 12 heads x <= 19 cached positions (a dense
// [R, H, 1, cap] problem with an explicit mask); the bytes are 2 x (pos + 1) x head_dim per (row, head), a streaming read.
// One wave per (row, query head).  A 16-byte vector is EPV elements; LP = head_dim / EPV lanes cover one cached position,
// so one load instruction of the wave reads PP = 64 / LP consecutive positions - 1 KB of consecutive addresses.  Every
// position slot keeps its own online softmax (running maximum, denominator, weighted V sum) over the positions it is
// dealt; the PP slots are merged once at the end.  The new token's K / V come straight from the projection output and
// are written to their slab position by the wave of the group's first query head (the append of glb_kv_append, fused).
// float32 accumulation whatever the element type.
// ---------------------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ void unpack16(const u32x4 &w, float *x) {
  const uint32_t v[4] = {w.x, w.y, w.z, w.w};
  if constexpr (DT == GLB_F32) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = __uint_as_float(v[k]);
  } else if constexpr (DT == GLB_BF16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[2 * k] = __uint_as_float(v[k] << 16);
      x[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[2 * k] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v[k] & 0xffffu));
      x[2 * k + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v[k] >> 16));
    }
  }
}

template <int DT>
__device__ __forceinline__ u32x4 pack16v(const float *x) {
  if constexpr (DT == GLB_F32) {
    return u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
  } else {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (DT == GLB_BF16) {
        const __bf16 a = (__bf16)x[2 * k], b = (__bf16)x[2 * k + 1];
        w[k] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
      } else {
        const _Float16 a = (_Float16)x[2 * k], b = (_Float16)x[2 * k + 1];
        w[k] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
      }
    }
    return u32x4{w[0], w[1], w[2], w[3]};
  }
}

struct AttnArgs {
  const char *q, *k_new, *v_new;
  char *k_slab, *v_slab, *out;
  const int32_t *pos;
  int64_t q_sr, q_sh, k_sr, k_sh, v_sr, v_sh;  // element strides (row, head) of the query, the new K, the new V
  int32_t R, H, Hkv, cap;
  float scale;
};

template <int DT, int DH>
__global__ __launch_bounds__(64) void slab_attention_kernel(const AttnArgs a) {
  constexpr int ES = DT == GLB_F32 ? 4 : 2, EPV = 16 / ES, LP = DH / EPV, PP = 64 / LP;
  static_assert(LP >= 1 && LP <= 64 && PP * LP == 64, "head_dim / vector width must divide the wave");
  const int lane = threadIdx.x, j = lane / LP, i = lane - j * LP;
  const int r = blockIdx.x / a.H, h = blockIdx.x - r * a.H;
  const int G = a.H / a.Hkv, hk = h / G;
  int p_new = a.pos[r];  // the token being appended sits at p_new; positions 0 .. p_new are attended to
  const bool bad_pos = p_new < 0 || p_new >= a.cap;  // a position outside the row: the caller's bug - no fault, no plausible answer: NaN out, nothing appended
  p_new = p_new < 0 ? 0 : (p_new < a.cap ? p_new : a.cap - 1);
  float qf[EPV], kn[EPV], vn[EPV];
  unpack16<DT>(*reinterpret_cast<const u32x4 *>(a.q + ((int64_t)r * a.q_sr + (int64_t)h * a.q_sh + i * EPV) * ES), qf);
  const u32x4 kn_raw = *reinterpret_cast<const u32x4 *>(a.k_new + ((int64_t)r * a.k_sr + (int64_t)hk * a.k_sh + i * EPV) * ES);
  const u32x4 vn_raw = *reinterpret_cast<const u32x4 *>(a.v_new + ((int64_t)r * a.v_sr + (int64_t)hk * a.v_sh + i * EPV) * ES);
  unpack16<DT>(kn_raw, kn);
  unpack16<DT>(vn_raw, vn);
  const int64_t slab_row = ((int64_t)r * a.Hkv + hk) * a.cap;
  if (h == hk * G && j == 0 && !bad_pos) {  // the append: once per (row, KV head)
    *reinterpret_cast<u32x4 *>(a.k_slab + ((slab_row + p_new) * DH + i * EPV) * ES) = kn_raw;
    *reinterpret_cast<u32x4 *>(a.v_slab + ((slab_row + p_new) * DH + i * EPV) * ES) = vn_raw;
  }
  float m = -__builtin_huge_valf(), l = 0.0f, acc[EPV];
#pragma unroll
  for (int k = 0; k < EPV; ++k) acc[k] = 0.0f;
  for (int p0 = 0; p0 <= p_new; p0 += PP) {
    const int p = p0 + j;
    float kf[EPV], vf[EPV];
    if (p < p_new) {
      unpack16<DT>(*reinterpret_cast<const u32x4 *>(a.k_slab + ((slab_row + p) * DH + i * EPV) * ES), kf);
      unpack16<DT>(*reinterpret_cast<const u32x4 *>(a.v_slab + ((slab_row + p) * DH + i * EPV) * ES), vf);
    } else {
#pragma unroll
      for (int k = 0; k < EPV; ++k) kf[k] = kn[k], vf[k] = vn[k];
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < EPV; ++k) s = __builtin_fmaf(qf[k], kf[k], s);
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o, 64);  // the LP lanes of a position end up with its score
    if (p <= p_new) {
      s *= a.scale;
      const float m2 = fmaxf(m, s), c = __expf(m - m2), w = __expf(s - m2);
      l = l * c + w;
#pragma unroll
      for (int k = 0; k < EPV; ++k) acc[k] = __builtin_fmaf(acc[k], c, w * vf[k]);
      m = m2;
    }
  }
  // merge the PP position slots (lanes i, i + LP, i + 2 LP, ...)
#pragma unroll
  for (int o = LP; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o, 64), lo = __shfl_xor(l, o, 64);
    const float m2 = fmaxf(m, mo);
    const float c = m == m2 ? 1.0f : __expf(m - m2), co = mo == m2 ? 1.0f : __expf(mo - m2);  // (-inf - -inf: an empty slot)
    l = l * c + lo * co;
#pragma unroll
    for (int k = 0; k < EPV; ++k) acc[k] = acc[k] * c + __shfl_xor(acc[k], o, 64) * co;
    m = m2;
  }
  if (j == 0) {
    const float inv = bad_pos ? __builtin_nanf("") : 1.0f / l;
#pragma unroll
    for (int k = 0; k < EPV; ++k) acc[k] *= inv;
    *reinterpret_cast<u32x4 *>(a.out + (((int64_t)r * a.H + h) * DH + i * EPV) * ES) = pack16v<DT>(acc);
  }
}

template <int DT>
hipError_t launch_attention(const AttnArgs &a, int head_dim, hipStream_t s) {
  const dim3 grid((unsigned)((int64_t)a.R * a.H)), block(64);
  if (head_dim == 64) hipLaunchKernelGGL((slab_attention_kernel<DT, 64>), grid, block, 0, s, a);
  else if (head_dim == 128) hipLaunchKernelGGL((slab_attention_kernel<DT, 128>), grid, block, 0, s, a);
  else if (head_dim == 32) hipLaunchKernelGGL((slab_attention_kernel<DT, 32>), grid, block, 0, s, a);
  else if (head_dim == 16) hipLaunchKernelGGL((slab_attention_kernel<DT, 16>), grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// glb_short_attention: masked attention of the padded batches of short contexts the hot path feeds the transformer
// (hf.py:232-281: ragged contexts right-padded to the batch's longest, optionally behind zero-padded cached prefixes) -
// a dozen query positions against a dozen or two keys per (row, head).  PyTorch's SDPA spends 410 us per GPT-2-small layer
// on 919 rows x 12 heads x 13 tokens with the padding mask (a fifth of the 24 ms step); the arithmetic is 5 k MACs per
// (row, head).  One wave per (row, head, query position), the slab kernel's scheme: LP lanes per key, PP keys per pass,
// an online softmax per key slot, slots merged at the end; the (row, head)'s K / V are read by its Lq waves out of the
// L1 / L2.  Which keys a query may see comes from the boolean mask transformers built ([U, 1, Lq, Lk], true = attend),
// or without one from causality (key s <= query t + Lk - Lq).  A query that may see nothing gets zeros.
// (Measured and not kept: a tile of 256 / head_dim queries per wave with the whole dot product inside a lane - v_dot2_f32_bf16
// on the packed pairs for 16-bit types - scores through LDS and a second phase of four output dimensions per lane: a third
// of the instructions per query, but every lane then reads whole K / q rows 256 bytes apart, sixteen cache lines per load
// instruction: 105.6 us against 72.8 for fp32 GPT-2-small rows, 84.0 against 92.8 for bf16 Llama rows.)
// ---------------------------------------------------------------------------------------------------------------------
struct ShortArgs {
  const char *q, *k, *v;
  const uint8_t *mask;  // nullable
  char *out;
  int64_t q_sr, q_sh, q_sp, k_sr, k_sh, k_sp, v_sr, v_sh, v_sp, m_sr, m_sq;  // element strides: row, head, position
  int32_t U, H, Hkv, Lq, Lk;
  float scale;
};

template <int DT, int DH>
__global__ __launch_bounds__(64) void short_attention_kernel(const ShortArgs a) {
  constexpr int ES = DT == GLB_F32 ? 4 : 2, EPV = 16 / ES, LP = DH / EPV, PP = 64 / LP;
  static_assert(LP >= 1 && LP <= 64 && PP * LP == 64, "head_dim / vector width must divide the wave");
  const int lane = threadIdx.x, j = lane / LP, i = lane - j * LP;
  const int t = blockIdx.x % a.Lq, uh = blockIdx.x / a.Lq, h = uh % a.H, u = uh / a.H;
  const int hk = h / (a.H / a.Hkv);
  float qf[EPV];
  unpack16<DT>(*reinterpret_cast<const u32x4 *>(a.q + ((int64_t)u * a.q_sr + (int64_t)h * a.q_sh + (int64_t)t * a.q_sp + i * EPV) * ES), qf);
  const uint8_t *mrow = a.mask ? a.mask + (int64_t)u * a.m_sr + (int64_t)t * a.m_sq : nullptr;
  const int last = mrow ? a.Lk - 1 : (t + a.Lk - a.Lq < a.Lk - 1 ? t + a.Lk - a.Lq : a.Lk - 1);  // causal: keys beyond it are never seen
  const char *kb = a.k + ((int64_t)u * a.k_sr + (int64_t)hk * a.k_sh + i * EPV) * ES;
  const char *vb = a.v + ((int64_t)u * a.v_sr + (int64_t)hk * a.v_sh + i * EPV) * ES;
  float m = -__builtin_huge_valf(), l = 0.0f, acc[EPV];
#pragma unroll
  for (int k = 0; k < EPV; ++k) acc[k] = 0.0f;
  for (int p0 = 0; p0 <= last; p0 += PP) {
    const int p = p0 + j;
    const bool in = p <= last;
    const bool see = in && (mrow ? mrow[p] != 0 : true);
    float kf[EPV], vf[EPV];
    const int pc = in ? p : last;  // (lanes past the end read the last key and ignore it)
    unpack16<DT>(*reinterpret_cast<const u32x4 *>(kb + (int64_t)pc * a.k_sp * ES), kf);
    unpack16<DT>(*reinterpret_cast<const u32x4 *>(vb + (int64_t)pc * a.v_sp * ES), vf);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < EPV; ++k) s = __builtin_fmaf(qf[k], kf[k], s);
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o, 64);
    if (see) {
      s *= a.scale;
      const float m2 = fmaxf(m, s), c = __expf(m - m2), w = __expf(s - m2);
      l = l * c + w;
#pragma unroll
      for (int k = 0; k < EPV; ++k) acc[k] = __builtin_fmaf(acc[k], c, w * vf[k]);
      m = m2;
    }
  }
#pragma unroll
  for (int o = LP; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o, 64), lo = __shfl_xor(l, o, 64);
    const float m2 = fmaxf(m, mo);
    const float c = m == m2 ? 1.0f : __expf(m - m2), co = mo == m2 ? 1.0f : __expf(mo - m2);
    l = l * c + lo * co;
#pragma unroll
    for (int k = 0; k < EPV; ++k) acc[k] = acc[k] * c + __shfl_xor(acc[k], o, 64) * co;
    m = m2;
  }
  if (j == 0) {
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
    for (int k = 0; k < EPV; ++k) acc[k] *= inv;
    *reinterpret_cast<u32x4 *>(a.out + ((((int64_t)u * a.Lq + t) * a.H + h) * DH + i * EPV) * ES) = pack16v<DT>(acc);
  }
}

// Round 5: the same attention with a wave per (row, KV head) instead of per (row, head, query).  The wave above was a
// query: eight key slots of eight lanes, an online softmax per slot and a merge of the slots through three rounds of
// nine lane exchanges - ~140 instructions and three memory round trips for 13 keys, 191 000 waves a layer at 460 rows x
// 32 heads x 13 positions (96 us a layer: a tenth of the Llama-shaped step).  Here the SLOTS ARE QUERIES: the (row, KV
// head)'s K and V go to LDS once as float32 (13 keys x 64 dims: 6.6 KB), the wave walks the G x Lq queries that share
// them PP at a time (PP = 64 / LP slots of LP lanes, a lane holds the query's and the output's EPV-element chunk), a
// key's score is one dot product chunk per lane + log2(LP) DPP adds inside the slot, the scores of a query stay in
// registers, and nothing is ever merged across slots: ~40 instructions a query.  Keys up to KMAX (16 / 32) per call;
// longer key ranges keep the kernel above.
template <int LP>
__device__ __forceinline__ float slot_sum(float v) {
  // sum over the LP consecutive lanes of a slot, every lane gets it (DPP: quad_perm xor 1, xor 2, row_half_mirror, row_mirror)
  if constexpr (LP >= 2) v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false));
  if constexpr (LP >= 4) v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false));
  if constexpr (LP >= 8) v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false));
  if constexpr (LP >= 16) v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false));
  if constexpr (LP >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (LP >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int DT, int DH, int KMAX>
__global__ __launch_bounds__(64) void short_attention_rows_kernel(const ShortArgs a) {
  constexpr int ES = DT == GLB_F32 ? 4 : 2, EPV = 16 / ES, LP = DH / EPV, PP = 64 / LP;
  static_assert(LP >= 1 && LP <= 64 && PP * LP == 64, "head_dim / vector width must divide the wave");
  // up to 16 keys: a lane keeps its chunk of every key in registers (K never passes through LDS: with K and V both read
  // from LDS for every query group the LDS data path - 1 KB a read instruction - took as long as the arithmetic)
  constexpr bool KREG = KMAX * EPV <= 128;
  extern __shared__ float kv_lds[];  // V [Lk][DH] (and before it K [Lk][DH] when the keys do not fit registers), float32
  const int lane = threadIdx.x, j = lane / LP, i = lane - j * LP;
  const int G = a.H / a.Hkv;
  const int hk = blockIdx.x % a.Hkv, u = blockIdx.x / a.Hkv;
  float *Ks = kv_lds, *Vs = KREG ? kv_lds : kv_lds + a.Lk * DH;
  const char *kb = a.k + ((int64_t)u * a.k_sr + (int64_t)hk * a.k_sh) * ES;
  const char *vb = a.v + ((int64_t)u * a.v_sr + (int64_t)hk * a.v_sh) * ES;
  const int nq = G * a.Lq;
  // the first query group's vectors go out with the keys
  auto q_addr = [&](int qi) {
    const int qc = qi < nq ? qi : nq - 1;
    const int g = qc / a.Lq, t = qc - g * a.Lq;
    return a.q + ((int64_t)u * a.q_sr + (int64_t)(hk * G + g) * a.q_sh + (int64_t)t * a.q_sp + i * EPV) * ES;
  };
  u32x4 q_raw = *reinterpret_cast<const u32x4 *>(q_addr(j));
  float kr[KREG ? KMAX : 1][EPV];
  if constexpr (KREG) {
    u32x4 raw[KMAX];
#pragma unroll
    for (int p = 0; p < KMAX; ++p)
      raw[p] = *reinterpret_cast<const u32x4 *>(kb + ((int64_t)(p < a.Lk ? p : 0) * a.k_sp + i * EPV) * ES);
#pragma unroll
    for (int p = 0; p < KMAX; ++p) unpack16<DT>(raw[p], kr[p]);
  }
  for (int x = lane; x < a.Lk * LP; x += 64) {
    const int p = x / LP, c = x - p * LP;
    float vf[EPV];
    unpack16<DT>(*reinterpret_cast<const u32x4 *>(vb + ((int64_t)p * a.v_sp + c * EPV) * ES), vf);
#pragma unroll
    for (int k = 0; k < EPV; ++k) Vs[p * DH + c * EPV + k] = vf[k];
    if constexpr (!KREG) {
      float kf[EPV];
      unpack16<DT>(*reinterpret_cast<const u32x4 *>(kb + ((int64_t)p * a.k_sp + c * EPV) * ES), kf);
#pragma unroll
      for (int k = 0; k < EPV; ++k) Ks[p * DH + c * EPV + k] = kf[k];
    }
  }
  // which keys a query position may see, one word per position, once per wave
  const uint32_t all_keys = a.Lk >= 32 ? 0xffffffffu : ((1u << a.Lk) - 1u);
  uint32_t *vis_t = reinterpret_cast<uint32_t *>(kv_lds + (KREG ? 1 : 2) * a.Lk * DH);
  for (int t = lane; t < a.Lq; t += 64) {
    uint32_t vis = 0;
    if (a.mask) {
      const uint8_t *mrow = a.mask + (int64_t)u * a.m_sr + (int64_t)t * a.m_sq;
      uint8_t mb[KMAX];
#pragma unroll
      for (int p = 0; p < KMAX; ++p) mb[p] = p < a.Lk ? mrow[p] : (uint8_t)0;
#pragma unroll
      for (int p = 0; p < KMAX; ++p) vis |= (mb[p] != 0 ? 1u : 0u) << p;
    } else {
      const int last = t + a.Lk - a.Lq;  // causal: keys 0 .. last
      vis = last >= 31 ? 0xffffffffu : ((2u << last) - 1u);
    }
    vis_t[t] = vis & all_keys;
  }
  __syncthreads();
  for (int q0 = 0; q0 < nq; q0 += PP) {
    const int qi = q0 + j;
    const bool live = qi < nq;
    const int qc = live ? qi : nq - 1;
    const int g = qc / a.Lq, t = qc - g * a.Lq, h = hk * G + g;
    float qf[EPV];
    unpack16<DT>(q_raw, qf);
    if (q0 + PP < nq) q_raw = *reinterpret_cast<const u32x4 *>(q_addr(qi + PP));  // the next group's vectors travel while this one is worked on
    const uint32_t vis = vis_t[t];  // bit p: query t may see key p
    float sc[KMAX];
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int p = 0; p < KMAX; ++p) {
      sc[p] = -__builtin_huge_valf();
      if (p < a.Lk) {  // (wave-uniform)
        float s = 0.0f;
        if constexpr (KREG) {
#pragma unroll
          for (int k = 0; k < EPV; ++k) s = __builtin_fmaf(qf[k], kr[p][k], s);
        } else {
          const float *kp = Ks + p * DH + i * EPV;
#pragma unroll
          for (int k = 0; k < EPV; ++k) s = __builtin_fmaf(qf[k], kp[k], s);
        }
        s = slot_sum<LP>(s) * a.scale;
        sc[p] = ((vis >> p) & 1u) ? s : -__builtin_huge_valf();
        m = fmaxf(m, sc[p]);
      }
    }
    float l = 0.0f, acc[EPV];
#pragma unroll
    for (int k = 0; k < EPV; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int p = 0; p < KMAX; ++p) {
      if (p < a.Lk) {
        const float w = ((vis >> p) & 1u) ? __expf(sc[p] - m) : 0.0f;
        const float *vp = Vs + p * DH + i * EPV;
        l += w;
#pragma unroll
        for (int k = 0; k < EPV; ++k) acc[k] = __builtin_fmaf(w, vp[k], acc[k]);
      }
    }
    if (live) {
      const float inv = l > 0.0f ? 1.0f / l : 0.0f;  // a query that sees nothing (padding): zeros
#pragma unroll
      for (int k = 0; k < EPV; ++k) acc[k] *= inv;
      *reinterpret_cast<u32x4 *>(a.out + ((((int64_t)u * a.Lq + t) * a.H + h) * DH + i * EPV) * ES) = pack16v<DT>(acc);
    }
  }
}

template <int DT, int DH>
hipError_t launch_short_rows(const ShortArgs &a, hipStream_t s) {
  const dim3 grid((unsigned)((int64_t)a.U * a.Hkv)), block(64);
  const size_t lds = (size_t)2 * a.Lk * DH * sizeof(float) + (size_t)a.Lq * sizeof(uint32_t);
  if (a.Lk <= 16) hipLaunchKernelGGL((short_attention_rows_kernel<DT, DH, 16>), grid, block, lds, s, a);
  else hipLaunchKernelGGL((short_attention_rows_kernel<DT, DH, 32>), grid, block, lds, s, a);
  return hipGetLastError();
}

template <int DT>
hipError_t launch_short_attention(const ShortArgs &a, int head_dim, hipStream_t s) {
  if (a.Lk <= 32 && a.Lq <= 2048 && (int64_t)a.U * a.Hkv <= 0x7fffffffll) {  // the (row, KV head) form: K / V in LDS once, slots are queries
    if (head_dim == 64) return launch_short_rows<DT, 64>(a, s);
    if (head_dim == 128) return launch_short_rows<DT, 128>(a, s);
    if (head_dim == 32) return launch_short_rows<DT, 32>(a, s);
    if (head_dim == 16) return launch_short_rows<DT, 16>(a, s);
    return hipErrorInvalidValue;
  }
  const dim3 grid((unsigned)((int64_t)a.U * a.H * a.Lq)), block(64);
  if (head_dim == 64) hipLaunchKernelGGL((short_attention_kernel<DT, 64>), grid, block, 0, s, a);
  else if (head_dim == 128) hipLaunchKernelGGL((short_attention_kernel<DT, 128>), grid, block, 0, s, a);
  else if (head_dim == 32) hipLaunchKernelGGL((short_attention_kernel<DT, 32>), grid, block, 0, s, a);
  else if (head_dim == 16) hipLaunchKernelGGL((short_attention_kernel<DT, 16>), grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---- KV slab rows: append, gather (moved here from glb_api.hip in round 5) ---------------------------------------------------

// slab[row_of[i] (or i), h, pos[i], :] = rows[i, h, :]  (one new token per forward row; rows may be a strided view)
template <typename VT>
__global__ void kv_append_kernel(VT *slab, const VT *rows, const int32_t *pos, const int32_t *row_of, int64_t n_rows,
                                 int64_t heads, int64_t cap, int64_t row_vecs, int64_t rows_stride_row,
                                 int64_t rows_stride_head) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_rows * heads * row_vecs) return;
  const int64_t x = gid % row_vecs, h = (gid / row_vecs) % heads, i = gid / (row_vecs * heads);
  const int64_t p = pos[i], r = row_of ? (int64_t)row_of[i] : i;
  if (p < 0 || p >= cap || r < 0) return;
  slab[((r * heads + h) * cap + p) * row_vecs + x] = rows[i * rows_stride_row + h * rows_stride_head + x];
}

// dst[t][i, h, p, :] = src[t][src_row_of[i], h, p, :] for p < len_of[i]; src_row_of[i] < 0 leaves row i alone.
// One launch moves every layer's K and V (pointer tables): fan-out of prompt KV to particles, ancestor gather.
template <typename VT>
__global__ void kv_gather_rows_kernel(const VT *const *src, VT *const *dst, int64_t n_rows, int64_t heads,
                                      int64_t row_vecs, int64_t src_cap, int64_t dst_cap, const int32_t *src_row_of,
                                      const int32_t *len_of) {
  const int t = blockIdx.y;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_row = heads * dst_cap * row_vecs;
  if (gid >= n_rows * per_row) return;
  const int64_t i = gid / per_row, r = gid % per_row;
  const int64_t x = r % row_vecs, p = (r / row_vecs) % dst_cap, h = r / (row_vecs * dst_cap);
  const int64_t sr = src_row_of[i];
  if (sr < 0 || p >= len_of[i] || p >= src_cap) return;
  dst[t][((i * heads + h) * dst_cap + p) * row_vecs + x] = src[t][((sr * heads + h) * src_cap + p) * row_vecs + x];
}

__global__ void gather_rows_i32_kernel(const int32_t *src, int64_t src_ld, const int32_t *row_of, int64_t n,
                                       int64_t width, int32_t *dst, int64_t dst_ld) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * width) return;
  const int64_t i = gid / width, j = gid % width;
  dst[i * dst_ld + j] = src[(int64_t)row_of[i] * src_ld + j];
}


}  // namespace

extern "C" {

int glb_kv_append(void *slab, const void *new_rows, const int32_t *pos, const int32_t *row_of, int64_t n_rows,
                  int64_t heads, int64_t cap, int64_t head_dim, int64_t new_stride_row, int64_t new_stride_head,
                  int32_t elem_bytes, void *stream) {
  if (!slab || !new_rows || !pos) return glb::api_fail(GLB_EINVAL, "null pointer");
  if (n_rows <= 0 || heads <= 0 || cap <= 0 || head_dim <= 0) return glb::api_fail(GLB_EINVAL, "bad sizes");
  if (elem_bytes != 2 && elem_bytes != 4) return glb::api_fail(GLB_EINVAL, "elem_bytes must be 2 or 4");
  const int64_t rowb = head_dim * elem_bytes;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = rowb % 16 == 0 && ((uintptr_t)slab) % 16 == 0 && ((uintptr_t)new_rows) % 16 == 0 &&
                    (new_stride_row * elem_bytes) % 16 == 0 && (new_stride_head * elem_bytes) % 16 == 0;
  if (wide) {
    const int64_t rv = rowb / 16, total = n_rows * heads * rv;
    hipLaunchKernelGGL(kv_append_kernel<uint4>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (uint4 *)slab,
                       (const uint4 *)new_rows, pos, row_of, n_rows, heads, cap, rv, new_stride_row * elem_bytes / 16,
                       new_stride_head * elem_bytes / 16);
  } else if (elem_bytes == 4) {
    const int64_t total = n_rows * heads * head_dim;
    hipLaunchKernelGGL(kv_append_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (uint32_t *)slab,
                       (const uint32_t *)new_rows, pos, row_of, n_rows, heads, cap, head_dim, new_stride_row, new_stride_head);
  } else {
    const int64_t total = n_rows * heads * head_dim;
    hipLaunchKernelGGL(kv_append_kernel<uint16_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (uint16_t *)slab,
                       (const uint16_t *)new_rows, pos, row_of, n_rows, heads, cap, head_dim, new_stride_row, new_stride_head);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "kv_append launch");
  return GLB_OK;
}

int glb_kv_gather_rows(const void *const *src, void *const *dst, int64_t n_tensors, int64_t n_rows, int64_t heads,
                       int64_t head_dim, int64_t src_cap, int64_t dst_cap, const int32_t *src_row_of,
                       const int32_t *len_of, int32_t elem_bytes, void *stream) {
  if (!src || !dst || !src_row_of || !len_of) return glb::api_fail(GLB_EINVAL, "null pointer");
  if (n_tensors <= 0 || n_tensors > 65535 || n_rows <= 0 || heads <= 0 || head_dim <= 0 || src_cap <= 0 || dst_cap <= 0)
    return glb::api_fail(GLB_EINVAL, "bad sizes");
  if (elem_bytes != 2 && elem_bytes != 4) return glb::api_fail(GLB_EINVAL, "elem_bytes must be 2 or 4");
  const int64_t rowb = head_dim * elem_bytes;
  hipStream_t s = (hipStream_t)stream;
  if (rowb % 16 == 0) {  // slabs come from the allocator: 16-byte aligned
    const int64_t rv = rowb / 16, total = n_rows * heads * dst_cap * rv;
    hipLaunchKernelGGL(kv_gather_rows_kernel<uint4>, dim3(blocks_for(total, 256), (unsigned)n_tensors), dim3(256), 0, s,
                       (const uint4 *const *)src, (uint4 *const *)dst, n_rows, heads, rv, src_cap, dst_cap,
                       src_row_of, len_of);
  } else if (elem_bytes == 4) {
    const int64_t total = n_rows * heads * dst_cap * head_dim;
    hipLaunchKernelGGL(kv_gather_rows_kernel<uint32_t>, dim3(blocks_for(total, 256), (unsigned)n_tensors), dim3(256), 0,
                       s, (const uint32_t *const *)src, (uint32_t *const *)dst, n_rows, heads, head_dim, src_cap,
                       dst_cap, src_row_of, len_of);
  } else {
    const int64_t total = n_rows * heads * dst_cap * head_dim;
    hipLaunchKernelGGL(kv_gather_rows_kernel<uint16_t>, dim3(blocks_for(total, 256), (unsigned)n_tensors), dim3(256), 0,
                       s, (const uint16_t *const *)src, (uint16_t *const *)dst, n_rows, heads, head_dim, src_cap,
                       dst_cap, src_row_of, len_of);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "kv_gather_rows launch");
  return GLB_OK;
}

int glb_gather_rows_i32(const int32_t *src, int64_t src_ld, const int32_t *row_of, int64_t n, int64_t width,
                        int32_t *dst, int64_t dst_ld, void *stream) {
  if (!src || !row_of || !dst) return glb::api_fail(GLB_EINVAL, "null pointer");
  if (n <= 0 || width <= 0 || src_ld < width || dst_ld < width) return glb::api_fail(GLB_EINVAL, "bad sizes");
  hipLaunchKernelGGL(gather_rows_i32_kernel, dim3(blocks_for(n * width, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     src_ld, row_of, n, width, dst, dst_ld);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "gather_rows_i32 launch");
  return GLB_OK;
}


int glb_slab_attention(const void *q, int64_t q_stride_row, int64_t q_stride_head, const void *k_new, int64_t k_stride_row,
                       int64_t k_stride_head, const void *v_new, int64_t v_stride_row, int64_t v_stride_head, void *k_slab,
                       void *v_slab, const int32_t *pos,
                       int64_t n_rows, int64_t heads, int64_t kv_heads, int64_t cap, int64_t head_dim, float scale,
                       int32_t dtype, void *out, void *stream) {
  if (!q || !k_new || !v_new || !k_slab || !v_slab || !pos || !out) return glb::api_fail(GLB_EINVAL, "glb_slab_attention: null pointer");
  if (dtype < GLB_F32 || dtype > GLB_F16) return glb::api_fail(GLB_EINVAL, "bad dtype %d", dtype);
  if (n_rows <= 0 || heads <= 0 || kv_heads <= 0 || cap <= 0 || heads % kv_heads || n_rows * heads > 0x7fffffffll)
    return glb::api_fail(GLB_EINVAL, "glb_slab_attention: bad sizes");
  if (head_dim != 16 && head_dim != 32 && head_dim != 64 && head_dim != 128)
    return glb::api_fail(GLB_EUNSUPPORTED, "glb_slab_attention: head_dim %lld (16, 32, 64 and 128 are built)", (long long)head_dim);
  const int es = dtype == GLB_F32 ? 4 : 2;
  if (((uintptr_t)q | (uintptr_t)k_new | (uintptr_t)v_new | (uintptr_t)k_slab | (uintptr_t)v_slab | (uintptr_t)out) % 16 ||
      (q_stride_row * es) % 16 || (q_stride_head * es) % 16 || (k_stride_row * es) % 16 || (k_stride_head * es) % 16 ||
      (v_stride_row * es) % 16 || (v_stride_head * es) % 16)
    return glb::api_fail(GLB_EINVAL, "glb_slab_attention: pointers and strides must be 16-byte aligned");
  AttnArgs a{};
  a.q = (const char *)q;
  a.k_new = (const char *)k_new;
  a.v_new = (const char *)v_new;
  a.k_slab = (char *)k_slab;
  a.v_slab = (char *)v_slab;
  a.out = (char *)out;
  a.pos = pos;
  a.q_sr = q_stride_row;
  a.q_sh = q_stride_head;
  a.k_sr = k_stride_row;
  a.k_sh = k_stride_head;
  a.v_sr = v_stride_row;
  a.v_sh = v_stride_head;
  a.R = (int32_t)n_rows;
  a.H = (int32_t)heads;
  a.Hkv = (int32_t)kv_heads;
  a.cap = (int32_t)cap;
  a.scale = scale;
  hipError_t e;
  switch (dtype) {
    case GLB_F32: e = launch_attention<GLB_F32>(a, (int)head_dim, (hipStream_t)stream); break;
    case GLB_BF16: e = launch_attention<GLB_BF16>(a, (int)head_dim, (hipStream_t)stream); break;
    default: e = launch_attention<GLB_F16>(a, (int)head_dim, (hipStream_t)stream); break;
  }
  if (e != hipSuccess) return glb::api_hip_fail(e, "slab_attention launch");
  return GLB_OK;
}

int glb_match_rows(const int32_t *tokens, const int64_t *starts, const int32_t *lengths, const int32_t *rep,
                   const int32_t *n_groups, int64_t n, const int32_t *row_tok, const int32_t *row_len,
                   const uint64_t *row_hash, int64_t n_rows, int64_t cap, int32_t *out_old, uint64_t *out_hash,
                   void *stream) {
  if (!tokens || !starts || !lengths || !rep || !n_groups || !row_tok || !row_len || !row_hash || !out_old)
    return glb::api_fail(GLB_EINVAL, "glb_match_rows: null pointer");
  if (n <= 0 || n_rows <= 0 || cap <= 0 || n > 0x7fffffffll || n_rows > 0x7fffffffll || cap > 0x7fffffffll)
    return glb::api_fail(GLB_EINVAL, "glb_match_rows: bad sizes");
  hipLaunchKernelGGL(match_rows_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, tokens, starts, lengths, rep,
                     n_groups, row_tok, row_len, row_hash, (int32_t)n_rows, (int32_t)cap, out_old, out_hash);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "match_rows launch");
  return GLB_OK;
}

size_t glb_kv_plan_workspace(int64_t n, int64_t n_rows) {
  if (n <= 0 || n_rows <= 0) return 0;
  return (size_t)(2 * n_rows + n) * sizeof(int32_t) + 256;
}

int glb_kv_plan(const glb_kv_plan_args *a, void *stream) {
  if (!a) return glb::api_fail(GLB_EINVAL, "glb_kv_plan: null args");
  if (a->struct_size != sizeof(glb_kv_plan_args))
    return glb::api_fail(GLB_EINVAL, "glb_kv_plan_args.struct_size %u != %zu (ABI mismatch)", a->struct_size, sizeof(glb_kv_plan_args));
  if (a->n <= 0 || a->n_rows <= 0 || a->cap <= 0 || a->n > 0x7fffffffll || a->n_rows > 0x7fffffffll || a->cap > 0x7fffffffll)
    return glb::api_fail(GLB_EINVAL, "glb_kv_plan: bad sizes");
  if (!a->group_of || !a->rep || !a->n_groups || !a->old_row || !a->lengths)
    return glb::api_fail(GLB_EINVAL, "glb_kv_plan: null input");
  if (!a->out_group_row || !a->out_logits_row || !a->out_rows_a || !a->out_ctx_a || !a->out_pos_a || !a->out_ctx_b ||
      !a->out_rows_b || !a->out_copy_src || !a->out_copy_len || !a->out_ctx_of_row || !a->out_pos_of_row || !a->out_head)
    return glb::api_fail(GLB_EINVAL, "glb_kv_plan: null output");
  if (a->row_tok && (!a->row_len || !a->row_hash || !a->group_hash || !a->tokens || !a->starts))
    return glb::api_fail(GLB_EINVAL, "glb_kv_plan: row table given without its columns / the contexts");
  if (!a->workspace || ((uintptr_t)a->workspace) % 4 || a->workspace_bytes < glb_kv_plan_workspace(a->n, a->n_rows))
    return glb::api_fail(GLB_ENOSPC, "glb_kv_plan: workspace %zu < %zu bytes", a->workspace_bytes, glb_kv_plan_workspace(a->n, a->n_rows));
  PlanArgs p{};
  p.n = (int32_t)a->n;
  p.R = (int32_t)a->n_rows;
  p.cap = (int32_t)a->cap;
  p.group_of = a->group_of;
  p.rep = a->rep;
  p.n_groups = a->n_groups;
  p.old_src = a->old_row;
  p.old_sel = a->old_row_by_context ? a->rep : nullptr;
  p.lengths = a->lengths;
  p.stamps = a->row_stamps;
  p.call_no = a->call_no;
  p.grp_row = a->out_group_row;
  p.inv = a->out_logits_row;
  p.rows_a = a->out_rows_a;
  p.reps_a = a->out_ctx_a;
  p.pos_a = a->out_pos_a;
  p.reps_b = a->out_ctx_b;
  p.rows_b = a->out_rows_b;
  p.copy_src = a->out_copy_src;
  p.copy_len = a->out_copy_len;
  p.ctx_of_row = a->out_ctx_of_row;
  p.pos_full = a->out_pos_of_row;
  p.row_of_new = a->out_row_of_context;
  p.head = a->out_head;
  p.row_tok = a->row_tok;
  p.row_len = a->row_len;
  p.row_hash = a->row_hash;
  p.grp_hash = a->group_hash;
  p.tok = a->tokens;
  p.st = a->starts;
  int32_t *ws = (int32_t *)a->workspace;
  p.keeper = ws;
  p.free_by_rank = ws + a->n_rows;
  p.flags = ws + 2 * a->n_rows;
  // (the ages of up to 15 360 rows fit the 64 KB of LDS a launch gets without asking for more; clamping a stamp's age is
  // exact for the order as long as calls are fewer than 2^31 apart)
  p.lds_rows = a->row_stamps && a->n_rows <= 15360 ? (int32_t)a->n_rows : 0;
  hipLaunchKernelGGL(kv_plan_kernel, dim3(1), dim3(kT), (size_t)p.lds_rows * sizeof(int), (hipStream_t)stream, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "kv_plan launch");
  return GLB_OK;
}

int glb_short_attention(const void *q, const int64_t q_strides[3], const void *k, const int64_t k_strides[3], const void *v,
                        const int64_t v_strides[3], const uint8_t *mask, int64_t mask_stride_row, int64_t mask_stride_query,
                        int64_t n_rows, int64_t heads, int64_t kv_heads, int64_t q_len, int64_t k_len, int64_t head_dim,
                        float scale, int32_t dtype, void *out, void *stream) {
  if (!q || !k || !v || !out || !q_strides || !k_strides || !v_strides) return glb::api_fail(GLB_EINVAL, "glb_short_attention: null pointer");
  if (dtype < GLB_F32 || dtype > GLB_F16) return glb::api_fail(GLB_EINVAL, "bad dtype %d", dtype);
  if (n_rows <= 0 || heads <= 0 || kv_heads <= 0 || q_len <= 0 || k_len <= 0 || heads % kv_heads || k_len < q_len ||
      n_rows * heads * q_len > 0x7fffffffll || k_len > 0x7fffffffll)
    return glb::api_fail(GLB_EINVAL, "glb_short_attention: bad sizes");
  if (head_dim != 16 && head_dim != 32 && head_dim != 64 && head_dim != 128)
    return glb::api_fail(GLB_EUNSUPPORTED, "glb_short_attention: head_dim %lld (16, 32, 64 and 128 are built)", (long long)head_dim);
  const int es = dtype == GLB_F32 ? 4 : 2;
  bool aligned = (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16) == 0;
  for (int d = 0; d < 3; ++d) aligned = aligned && (q_strides[d] * es) % 16 == 0 && (k_strides[d] * es) % 16 == 0 && (v_strides[d] * es) % 16 == 0;
  if (!aligned) return glb::api_fail(GLB_EINVAL, "glb_short_attention: pointers and strides must be 16-byte aligned");
  ShortArgs a{};
  a.q = (const char *)q;
  a.k = (const char *)k;
  a.v = (const char *)v;
  a.mask = mask;
  a.out = (char *)out;
  a.q_sr = q_strides[0], a.q_sh = q_strides[1], a.q_sp = q_strides[2];
  a.k_sr = k_strides[0], a.k_sh = k_strides[1], a.k_sp = k_strides[2];
  a.v_sr = v_strides[0], a.v_sh = v_strides[1], a.v_sp = v_strides[2];
  a.m_sr = mask_stride_row;
  a.m_sq = mask_stride_query;
  a.U = (int32_t)n_rows;
  a.H = (int32_t)heads;
  a.Hkv = (int32_t)kv_heads;
  a.Lq = (int32_t)q_len;
  a.Lk = (int32_t)k_len;
  a.scale = scale;
  hipError_t e;
  switch (dtype) {
    case GLB_F32: e = launch_short_attention<GLB_F32>(a, (int)head_dim, (hipStream_t)stream); break;
    case GLB_BF16: e = launch_short_attention<GLB_BF16>(a, (int)head_dim, (hipStream_t)stream); break;
    default: e = launch_short_attention<GLB_F16>(a, (int)head_dim, (hipStream_t)stream); break;
  }
  if (e != hipSuccess) return glb::api_hip_fail(e, "short_attention launch");
  return GLB_OK;
}

}  // extern "C"
