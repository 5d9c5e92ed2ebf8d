// glb_trie.hip — token -> byte trie masses with ONE ROW of a part of the trie resident in LDS (include/glb.h:
// glb_trie_rows; replaces trie/base.py:346-393's per-row loops and trie/parallel.py:92-145's sparse product for
// batches).
//
// The level-synchronous kernels of glb_api.hip keep the values node-major in HBM (scr[slot][row]): the weights pass
// through a transpose (206 MB in, 206 MB out in 256-byte runs at 1024 x 50257), every slot is read once more as a child
// and the result is transposed back for a row-major caller - 750 MB of scattered traffic at ~3 TB/s for 478 MB of
// algorithmic bytes, seven launches.  Here a workgroup owns (row, part): the host cuts the folded trie into parts of
// subtrees that fit the LDS (trie.TokenByteTrie.plan), numbered breadth first so that a node's children are
// consecutive and a depth is a range.  The workgroup gathers its part's token weights from the row (tokens ascending:
// the row is read line by line; the parts of a row run on the same XCD, so the row comes from HBM once and from that
// XCD's L2 afterwards), reduces depth by depth in LDS - a thread per node, children added in ascending order in double,
// the node stored as float32: the arithmetic and the order of the reference's loop - and writes its run of the row of
// the output.  The few nodes above the parts (the root and what else has more descendants than a part holds) are a
// last small part whose leaves are the parts' subtree roots, one more launch of the same kernel.
// Traffic: the weights once, the output once.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

#include "../../include/glb.h"
#include "glb_common.hpp"
#include "glb_diag.hpp"
#include "glb_trie.hpp"

namespace {

constexpr int kDesc = 16;  // int32 words per part descriptor (trie.py: plan()["desc"])
enum { D_SLOT_BASE = 0, D_N_LOCAL, D_N_ROOTS, D_N_DEPTHS, D_DEPTH_OFF, D_CPTR_OFF, D_LEAF_OFF, D_N_LEAVES, D_CUT_BASE, D_NODE_OFF, D_N_NODES,
       D_INODE_OFF, D_N_INODES, D_IDEPTH_OFF, D_RUN_OFF, D_N_RUNS };
constexpr int kMaxThreads = 512;

struct TrieRowsParams {
  const void *ws;
  int64_t ld;
  int32_t n_rows, n_parts, from_logprobs, op;
  const float *lse;
  float scale;
  const int32_t *desc, *idepth, *leaf_src, *leaf_local, *run_tab, *top_local;
  const uint16_t *cptr16, *inode16, *pn_local16;
  const uint16_t *tok_local16;  // sweep plans: [n_parts][vocab_pad] the token's local slot in the part (another part's: a word of the slack, n_local + 0..31)
  const uint64_t *inode64;      // sweep plans: the internal nodes as inode16 lists them: slot | first child << 16 | children << 32
  int32_t vocab, vocab_pad;

  int32_t top_base, n_cut;
  float *cut_vals;  // [n_rows][n_cut]: the values of the parts' subtree roots, the leaves of the top
  float *out_slots;
  int64_t out_slots_ld;
  float *out_nodes;
  int64_t out_nodes_ld;
  const int32_t *sel_slot;  // [n_sel]: the slots of the selected nodes (workspace); per-row selections: [n_rows][n_sel]
  int64_t sel_stride;       // 0: one selection for every row; n_sel: a selection per row
  const uint64_t *need;     // per-row selections: bit p of need[r] = row r needs part p (bit 63: the top); else null
  int32_t n_sel;
  float *out_sel;
  int64_t out_sel_ld;
  GLB_DIAG(uint64_t *stamps;)  // diagnostic build (tools/dbg/stamps_trie.py): per workgroup [start, leaves in flight, leaves in LDS, reduced, written]
};

// LDS of a part: val[n_local] float32, then as 16-bit words (a part has fewer than 65536 slots) the part's child
// pointers and its internal nodes, depth by depth: the reduction touches no global memory, and a thread gets an
// INTERNAL node (three slots in four are leaves; dealt by slot, every wave would run a node's loop for a handful of
// busy lanes - 6.6 us of a workgroup's 17; within a depth the host lists the nodes with the most children first, so the
// lanes of a wave run loops of about the same length).  Everything that reads an index from global memory and then uses
// it (token -> weight, node -> output position) is unrolled with the index loads in front: one memory latency per 4-16
// elements instead of one or two per element.

struct PartView {  // one part as the workgroup sees it
  const int32_t *d;  // its descriptor (workgroup-uniform: scalar loads)
  int n_local, n_depths, n_leaves, n_inodes, cp_words, in_words;
  __device__ explicit PartView(const int32_t *desc)
      : d(desc), n_local(desc[D_N_LOCAL]), n_depths(desc[D_N_DEPTHS]), n_leaves(desc[D_N_LEAVES]), n_inodes(desc[D_N_INODES]),
        cp_words((desc[D_N_LOCAL] + 2) >> 1), in_words((desc[D_N_INODES] + 1) >> 1) {}  // 32-bit words of the two 16-bit tables
};

// The part's tables: child pointers and internal nodes (two 16-bit entries a word; both start on a word in global
// memory), then the depth table (n_depths + 1 words; the plan keeps n_depths <= 30 and counts 32 words for it) - one run
// of words in LDS behind the values.
__device__ __forceinline__ uint32_t part_table_word(const TrieRowsParams &p, const PartView &v, int i) {
  const uint32_t *g = reinterpret_cast<const uint32_t *>(p.cptr16 + v.d[D_CPTR_OFF]);
  const uint32_t *gi = reinterpret_cast<const uint32_t *>(p.inode16 + v.d[D_INODE_OFF]);
  const uint32_t *gd = reinterpret_cast<const uint32_t *>(p.idepth + v.d[D_IDEPTH_OFF]);
  const int a = v.cp_words, b = a + v.in_words;
  const uint32_t *q = i < a ? g + i : (i < b ? gi + (i - a) : gd + (i - b));
  return *q;
}

// depth by depth, deepest first: a thread per internal node, its children consecutive in LDS, added in ascending order
// in double, the node stored as float32.  Ends on a barrier.
__device__ __forceinline__ void part_reduce(const TrieRowsParams &p, const PartView &v, float *val, const uint32_t *tab, int tid, int nt) {
  const uint16_t *cp16 = reinterpret_cast<const uint16_t *>(tab), *in16 = reinterpret_cast<const uint16_t *>(tab + v.cp_words);
  const uint32_t *idp = tab + v.cp_words + v.in_words;  // (in LDS: a scalar load from global memory per depth cost the workgroup a memory latency each)
  for (int k = v.n_depths - 2; k >= 0; --k) {  // (the deepest depth holds leaves only)
    const int lo = idp[k], hi = idp[k + 1];
    for (int i = lo + tid; i < hi; i += nt) {
      const int s = in16[i], c0 = cp16[s], c1 = cp16[s + 1];
      double acc = 0.0;
      if (p.op == GLB_TRIE_SUM) {
        int c = c0;
        for (; c + 8 <= c1; c += 8) {  // eight loads in flight; added one by one in ascending order all the same
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = val[c + j];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += (double)x[j];
        }
        if (c < c1) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 7; ++j) x[j] = val[c + j < c1 ? c + j : c];
#pragma unroll
          for (int j = 0; j < 7; ++j)
            if (c + j < c1) acc += (double)x[j];
        }
      } else {
        for (int c = c0; c < c1; ++c) acc = fmax(acc, (double)val[c]);
      }
      val[s] = (float)acc;
    }
    __syncthreads();
  }
}

// the part's values out: its run of the slot-major row, the trie nodes it holds, the selected nodes that are its own
// kW: 8-byte reads of node lists / selected slots a thread has in flight; OUT: the outputs compiled in (1 slots, 2 nodes,
// 4 selected) - the sweep kernel has few registers to spare and is instantiated per kind of output
template <int kW = 4, int OUT = 7>
__device__ __forceinline__ void part_write(const TrieRowsParams &p, const PartView &v, const float *val, bool top, int r, int tid, int nt) {
  if ((OUT & 1) && p.out_slots) {
    float *o = p.out_slots + (int64_t)r * p.out_slots_ld;
    if (top) {
      const int n_top = v.n_local - v.n_leaves;
      for (int i = tid; i < n_top; i += nt) o[p.top_base + i] = val[p.top_local[i]];
    } else {
      o += v.d[D_SLOT_BASE];
      typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // (the run of a row starts on any word)
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const int n4 = v.n_local & ~3;
      for (int i = 4 * tid; i < n4; i += 4 * nt) *reinterpret_cast<f32x4_u *>(o + i) = *reinterpret_cast<const f32x4_t *>(val + i);
      for (int i = n4 + tid; i < v.n_local; i += nt) o[i] = val[i];
    }
  }
  if ((OUT & 2) && p.out_nodes) {
    // the part's nodes are a few runs of consecutive ids (a subtree is an interval of the post-order numbering, the
    // one-child nodes folded into its root follow it): consecutive stores, one 16-bit local slot read per node
    float *o = p.out_nodes + (int64_t)r * p.out_nodes_ld;
    const uint16_t *nl = p.pn_local16 + v.d[D_NODE_OFF];
    const int32_t *runs = p.run_tab + 2 * v.d[D_RUN_OFF];
    const int n_runs = v.d[D_N_RUNS];
    typedef uint64_t u64_u __attribute__((aligned(2)));
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    int at = 0;
    for (int k = 0; k < n_runs; ++k) {  // (workgroup-uniform)
      const int lo = runs[2 * k], cnt = runs[2 * k + 1], cnt4 = cnt & ~3;
      // four nodes a lane: one 8-byte read of local slots, one 16-byte store (1 KB of consecutive output a wave instruction)
      for (int i0 = 4 * tid; i0 < cnt4; i0 += 4 * nt * kW) {
        uint64_t q[kW];
#pragma unroll
        for (int j = 0; j < kW; ++j) q[j] = i0 + j * 4 * nt < cnt4 ? *reinterpret_cast<const u64_u *>(nl + at + i0 + j * 4 * nt) : 0ull;
#pragma unroll
        for (int j = 0; j < kW; ++j) {
          const int i = i0 + j * 4 * nt;
          if (i < cnt4) {
            const f32x4_u x{val[q[j] & 0xffffu], val[(q[j] >> 16) & 0xffffu], val[(q[j] >> 32) & 0xffffu], val[q[j] >> 48]};
            *reinterpret_cast<f32x4_u *>(o + lo + i) = x;
          }
        }
      }
      for (int i = cnt4 + tid; i < cnt; i += nt) o[lo + i] = val[nl[at + i]];
      at += cnt;
    }
  }
  if ((OUT & 4) && p.out_sel) {
    float *o = p.out_sel + (int64_t)r * p.out_sel_ld;
    const int base = v.d[D_SLOT_BASE];
    const int32_t *sel_slot = p.sel_slot + (int64_t)r * p.sel_stride;
    for (int j0 = tid; j0 < p.n_sel; j0 += nt * kW) {
      int sl[kW];
#pragma unroll
      for (int j = 0; j < kW; ++j) {
        const int jj = j0 + j * nt;
        sl[j] = jj < p.n_sel ? sel_slot[jj] : -1;
      }
#pragma unroll
      for (int j = 0; j < kW; ++j) {
        const int jj = j0 + j * nt, s = sl[j];
        if (top) {
          if (s >= base) o[jj] = val[p.top_local[s - base]];
        } else {
          if (s >= base && s < base + v.n_local) o[jj] = val[s - base];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// (row, part) per workgroup.  TOP: the part above the cut (leaves = the cut roots' values of the row, a second launch);
// otherwise block b = (row, part) with the parts of a row on consecutive blocks of the SAME XCD (blocks go round the
// eight XCDs one by one): the row comes from HBM once and from that XCD's L2 for its other parts.
// Measured and not kept (tools/dbg/stamps_trie.py, 1024 x 50257, profiles/r04/stamps_trie_*): the whole row in a
// workgroup's registers and the parts through its LDS one after the other (one workgroup a CU, every latency exposed:
// 79 us a row); a part reading the whole row in one coalesced sweep and picking its tokens by a mask (more trips to
// memory than the gathers: 14 against 8-10 us); two rows a workgroup sharing tables, indices and the reduction (the
// values of two rows halve the workgroups a CU holds: 257 against 206 us).
// ---------------------------------------------------------------------------------------------------------------------
template <int DT, bool TOP, int kL>  // kL: leaves (and table words) a thread has in flight
__global__ __launch_bounds__(kMaxThreads, 6) void trie_rows_kernel(TrieRowsParams p) {
  extern __shared__ float val[];
  int r, part;
  if constexpr (TOP) {
    r = blockIdx.x;
    part = p.n_parts;
  } else {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    part = q % p.n_parts;
    r = (q / p.n_parts) * 8 + xcd;
    if (r >= p.n_rows) return;
  }
  if (p.need && !((p.need[r] >> (TOP ? 63 : part)) & 1ull)) return;  // (per-row selections: this row asks nothing of this part)
  const PartView v(p.desc + part * kDesc);
  const int tid = threadIdx.x, nt = blockDim.x;
  GLB_DIAG(uint64_t *st = (!TOP && p.stamps && blockIdx.x < 65536u) ? p.stamps + (int64_t)blockIdx.x * 8 : nullptr;)
#define GLB_TRIE_STAMP(k) GLB_DIAG(if (st && tid == 0) st[k] = GLB_NOW();)
  GLB_TRIE_STAMP(0)
  uint32_t *tab = reinterpret_cast<uint32_t *>(val + v.n_local);
  // ---- tables and leaves, two memory latencies in all: the table words and the leaves' indices (token, local slot) go
  // out together; the weights the indices name follow; the table words go to LDS while those are on their way.  (One
  // leaf at a time and the tables behind them, a workgroup spent 8-10 of its 17 us here: every trip to memory takes
  // 1.5-2.5 us with the chip full of these workgroups.)
  {
    constexpr int kT = kL;
    const int tot_words = v.cp_words + v.in_words + v.n_depths + 1;
    uint32_t tw[kT];
#pragma unroll
    for (int j = 0; j < kT; ++j) tw[j] = tid + j * nt < tot_words ? part_table_word(p, v, tid + j * nt) : 0u;
    const int32_t *src = p.leaf_src + v.d[D_LEAF_OFF], *loc = p.leaf_local + v.d[D_LEAF_OFF];
    const float *cv = TOP ? p.cut_vals + (int64_t)r * p.n_cut : nullptr;
    const float lse = (!TOP && p.lse) ? p.lse[r] : 0.0f;
    const int64_t row = (int64_t)r * p.ld;
    for (int i0 = tid; i0 < v.n_leaves || i0 == tid; i0 += nt * kL) {
      int sj[kL], lj[kL];
      float w[kL];
#pragma unroll
      for (int j = 0; j < kL; ++j) {
        const int i = i0 + j * nt;
        const bool ok = i < v.n_leaves;
        sj[j] = ok ? src[i] : -1;
        lj[j] = ok ? loc[i] : 0;
      }
#pragma unroll
      for (int j = 0; j < kL; ++j) {
        w[j] = 0.0f;
        if (sj[j] >= 0) {
          if constexpr (TOP) w[j] = cv[sj[j]];
          else w[j] = glb::trie_weight_load<DT>(p.ws, row + sj[j]);
        }
      }
      if (i0 == tid) {  // (first trip: the table words have arrived with the indices)
#pragma unroll
        for (int j = 0; j < kT; ++j)
          if (tid + j * nt < tot_words) tab[tid + j * nt] = tw[j];
        for (int i = tid + kT * nt; i < tot_words; i += nt) tab[i] = part_table_word(p, v, i);  // (a part too big for kT words a thread)
      }
#pragma unroll
      for (int j = 0; j < kL; ++j)
        if (sj[j] >= 0) val[lj[j]] = TOP ? w[j] : glb::trie_weight_value(w[j], p.from_logprobs, p.scale, lse);
    }
  }
  GLB_TRIE_STAMP(1)
  __syncthreads();
  GLB_TRIE_STAMP(2)
  part_reduce(p, v, val, tab, tid, nt);
  GLB_TRIE_STAMP(3)
  part_write(p, v, val, TOP, r, tid, nt);
  if constexpr (!TOP) {
    if (p.cut_vals) {
      float *cv = p.cut_vals + (int64_t)r * p.n_cut + v.d[D_CUT_BASE];
      const int n_roots = v.d[D_N_ROOTS];
      for (int i = tid; i < n_roots; i += nt) cv[i] = val[i];
    }
  }
  GLB_DIAG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");)
  GLB_TRIE_STAMP(4)
  GLB_DIAG(if (st && tid == 0) st[5] = ((uint64_t)part << 32) | (uint32_t)r;)
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: a (row, part) is still one workgroup's, but the row is READ FRONT TO BACK instead of gathered.  A gathered part
// touches nearly every 64-byte sector of the row for 1.8 of its tokens, so nine parts ask the L2 for the row nine times
// over, a request per 1.8 tokens: 28 M sector requests a launch at 1024 x 50257, the rate that bounds the kernel above.
// Here only the part's VALUES live in LDS (4 bytes a slot: parts of 40 000 slots, two for the 66 k slots of a 50 k
// vocabulary, one 1024-thread workgroup a CU); the workgroup streams the whole row in 16-byte loads and stores EVERY
// token's weight through `tok_local16[part]` (the token's slot in this part; another part's token: a word of the slack
// behind the values); the part's internal nodes (`inode64`) and the tokens' slots stay in registers from row to row - the
// workgroup is persistent (trie_sweep_kernel below).  Requests per row: parts x the row in 128-byte lines instead of a
// sector per 1.8 tokens per part (20.8 M L1 accesses a call where the gathered kernel makes 103 M).
// ---------------------------------------------------------------------------------------------------------------------
template <int DT>
struct RowVec;
template <>
struct RowVec<GLB_F32> {
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // (a row starts on any word)
  f32x4_u a, b;
  __device__ __forceinline__ void load(const void *ws, int64_t idx) {
    const float *q = reinterpret_cast<const float *>(ws) + idx;
    a = *reinterpret_cast<const f32x4_u *>(q);  // (plain loads: the row's other parts read it from this XCD's L2)
    b = *reinterpret_cast<const f32x4_u *>(q + 4);
  }
  __device__ __forceinline__ float get(int k) const { return k < 4 ? a[k] : b[k - 4]; }
  __device__ __forceinline__ void keep() const { asm volatile("" ::"v"(a), "v"(b)); }  // (a use: the registers stay the vector's own up to here)
};
template <int DT>
struct RowVec {  // 16-bit rows: eight tokens in 16 bytes, on any 2-byte boundary
  typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
  u32x4_u a;
  __device__ __forceinline__ void load(const void *ws, int64_t idx) {
    a = *reinterpret_cast<const u32x4_u *>(reinterpret_cast<const uint16_t *>(ws) + idx);
  }
  __device__ __forceinline__ float get(int k) const { return glb::trie_upcast<DT>((uint16_t)(a[k >> 1] >> ((k & 1) * 16))); }
  __device__ __forceinline__ void keep() const { asm volatile("" ::"v"(a)); }
};

// The reduction of a swept part.  A node: its children are consecutive in LDS, added one by one in ascending order in
// double, the node stored as float32 - the arithmetic and the order of part_reduce.  A depth lists the nodes with the most
// children first, so a wave's nodes have about the same number: the wave takes eight or four children a trip, whatever
// its widest node needs.  A lane reads at most 7 words past its node's last child (inside the part's LDS: 32 words follow
// the values; what it reads there is selected away and +0 added instead, which changes no bit of a sum that started from
// +0); a lane whose node is finished while the wave's widest one is not reads its own first children again, not further
// and further past its node.  OP at compile time: no branch inside a node.
template <int OP>
__device__ __forceinline__ void sweep_node(float *val, uint64_t e) {
  const int s = (int)(e & 0xffffu), c0 = (int)((e >> 16) & 0xffffu), cnt = (int)(e >> 32);
  double acc = 0.0;
  auto add = [&](float x, bool in) {
    if constexpr (OP == GLB_TRIE_SUM) acc += (double)(in ? x : 0.0f);
    else acc = in ? fmax(acc, (double)x) : acc;
  };
  if (!__any(cnt > 4)) {  // the wave's widest node has at most four children: one trip of four
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = val[c0 + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) add(x[j], j < cnt);
  } else {
    // eight children a trip; the NEXT trip's eight are read before this trip's are added (round 6): a trip was an LDS round
    // trip and then a chain of eight dependent double additions, one after the other - for the 256 children of a node near
    // the root 32 times over, and a depth takes as long as its widest node
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = val[c0 + j];
    for (int base = 0;; base += 8) {
      const int rem = cnt - base;
      const bool more = __any(rem > 8);
      float nx[8];
      if (more) {
        const float *q = val + c0 + (rem > 8 ? base + 8 : 0);  // (a finished lane reads its own first children again)
#pragma unroll
        for (int j = 0; j < 8; ++j) nx[j] = q[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) add(x[j], j < rem);
      if (!more) break;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = nx[j];
    }
  }
  val[s] = (float)acc;
}

// depth by depth, deepest internal depth first; trip t of the work list is ent[t] (in registers for the workgroup's life)
template <int OP, int kNT, int kQ>
__device__ __forceinline__ void sweep_reduce(float *val, const uint64_t (&ent)[kQ], int dv, int n_depths, int tid, const uint64_t *in64 GLB_DIAG(, uint64_t *st)) {
  constexpr int nt = kNT;
  int k = n_depths - 2, j = 0;
#pragma unroll
  for (int t = 0; t < kQ; ++t) {
    if (k >= 0) {
      const int lo = __builtin_amdgcn_readlane(dv, k), hi = __builtin_amdgcn_readlane(dv, k + 1);
      if (lo + tid + j * nt < hi) {
        uint64_t e = ent[t];
        asm volatile("" : "+v"(e));  // (unpacked here, row by row, not hoisted out of the rows' loop into three registers a node)
        sweep_node<OP>(val, e);
      }
      if (lo + (j + 1) * nt < hi) {
        ++j;
      } else {
        --k, j = 0;
        __syncthreads();
        GLB_DIAG(if (st && tid == 0 && n_depths - 3 - k < 3) st[5 + n_depths - 3 - k] = GLB_NOW();)
      }
    }
  }
  while (k >= 0) {  // (a part with more than kQ trips: the rest straight from global memory)
    const int lo = __builtin_amdgcn_readlane(dv, k), hi = __builtin_amdgcn_readlane(dv, k + 1);
    for (int i = lo + tid + j * nt; i < hi; i += nt) sweep_node<OP>(val, in64[i]);
    --k, j = 0;
    __syncthreads();
  }
}

// A PERSISTENT workgroup per (part, lane g of n_lanes): rows g, g + n_lanes, ... .  One workgroup fills a CU (the values
// of 33 000 slots), so nothing else hides its trips to memory, and with every CU reading at once a row and its table
// arrive at the chip's rate, 30 GB/s a CU - half a one-row workgroup's life.  What is the same for every row therefore
// stays in registers for the workgroup's life - the tokens' slots (tok_local16: 4 registers an 8-token unit, kU units a
// thread: a 50 k row over 1024 threads) and the first kQ trips of the reduction's work list - and the first kX units of
// the NEXT row are on their way while the current one is reduced and written (128 registers a thread hold no more).
template <int DT, int kNT, int OUT>
__global__ __launch_bounds__(kNT) void trie_sweep_kernel(TrieRowsParams p, int n_lanes, int stagger) {
  extern __shared__ float val[];
  const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
  const int part = q % p.n_parts;
  const int g = (q / p.n_parts) * 8 + xcd;  // (the parts of a row on the same XCD, side by side in time: the row comes from HBM once)
  if (g >= n_lanes) return;
  const PartView v(p.desc + part * kDesc);
  const int tid = threadIdx.x;
  constexpr int nt = kNT;
  constexpr int kU = 7;   // units whose slots a thread keeps in registers
  constexpr int kX = DT == GLB_F32 ? 0 : 4;  // units of the next row a thread has in flight during the reduction (16-bit rows: 4
                                             // registers a unit; a float32 row's 8 do not fit next to the slots and the work list)
  constexpr int kB = 3;                      // units a trip for the rest
  constexpr int kQ = 12;  // trips of the reduction's work list a thread keeps in registers (33 000 slots, 1024 threads: 11)
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const uint16_t *tl = p.tok_local16 + (int64_t)part * p.vocab_pad;
  const int full = p.vocab >> 3;  // units of 8 tokens that lie inside the row
  // -- what every row shares
  // the reduction's work list: depth k (deepest internal depth first), trip j -> node lo_k + tid + j * nt of the part's list;
  // the depth table (<= 31 words) sits in a register, a word a lane, and the walk over (k, j) is scalar
  const int32_t *gdp = p.idepth + v.d[D_IDEPTH_OFF];
  const int lane = tid & 63;
  const int dv = lane <= v.n_depths ? gdp[lane] : 0;
  const uint64_t *in64 = p.inode64 + v.d[D_INODE_OFF];
  uint64_t ent[kQ];
  {
    int k = v.n_depths - 2, j = 0;
#pragma unroll
    for (int t = 0; t < kQ; ++t) {
      uint64_t e = 0ull;
      if (k >= 0) {
        const int lo = __builtin_amdgcn_readlane(dv, k), hi = __builtin_amdgcn_readlane(dv, k + 1);
        const int i = lo + tid + j * nt;
        if (i < hi) e = in64[i];
        if (lo + (j + 1) * nt < hi) ++j;
        else --k, j = 0;
      }
      ent[t] = e;
    }
  }
  // A thread's units are tid + j * nt; one past the row's last whole unit becomes that last unit once more (the same
  // weights to the same slots a second time): no load and no store of the rows' loop hangs on a per-lane condition, so the
  // compiler sees straight code and waits for a load where it is used.  (A vocabulary under 8 tokens has no whole unit.)
  const int last = full > 0 ? full - 1 : 0;
  auto uo = [&](int j) { return 8 * (tid + j * nt < full ? tid + j * nt : last); };
  u32x4_t tk[kU];
#pragma unroll
  for (int j = 0; j < kU; ++j) tk[j] = *reinterpret_cast<const u32x4_t *>(tl + uo(j));  // (the table is padded to whole units)
  // (everything above has arrived before the first row sets out: inside the rows' loop the compiler then waits for a row's
  // loads where they are used, not - to be safe about these - for all of memory at every trip of the reduction, which
  // would end the next row's head start)
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  // every token is stored: another part's token names a word of the slack behind the values
  auto put = [&](const u32x4_t &t, const RowVec<DT> &x, float lse) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t loc = (t[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
      val[loc] = glb::trie_weight_value(x.get(k), p.from_logprobs, p.scale, lse);
    }
    __builtin_amdgcn_sched_barrier(0);  // (a unit at a time: scheduled for latency, a trip's 32 weights and addresses take 64 registers)
  };
  // -- the rows
  auto next_row = [&](int r) {  // (per-row selections: a row that asks nothing of this part is skipped)
    while (r < p.n_rows && p.need && !((p.need[r] >> part) & 1ull)) r += n_lanes;
    return r;
  };
  // Every workgroup starts at once and takes the same time a row, so all of them would read at the same time and reduce at
  // the same time: the reads at 30 GB/s a CU (the chip's rate over 256), the memory idle in between.  Every other group of
  // eight lanes therefore waits half a row's time after its first row (once; `stagger`: the host asks for it when a lane
  // has rows enough to win it back): from then on half the chip reads while the other half reduces - 50 GB/s a CU.
  bool wait_half = stagger && ((g >> 3) & 1);
  const uint64_t t_first = __builtin_amdgcn_s_memrealtime();
  // A trip of the loop: the head of row r sets out; the row BEFORE it, which is in LDS, is reduced and written; row r goes
  // to LDS.  A load is issued and used in the same trip, with the reduction in between - nothing is carried round the loop
  // in flight, which the compiler answers with waits for all of memory in the middle of the reduction.
  int r = next_row(g), r_prev = -1;
  for (;;) {
    const bool have = r < p.n_rows;
    if (!have && r_prev < 0) break;
    const int64_t row = (int64_t)(have ? r : r_prev) * p.ld;  // (no row left: the loads below read the last one's head again, unused)
    float lse = 0.0f;
    RowVec<DT> x[kX > 0 ? kX : 1];
    if (full > 0) {
      if (p.lse) lse = p.lse[have ? r : r_prev];
#pragma unroll
      for (int j = 0; j < kX; ++j) x[j].load(p.ws, row + uo(j));
    }
    if (r_prev >= 0) {
      GLB_DIAG(uint64_t *st = (p.stamps && r_prev < 8192) ? p.stamps + ((int64_t)r_prev * p.n_parts + part) * 8 : nullptr;)
      GLB_TRIE_STAMP(2)
      if (p.op == GLB_TRIE_SUM) sweep_reduce<GLB_TRIE_SUM, kNT, kQ>(val, ent, dv, v.n_depths, tid, in64 GLB_DIAG(, st));
      else sweep_reduce<GLB_TRIE_MAX, kNT, kQ>(val, ent, dv, v.n_depths, tid, in64 GLB_DIAG(, st));
      GLB_TRIE_STAMP(3)
      part_write<(OUT == 2 ? 4 : 1), OUT>(p, v, val, false, r_prev, tid, nt);  // (all nodes: four node-list reads in flight - with one the write-out waits for the L2 24 times)
      if (p.cut_vals) {
        float *cv = p.cut_vals + (int64_t)r_prev * p.n_cut + v.d[D_CUT_BASE];
        const int n_roots = v.d[D_N_ROOTS];
        for (int i = tid; i < n_roots; i += nt) cv[i] = val[i];
      }
      GLB_TRIE_STAMP(4)
      __syncthreads();  // (the values are read out before the next row lands on them)
      if (wait_half) {
        wait_half = false;
        const uint64_t now = __builtin_amdgcn_s_memrealtime();
        uint64_t half = (now - t_first) >> 1;
        GLB_DIAG(if (stagger > 1) half = ((now - t_first) * (uint64_t)stagger) >> 4;)  // (diagnostic build: GLB_TRIE_STAGGER = sixteenths of a row's time)
        if (half > 2000) half = 2000;  // (20 us at 100 MHz: never a long wait, whatever the clock read)
        while (__builtin_amdgcn_s_memrealtime() - now < half) __builtin_amdgcn_s_sleep(8);
      }
    }
    // (the row's head and its lse own their registers through the reduction on every path: a register with a load out that
    // the reduction could reuse would have to be waited for there)
#pragma unroll
    for (int j = 0; j < kX; ++j) x[j].keep();
    asm volatile("" ::"v"(lse));
    if (!have) break;
    GLB_DIAG(uint64_t *st = (p.stamps && r < 8192) ? p.stamps + ((int64_t)r * p.n_parts + part) * 8 : nullptr;)
    GLB_TRIE_STAMP(0)
    if (full == 0 && p.lse) lse = p.lse[r];
    // the units that came ahead go to LDS while the first trip of the others is on its way
    // (the slots are unpacked here, row by row: hoisted out of the loop they would take two registers a token, not half a one)
    if (full > 0) {
#pragma unroll
      for (int j0 = kX; j0 < kU; j0 += kB) {
        RowVec<DT> y[kB];
#pragma unroll
        for (int j = j0; j < j0 + kB && j < kU; ++j) y[j - j0].load(p.ws, row + uo(j));
        if (j0 == kX) {
#pragma unroll
          for (int j = 0; j < kX; ++j) {
            u32x4_t tj = tk[j];
            asm volatile("" : "+v"(tj));
            put(tj, x[j], lse);
          }
        }
#pragma unroll
        for (int j = j0; j < j0 + kB && j < kU; ++j) {
          u32x4_t tj = tk[j];
          asm volatile("" : "+v"(tj));
          put(tj, y[j - j0], lse);
        }
      }
    }
    // ... the rest of a long row, two units a trip, slots and all ...
    for (int u0 = tid + kU * nt; u0 < full; u0 += nt * 2) {
      u32x4_t t[2];
      RowVec<DT> z[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = u0 + j * nt < full ? u0 + j * nt : last;
        t[j] = *reinterpret_cast<const u32x4_t *>(tl + 8 * u);
        z[j].load(p.ws, row + 8 * u);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) put(t[j], z[j], lse);
    }
    // ... and the last, partial unit element by element: nothing is read past the row
    const int t_tail = 8 * full + tid;
    if (t_tail < p.vocab) val[tl[t_tail]] = glb::trie_weight<DT>(p.ws, row + t_tail, p.from_logprobs, p.scale, lse);
    GLB_TRIE_STAMP(1)
    __syncthreads();
    r_prev = r;
    r = next_row(r + n_lanes);
  }
}

// the selected nodes' slots, once per call (the same for every row)
__global__ void trie_sel_slots_kernel(const int32_t *sel, int32_t n_sel, int64_t n_nodes, const int32_t *slot_of, int32_t *sel_slot) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_sel) sel_slot[j] = (sel[j] >= 0 && sel[j] < n_nodes) ? slot_of[sel[j]] : -1;  // (a node outside the trie: no slot, its output stays untouched)
}

// per-row selections: every row's nodes -> slots (a node outside the trie, or negative: none - its output is 0), and the
// parts the row needs: the part that holds a slot (a part is a union of whole subtrees, so a node's descendants are in its
// own part), every part and the top for a slot above the cut
__global__ __launch_bounds__(256) void trie_sel_rows_kernel(const int32_t *sel, int64_t sel_stride, int32_t n_sel, int64_t n_nodes,
                                                            const int32_t *slot_of, const int32_t *desc, int32_t n_parts,
                                                            int32_t top_base, int32_t has_top, int32_t *sel_slot, uint64_t *need,
                                                            float *out_sel, int64_t out_sel_ld) {
  __shared__ unsigned long long bits;
  const int r = blockIdx.x;
  if (threadIdx.x == 0) bits = 0ull;
  __syncthreads();
  unsigned long long mine = 0ull;
  for (int k = threadIdx.x; k < n_sel; k += blockDim.x) {
    const int32_t node = sel[(int64_t)r * sel_stride + k];
    int32_t slot = (node >= 0 && node < n_nodes) ? slot_of[node] : -1;
    sel_slot[(int64_t)r * n_sel + k] = slot;
    if (slot < 0) {
      out_sel[(int64_t)r * out_sel_ld + k] = 0.0f;
    } else if (has_top && slot >= top_base) {
      mine = ~0ull;
    } else {
      for (int p = 0; p < n_parts; ++p)
        if (slot >= desc[p * kDesc + D_SLOT_BASE] && slot < desc[p * kDesc + D_SLOT_BASE] + desc[p * kDesc + D_N_LOCAL]) mine |= 1ull << p;
    }
  }
  if (mine) atomicOr(&bits, mine);
  __syncthreads();
  if (threadIdx.x == 0) need[r] = bits;
}

template <int DT, int kL>
hipError_t launch_rows1(const TrieRowsParams &p, int n_top, size_t lds, int threads, hipStream_t s) {
  static std::atomic<uint64_t> big_lds{0}, big_lds_top{0};  // (more than 64 KB of dynamic LDS: allowed per kernel and device)
  hipError_t e0 = hipSuccess;
  if (lds > 64 * 1024) {  // (parts of up to 64 KB need no attribute: a build for a device with less LDS still serves them)
    e0 = glb::allow_dynamic_lds((const void *)trie_rows_kernel<DT, false, kL>, 160 * 1024, big_lds);
    if (e0 == hipSuccess) e0 = glb::allow_dynamic_lds((const void *)trie_rows_kernel<DT, true, 8>, 160 * 1024, big_lds_top);
  }
  if (e0 != hipSuccess) return e0;
  const unsigned blocks = (unsigned)(((int64_t)p.n_rows + 7) / 8 * 8 * p.n_parts);
  hipLaunchKernelGGL((trie_rows_kernel<DT, false, kL>), dim3(blocks), dim3(threads), lds, s, p);
  if (n_top > 0) hipLaunchKernelGGL((trie_rows_kernel<DT, true, 8>), dim3((unsigned)p.n_rows), dim3(256), lds, s, p);
  return hipGetLastError();
}

template <int DT, int kNT, int OUT>
hipError_t launch_sweep1(const TrieRowsParams &p, int n_top, size_t lds, size_t lds_top, int wg_per_cu, hipStream_t s) {
  static std::atomic<uint64_t> big_lds{0};
  if (lds > 64 * 1024) {
    hipError_t e0 = glb::allow_dynamic_lds((const void *)trie_sweep_kernel<DT, kNT, OUT>, 160 * 1024, big_lds);
    if (e0 != hipSuccess) return e0;
  }
  // as many persistent workgroups as the chip holds at once, shared out over the parts: n_lanes rows in flight
  static std::atomic<int> n_cus{0};
  int cus = n_cus.load(std::memory_order_relaxed);
  if (cus == 0) {
    int dev = 0;
    hipError_t e1 = hipGetDevice(&dev);
    if (e1 == hipSuccess) e1 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e1 != hipSuccess) return e1;
    if (cus < 8) cus = 8;
    n_cus.store(cus, std::memory_order_relaxed);
  }
  int n_lanes = (cus * wg_per_cu / p.n_parts) & ~7;
  if (n_lanes < 8) n_lanes = 8;
  const int rows8 = (p.n_rows + 7) & ~7;
  if (n_lanes > rows8) n_lanes = rows8;
  const unsigned blocks = (unsigned)(n_lanes * p.n_parts);
  // (where the next row is not on its way during the reduction - float32 rows - and the output is short; five rows a lane and
  // more; every workgroup resident.  Measured at 1024 x 50257: float32 slots 150 -> 142 us; 16-bit rows, whose reads overlap
  // already, and all-nodes outputs, bound by their writes, lose 5-7 %)
  int stagger = DT == GLB_F32 && OUT != 2 && OUT != 7 && p.n_rows >= 5 * n_lanes && blocks <= (unsigned)(cus * wg_per_cu) ? 1 : 0;
  GLB_DIAG(if (const char *ev = getenv("GLB_TRIE_STAGGER")) stagger = atoi(ev);)
  hipLaunchKernelGGL((trie_sweep_kernel<DT, kNT, OUT>), dim3(blocks), dim3(kNT), lds, s, p, n_lanes, stagger);
  if (n_top > 0) hipLaunchKernelGGL((trie_rows_kernel<DT, true, 8>), dim3((unsigned)p.n_rows), dim3(256), lds_top, s, p);
  return hipGetLastError();
}

// a kernel per kind of output (1 slots, 2 all nodes, 4 selected nodes; several at once: the one that writes them all)
template <int DT>
hipError_t launch_sweep(const TrieRowsParams &p, int n_top, size_t lds, size_t lds_top, hipStream_t s) {
  const int out = (p.out_slots ? 1 : 0) | (p.out_nodes ? 2 : 0) | (p.out_sel ? 4 : 0);
  if (out == 1) return launch_sweep1<DT, 1024, 1>(p, n_top, lds, lds_top, 1, s);
  if (out == 2) return launch_sweep1<DT, 1024, 2>(p, n_top, lds, lds_top, 1, s);
  if (out == 4) return launch_sweep1<DT, 1024, 4>(p, n_top, lds, lds_top, 1, s);
  return launch_sweep1<DT, 1024, 7>(p, n_top, lds, lds_top, 1, s);
}

// deep: twelve leaves a thread in flight (73 registers: three 512-thread workgroups a CU, what parts of 9 500 slots allow
// by their LDS; the leaves of such a part in ONE trip); otherwise eight (53 registers).  Sixteen: 93 registers, two
// workgroups a CU - 234 against 206 us at 1024 x 50257.
template <int DT>
hipError_t launch_rows(const TrieRowsParams &p, int n_top, size_t lds, int threads, bool deep, hipStream_t s) {
  return deep ? launch_rows1<DT, 12>(p, n_top, lds, threads, s) : launch_rows1<DT, 8>(p, n_top, lds, threads, s);
}

}  // namespace

constexpr int64_t kMaxSel = 1 << 20;  // selected nodes per call (their slots sit behind the cut values in the workspace)
static size_t cut_bytes(int64_t n_rows, const glb_trie_plan *pl) {
  return pl->n_top > 0 ? (((size_t)n_rows * (size_t)pl->n_cut * sizeof(float) + 255) & ~(size_t)255) : 0;
}

extern "C" {

size_t glb_trie_rows_workspace(int64_t n_rows, const glb_trie_plan *plan) {
  if (!plan || n_rows <= 0) return 0;
  return cut_bytes(n_rows, plan) + kMaxSel * sizeof(int32_t) + (size_t)n_rows * sizeof(uint64_t);
}

int glb_trie_rows(const glb_trie_rows_args *a, const glb_trie_plan *pl, void *stream) {
  using glb::api_fail;
  if (!a || !pl) return api_fail(GLB_EINVAL, "glb_trie_rows: null args");
  if (a->struct_size != sizeof(glb_trie_rows_args))
    return api_fail(GLB_EINVAL, "glb_trie_rows_args.struct_size %u != %zu (ABI mismatch)", a->struct_size, sizeof(glb_trie_rows_args));
  if (pl->struct_size != sizeof(glb_trie_plan))
    return api_fail(GLB_EINVAL, "glb_trie_plan.struct_size %u != %zu (ABI mismatch)", pl->struct_size, sizeof(glb_trie_plan));
  if (a->n_rows == 0) return GLB_OK;
  if (!a->weights || a->n_rows < 0 || a->vocab <= 0 || a->ld < a->vocab)
    return api_fail(GLB_EINVAL, "glb_trie_rows: bad weights shape");
  if (a->n_rows > INT32_MAX / 2 || a->ld * a->n_rows <= 0) return api_fail(GLB_EINVAL, "glb_trie_rows: too many rows");
  if (a->dtype != GLB_F32 && a->dtype != GLB_BF16 && a->dtype != GLB_F16) return api_fail(GLB_EINVAL, "glb_trie_rows: bad dtype");
  if (a->op != GLB_TRIE_SUM && a->op != GLB_TRIE_MAX) return api_fail(GLB_EINVAL, "glb_trie_rows: bad op");
  if (!(a->logit_scale == a->logit_scale)) return api_fail(GLB_EINVAL, "glb_trie_rows: logit_scale is NaN");
  if (!a->out_slots && !a->out_nodes && !(a->out_sel && a->n_sel > 0)) return api_fail(GLB_EINVAL, "glb_trie_rows: no output requested");
  if (pl->n_parts <= 0 || pl->max_local <= 0 || pl->n_slots <= 0 || pl->n_top < 0 || pl->n_cut < 0 || !pl->desc ||
      !pl->idepth || !pl->cptr16 || !pl->inode16 || !pl->leaf_src || !pl->leaf_local || !pl->slot_of)
    return api_fail(GLB_EINVAL, "glb_trie_rows: incomplete plan");
  const size_t lds = (size_t)pl->lds_bytes;
  const bool sweep = pl->tok_local16 != nullptr;  // (a plan for the kernel that reads a row front to back)
  if (sweep && (!pl->inode64 || pl->vocab != a->vocab || pl->lds_top_bytes < 0 || pl->lds_top_bytes > 64 * 1024 || pl->max_local > 65500))
    return api_fail(GLB_EINVAL, "glb_trie_rows: sweep plan without inode64, for another vocabulary (%d, rows of %lld), or with a top over 64 KB",
                    pl->vocab, (long long)a->vocab);
  if (pl->lds_bytes < pl->max_local * (sweep ? 4 : 6) || lds > 160 * 1024 || pl->max_local >= 65536)
    return api_fail(GLB_EINVAL, "glb_trie_rows: a part of %d slots does not fit the LDS", pl->max_local);
  if (a->out_nodes && (!pl->run_tab || !pl->pn_local16)) return api_fail(GLB_EINVAL, "glb_trie_rows: plan without node lists");
  if (pl->n_top > 0 && !pl->top_local) return api_fail(GLB_EINVAL, "glb_trie_rows: plan without top_local");
  if ((a->out_slots && a->out_slots_ld < pl->n_slots) || (a->out_nodes && a->out_nodes_ld < pl->n_nodes) ||
      (a->out_sel && a->out_sel_ld < a->n_sel))
    return api_fail(GLB_EINVAL, "glb_trie_rows: output pitch below its width");
  if (a->n_sel < 0 || a->n_sel > kMaxSel || (a->out_sel && a->n_sel > 0 && !a->sel_nodes))
    return api_fail(GLB_EINVAL, "glb_trie_rows: bad selection");
  const bool per_row = a->sel_row_stride != 0 && a->out_sel && a->n_sel > 0;
  if (per_row && (a->sel_row_stride < a->n_sel || a->n_rows * a->n_sel > kMaxSel || pl->n_parts > 62))
    return api_fail(GLB_EINVAL, "glb_trie_rows: per-row selections need sel_row_stride >= n_sel, n_rows * n_sel <= 2^20 and at most 62 parts");
  const size_t need = glb_trie_rows_workspace(a->n_rows, pl);
  if (need && (!a->workspace || a->workspace_bytes < need))
    return api_fail(GLB_ENOSPC, "glb_trie_rows: %zu bytes of workspace needed (glb_trie_rows_workspace)", need);
  TrieRowsParams p{};
  p.ws = a->weights;
  p.ld = a->ld;
  p.n_rows = (int32_t)a->n_rows;
  p.n_parts = pl->n_parts;
  p.from_logprobs = a->from_logprobs;
  p.op = a->op;
  p.lse = a->lse;
  p.scale = a->from_logprobs ? a->logit_scale : 1.0f;
  p.desc = pl->desc;
  p.idepth = pl->idepth;
  p.cptr16 = pl->cptr16;
  p.inode16 = pl->inode16;
  p.leaf_src = pl->leaf_src;
  p.leaf_local = pl->leaf_local;
  p.run_tab = pl->run_tab;
  p.pn_local16 = pl->pn_local16;
  p.top_local = pl->top_local;
  p.tok_local16 = pl->tok_local16;
  p.inode64 = pl->inode64;
  p.vocab = (int32_t)a->vocab;
  p.vocab_pad = (int32_t)((a->vocab + 7) & ~(int64_t)7);
  p.top_base = pl->top_base;
  p.n_cut = pl->n_cut;
  p.cut_vals = pl->n_top > 0 ? (float *)a->workspace : nullptr;
  p.out_slots = a->out_slots;
  p.out_slots_ld = a->out_slots_ld;
  p.out_nodes = a->out_nodes;
  p.out_nodes_ld = a->out_nodes_ld;
  p.n_sel = a->out_sel ? (int32_t)a->n_sel : 0;
  p.sel_slot = (const int32_t *)((char *)a->workspace + cut_bytes(a->n_rows, pl));
  GLB_DIAG(p.stamps = p.n_sel == 0 ? (uint64_t *)p.sel_slot : nullptr;)  // (the selection's area, when there is no selection)
  p.out_sel = a->n_sel > 0 ? a->out_sel : nullptr;
  p.out_sel_ld = a->out_sel_ld;
  hipStream_t s = (hipStream_t)stream;
  if (per_row) {
    uint64_t *need = (uint64_t *)((char *)a->workspace + cut_bytes(a->n_rows, pl) + kMaxSel * sizeof(int32_t));
    p.sel_stride = p.n_sel;
    p.need = need;
    hipLaunchKernelGGL(trie_sel_rows_kernel, dim3((unsigned)p.n_rows), dim3(256), 0, s, a->sel_nodes, a->sel_row_stride, p.n_sel, pl->n_nodes,
                       pl->slot_of, pl->desc, pl->n_parts, pl->top_base, pl->n_top > 0 ? 1 : 0, (int32_t *)p.sel_slot, need, p.out_sel,
                       p.out_sel_ld);
  } else if (p.out_sel) {
    hipLaunchKernelGGL(trie_sel_slots_kernel, dim3((unsigned)((p.n_sel + 255) / 256)), dim3(256), 0, s, a->sel_nodes, p.n_sel,
                       pl->n_nodes, pl->slot_of, (int32_t *)p.sel_slot);
  }
  // threads per workgroup: as many workgroups as the LDS lets a CU hold.  (1024-thread workgroups were never seen two to
  // a CU, whatever their LDS - tools/dbg/stamps_trie.py; 512-thread ones are.)
  int threads = 512;
  if (lds * 8 <= 160 * 1024) threads = 256;
  bool deep = pl->max_local > 8 * threads;  // (a part's leaves - three slots in four - in one trip of twelve a thread)
  GLB_DIAG(if (const char *ev = getenv("GLB_TRIE_THREADS")) threads = atoi(ev);)
  GLB_DIAG(if (const char *ev = getenv("GLB_TRIE_DEEP")) deep = atoi(ev) != 0;)
  hipError_t e;
  if (sweep) {
    const size_t lds_top = (size_t)pl->lds_top_bytes;
    switch (a->dtype) {
      case GLB_F32: e = launch_sweep<GLB_F32>(p, pl->n_top, lds, lds_top, s); break;
      case GLB_BF16: e = launch_sweep<GLB_BF16>(p, pl->n_top, lds, lds_top, s); break;
      default: e = launch_sweep<GLB_F16>(p, pl->n_top, lds, lds_top, s); break;
    }
    if (e != hipSuccess) return glb::api_hip_fail(e, "glb_trie_rows launch");
    return GLB_OK;
  }
  switch (a->dtype) {
    case GLB_F32: e = launch_rows<GLB_F32>(p, pl->n_top, lds, threads, deep, s); break;
    case GLB_BF16: e = launch_rows<GLB_BF16>(p, pl->n_top, lds, threads, deep, s); break;
    default: e = launch_rows<GLB_F16>(p, pl->n_top, lds, threads, deep, s); break;
  }
  if (e != hipSuccess) return glb::api_hip_fail(e, "glb_trie_rows launch");
  return GLB_OK;
}

}  // extern "C"
