// glb_trie_levels.hip - the LEVEL-SYNCHRONOUS token -> byte trie kernels (include/glb.h: glb_trie_masses, glb_trie_reduce).
// Round 4's glb_trie_rows (glb_trie.hip: a row of a part of the folded trie resident in LDS) serves every row-major
// result; these remain for what it does not: node-major results (`keep_node_major`, scr[node][row]), tries without a plan
// (a node with more children than a part of the LDS holds) and the reference-shaped `glb_trie_reduce` entry point.  They
// live in a translation unit of their own so that the product's main file holds the hot path only.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/glb.h"
#include "glb_common.hpp"
#include "glb_trie.hpp"

namespace {

// Token->byte trie masses (trie/base.py:346-393), level-synchronous over ALL weight rows at once: one launch puts the
// token weights on the leaves, then one launch per tree level (deepest first) gives every node of that level the sum
// (or the maximum, floored at 0 as in the reference) of its children - taken in ascending child order and accumulated
// in double, the reference's sequential order - and stores it as float32.  A thread is one (row, node); node ids are
// post-order, so the children of neighbouring nodes are neighbours in memory.  Values live in the output itself: no
// scratch, every output float written once and read once.
// (the leaf weight - trie_weight, trie_exp - lives in glb_trie.hpp: the row-resident kernel of glb_trie.hip shares it)
using glb::trie_weight;

template <int DT>
__global__ void trie_leaves_kernel(const void *ws, int64_t ld, int64_t n_rows, int32_t V, const int32_t *leaf_node,
                                   int from_logprobs, const float *lse, float scale, float *out, int64_t out_ld) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_rows * V) return;
  const int64_t r = gid / V, k = gid % V;
  out[r * out_ld + leaf_node[k]] = trie_weight<DT>(ws, r * ld + k, from_logprobs, scale, lse ? lse[r] : 0.0f);
}

__global__ void trie_level_kernel(int64_t n_rows, int32_t lo, int32_t hi, const int32_t *level_nodes,
                                  const int32_t *child_ptr, const int32_t *child_idx, int op, float *out,
                                  int64_t out_ld) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t cnt = hi - lo;
  if (gid >= n_rows * cnt) return;
  const int64_t r = gid / cnt;
  const int32_t node = level_nodes[lo + (int32_t)(gid % cnt)];
  float *o = out + r * out_ld;
  double acc = 0.0;
  const int c0 = child_ptr[node], c1 = child_ptr[node + 1];
  if (op == 0)
    for (int c = c0; c < c1; ++c) acc += (double)o[child_idx[c]];
  else
    for (int c = c0; c < c1; ++c) acc = fmax(acc, (double)o[child_idx[c]]);
  o[node] = (float)acc;
}


// ---- node-major trie propagation (batches of >= 32 rows) -----------------------------------------------------------
// In the row-major output a node's children sit subtree sizes apart, so every child read of every row is a cache line
// of its own.  With the values held node-major, scr[node][row], the rows of a node are contiguous: a level is one
// coalesced sweep (each thread: one row of one node, its children's rows read from the same lines as its
// neighbours'), the leaves go in through a 64 x 64 LDS transpose of the weight rows and the result comes out through
// another one.  Same arithmetic per (row, node) as the row-major kernels: ascending children, double, float32 store.
constexpr int kTrieTile = 64;
constexpr int64_t kTrieNodeMajorRows = 32;  // below this the rows are too short a run to coalesce: row-major kernels

template <int DT>
__global__ __launch_bounds__(256) void trie_leaves_t_kernel(const void *ws, int64_t ld, int32_t n_rows, int32_t V,
                                                             const int32_t *leaf_node, int from_logprobs, const float *lse,
                                                             float scale, float *scr, int64_t pitch) {
  __shared__ float tile[kTrieTile][kTrieTile + 1];
  const int k0 = blockIdx.x * kTrieTile, r0 = blockIdx.y * kTrieTile;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int j = ty; j < kTrieTile; j += 4) {  // row r0 + j, token k0 + tx: coalesced over tokens
    const int r = r0 + j, k = k0 + tx;
    float v = 0.f;
    if (r < n_rows && k < V) v = trie_weight<DT>(ws, (int64_t)r * ld + k, from_logprobs, scale, lse ? lse[r] : 0.0f);
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < kTrieTile; j += 4) {  // token k0 + j, row r0 + tx: coalesced over rows
    const int k = k0 + j, r = r0 + tx;
    if (k < V && r < n_rows) scr[(int64_t)leaf_node[k] * pitch + r] = tile[tx][j];
  }
}

// one workgroup per (node of the level, block of rows); a thread owns four consecutive rows (one 16-byte load per child).
// (Several nodes per workgroup, taken one after the other, were measured: 8 per workgroup make a 1024-row batch 0.3 ms
// SLOWER - a node is one short chain of dependent loads, and what hides it is other workgroups, not fewer of them.)
constexpr int kTrieNodesPerGroup = 1;
__global__ __launch_bounds__(256) void trie_level_t_kernel(int32_t rows4, int32_t lo, int32_t hi, const int32_t *level_nodes,
                                                            const int32_t *child_ptr, const int32_t *child_idx, int op,
                                                            float4 *scr4) {
  const int32_t t = blockIdx.y * blockDim.x + threadIdx.x;
  if (t >= rows4) return;
  const int32_t first = lo + blockIdx.x * kTrieNodesPerGroup;
  const int32_t last = first + kTrieNodesPerGroup < hi ? first + kTrieNodesPerGroup : hi;
  for (int32_t j = first; j < last; ++j) {
    const int32_t node = level_nodes[j];  // workgroup-uniform: scalar loads
    const int c0 = child_ptr[node], c1 = child_ptr[node + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // children in batches of eight loads in flight (a node near the root has dozens: one load per trip made the top
    // levels of a 1024-row batch 13-18 us each, all of it latency); added in ascending order all the same
    for (int c = c0; c < c1; c += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c + j < c1) v[j] = scr4[(int64_t)child_idx[c + j] * rows4 + t];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (c + j < c1) {
          if (op == 0) {
            a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w;
          } else {
            a0 = fmax(a0, (double)v[j].x); a1 = fmax(a1, (double)v[j].y); a2 = fmax(a2, (double)v[j].z); a3 = fmax(a3, (double)v[j].w);
          }
        }
      }
    }
    scr4[(int64_t)node * rows4 + t] = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
  }
}

constexpr int kTrieWide = 256, kTrieWideLd = kTrieWide + 4;
// (A 64-token x 256-row tile for the leaves, 16 bytes per lane on both sides and 1 KB runs of a node's rows out, was
// measured at 1024 rows: 140-157 us against the 64 x 64 tile's 128-142 - the scattered node rows, not the run length
// inside them, are what the memory system sees.  Dropped.)

// node-major values -> row-major output for 256 nodes and more: a 256-node x 64-row tile, four rows of a node in, four
// nodes of a row out (1 KB runs of the output row instead of 256 bytes)
__global__ __launch_bounds__(256) void trie_untranspose_wide_kernel(const float *scr, int64_t pitch, int32_t n_rows,
                                                                     int32_t n_out, const int32_t *sel, float *out,
                                                                     int64_t out_ld) {
  __shared__ __attribute__((aligned(16))) float tile[kTrieTile][kTrieWideLd];
  // row blocks fastest: the workgroups that run together read the pieces of the same nodes' rows
  const int n0 = blockIdx.y * kTrieWide, r0 = blockIdx.x * kTrieTile;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 row quads x 16 nodes
  float4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int nd = n0 + ty + 16 * j;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nd < n_out) v[j] = *reinterpret_cast<const float4 *>(scr + (int64_t)(sel ? sel[nd] : nd) * pitch + r0 + 4 * tx);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    tile[4 * tx][ty + 16 * j] = v[j].x;
    tile[4 * tx + 1][ty + 16 * j] = v[j].y;
    tile[4 * tx + 2][ty + 16 * j] = v[j].z;
    tile[4 * tx + 3][ty + 16 * j] = v[j].w;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nd = n0 + 4 * lane;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int rr = 4 * j + w, r = r0 + rr;
    if (r >= n_rows || nd >= n_out) continue;
    const float4 t = *reinterpret_cast<const float4 *>(&tile[rr][4 * lane]);
    float *o = out + (int64_t)r * out_ld + nd;
    if (nd + 4 <= n_out) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f32x4_t __attribute__((aligned(4))) *>(o) = f32x4_t{t.x, t.y, t.z, t.w};
    } else {
      const float e[4] = {t.x, t.y, t.z, t.w};
      for (int q = 0; q < n_out - nd; ++q) o[q] = e[q];
    }
  }
}

// node-major values -> row-major output; with sel: only the nodes sel[0 .. n_out) (out[r, j] = value of node sel[j])
__global__ __launch_bounds__(256) void trie_untranspose_kernel(const float *scr, int64_t pitch, int32_t n_rows,
                                                                int32_t n_out, const int32_t *sel, float *out,
                                                                int64_t out_ld) {
  __shared__ float tile[kTrieTile][kTrieTile + 1];
  const int n0 = blockIdx.x * kTrieTile, r0 = blockIdx.y * kTrieTile;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < kTrieTile; j += 4) {  // node n0 + j, row r0 + tx
    const int nd = n0 + j, r = r0 + tx;
    float v = 0.f;
    if (nd < n_out && r < n_rows) v = scr[(int64_t)(sel ? sel[nd] : nd) * pitch + r];
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < kTrieTile; j += 4) {  // row r0 + j, node n0 + tx
    const int r = r0 + j, nd = n0 + tx;
    if (r < n_rows && nd < n_out) out[(int64_t)r * out_ld + nd] = tile[tx][j];
  }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t glb_trie_workspace(int64_t n_rows, int64_t n_nodes) {
  if (n_rows < kTrieNodeMajorRows || n_nodes <= 0) return 0;
  return align256((size_t)n_nodes * (size_t)((n_rows + 63) & ~(int64_t)63) * sizeof(float));
}

int glb_trie_masses(const glb_trie_args *a, void *stream) {
  if (!a) return glb::api_fail(GLB_EINVAL, "glb_trie_masses: null args");
  if (a->struct_size != sizeof(glb_trie_args))
    return glb::api_fail(GLB_EINVAL, "glb_trie_args.struct_size %u != %zu (ABI mismatch)", a->struct_size, sizeof(glb_trie_args));
  if (!a->weights || !a->leaf_node || !a->level_start_host || !a->level_nodes || !a->child_ptr || !a->child_idx)
    return glb::api_fail(GLB_EINVAL, "null pointer");
  const int64_t n_rows = a->n_rows, vocab = a->vocab, n_nodes = a->n_nodes, n_levels = a->n_levels;
  if (n_rows <= 0 || vocab <= 0 || n_nodes <= vocab || n_levels <= 0 || a->ld < vocab) return glb::api_fail(GLB_EINVAL, "bad sizes");
  if (vocab > 0x7fffff00ll || n_nodes > 0x7fffff00ll || n_rows > 0x7fffff00ll) return glb::api_fail(GLB_EINVAL, "size exceeds 31 bits");
  if (a->dtype < GLB_F32 || a->dtype > GLB_F16) return glb::api_fail(GLB_EINVAL, "bad dtype %d", a->dtype);
  if (a->op != GLB_TRIE_SUM && a->op != GLB_TRIE_MAX) return glb::api_fail(GLB_EINVAL, "bad op %d", a->op);
  if (!a->out && !a->out_sel && !a->keep_node_major) return glb::api_fail(GLB_EINVAL, "no output requested");
  if (a->out && a->out_ld < n_nodes) return glb::api_fail(GLB_EINVAL, "out_ld < n_nodes");
  if (a->out_sel && (!a->sel_nodes || a->n_sel <= 0 || a->out_sel_ld < a->n_sel)) return glb::api_fail(GLB_EINVAL, "bad node selection");
  if (!(a->logit_scale == a->logit_scale)) return glb::api_fail(GLB_EINVAL, "logit_scale is NaN");
  for (int64_t d = 0; d < n_levels; ++d)
    if (a->level_start_host[d + 1] < a->level_start_host[d] || a->level_start_host[d + 1] > n_nodes)
      return glb::api_fail(GLB_EINVAL, "level_start is not a monotone partition");
  hipStream_t s = (hipStream_t)stream;
  const int flp = a->from_logprobs ? 1 : 0;
  const bool node_major = (n_rows >= kTrieNodeMajorRows || a->out_sel || a->keep_node_major) && a->workspace &&
                          a->workspace_bytes >= glb_trie_workspace_ex(n_rows, n_nodes) && ((uintptr_t)a->workspace) % 16 == 0;
  if (node_major) {  // node-major values: every level is a coalesced sweep
    float *scr = (float *)a->workspace;
    const int64_t pitch = (n_rows + 63) & ~(int64_t)63;
    const dim3 tg(blocks_for(vocab, kTrieTile), blocks_for(n_rows, kTrieTile));
    switch (a->dtype) {
      case GLB_F32: hipLaunchKernelGGL(trie_leaves_t_kernel<GLB_F32>, tg, dim3(256), 0, s, a->weights, a->ld, (int32_t)n_rows, (int32_t)vocab, a->leaf_node, flp, a->lse, a->logit_scale, scr, pitch); break;
      case GLB_BF16: hipLaunchKernelGGL(trie_leaves_t_kernel<GLB_BF16>, tg, dim3(256), 0, s, a->weights, a->ld, (int32_t)n_rows, (int32_t)vocab, a->leaf_node, flp, a->lse, a->logit_scale, scr, pitch); break;
      default: hipLaunchKernelGGL(trie_leaves_t_kernel<GLB_F16>, tg, dim3(256), 0, s, a->weights, a->ld, (int32_t)n_rows, (int32_t)vocab, a->leaf_node, flp, a->lse, a->logit_scale, scr, pitch); break;
    }
    for (int64_t d = 0; d < n_levels; ++d) {
      const int32_t lo = a->level_start_host[d], hi = a->level_start_host[d + 1];
      if (hi == lo) continue;
      const int32_t rows4 = (int32_t)(pitch / 4);  // the padding rows of scr are computed too (never read back)
      const int bt = rows4 < 256 ? (rows4 + 63) & ~63 : 256;
      hipLaunchKernelGGL(trie_level_t_kernel, dim3(blocks_for(hi - lo, kTrieNodesPerGroup), blocks_for(rows4, bt)), dim3(bt), 0,
                         s, rows4, lo, hi, a->level_nodes, a->child_ptr, a->child_idx, (int)a->op, (float4 *)scr);
    }
    auto untranspose = [&](int64_t n_out, const int32_t *sel, float *out, int64_t out_ld) {
      if (n_out >= kTrieWide)
        hipLaunchKernelGGL(trie_untranspose_wide_kernel, dim3(blocks_for(n_rows, kTrieTile), blocks_for(n_out, kTrieWide)),
                           dim3(256), 0, s, scr, pitch, (int32_t)n_rows, (int32_t)n_out, sel, out, out_ld);
      else
        hipLaunchKernelGGL(trie_untranspose_kernel, dim3(blocks_for(n_out, kTrieTile), blocks_for(n_rows, kTrieTile)),
                           dim3(256), 0, s, scr, pitch, (int32_t)n_rows, (int32_t)n_out, sel, out, out_ld);
    };
    if (a->out) untranspose(n_nodes, nullptr, a->out, a->out_ld);
    if (a->out_sel) untranspose(a->n_sel, a->sel_nodes, a->out_sel, a->out_sel_ld);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return glb::api_hip_fail(e, "trie_masses launch");
    return GLB_OK;
  }
  if (!a->out) return glb::api_fail(GLB_ENOSPC, "node-major / selected output needs glb_trie_workspace_ex(n_rows, n_nodes) bytes of workspace");
  const unsigned lb = blocks_for(n_rows * vocab, 256);
  switch (a->dtype) {
    case GLB_F32: hipLaunchKernelGGL(trie_leaves_kernel<GLB_F32>, dim3(lb), dim3(256), 0, s, a->weights, a->ld, n_rows, (int32_t)vocab, a->leaf_node, flp, a->lse, a->logit_scale, a->out, a->out_ld); break;
    case GLB_BF16: hipLaunchKernelGGL(trie_leaves_kernel<GLB_BF16>, dim3(lb), dim3(256), 0, s, a->weights, a->ld, n_rows, (int32_t)vocab, a->leaf_node, flp, a->lse, a->logit_scale, a->out, a->out_ld); break;
    default: hipLaunchKernelGGL(trie_leaves_kernel<GLB_F16>, dim3(lb), dim3(256), 0, s, a->weights, a->ld, n_rows, (int32_t)vocab, a->leaf_node, flp, a->lse, a->logit_scale, a->out, a->out_ld); break;
  }
  for (int64_t d = 0; d < n_levels; ++d) {  // one launch per tree level, all rows
    const int32_t lo = a->level_start_host[d], hi = a->level_start_host[d + 1];
    if (hi == lo) continue;
    hipLaunchKernelGGL(trie_level_kernel, dim3(blocks_for(n_rows * (hi - lo), 256)), dim3(256), 0, s, n_rows, lo, hi,
                       a->level_nodes, a->child_ptr, a->child_idx, (int)a->op, a->out, a->out_ld);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return glb::api_hip_fail(e, "trie_masses launch");
  return GLB_OK;
}

size_t glb_trie_workspace_ex(int64_t n_rows, int64_t n_nodes) {
  if (n_rows <= 0 || n_nodes <= 0) return 0;
  return align256((size_t)n_nodes * (size_t)((n_rows + 63) & ~(int64_t)63) * sizeof(float));
}

int glb_trie_reduce(const float *weights, int64_t ld, int64_t n_rows, int64_t vocab, int64_t n_nodes, int64_t n_levels,
                    const int32_t *leaf_node, const int32_t *level_start_host, const int32_t *level_nodes,
                    const int32_t *child_ptr, const int32_t *child_idx, int32_t op, int32_t from_logprobs, float *out,
                    int64_t out_ld, void *workspace, size_t workspace_bytes, void *stream) {
  if (!out) return glb::api_fail(GLB_EINVAL, "null pointer");
  glb_trie_args a{};
  a.struct_size = sizeof(glb_trie_args);
  a.weights = weights;
  a.dtype = GLB_F32;
  a.ld = ld;
  a.n_rows = n_rows;
  a.vocab = vocab;
  a.logit_scale = 1.0f;
  a.from_logprobs = from_logprobs;
  a.op = op;
  a.n_nodes = n_nodes;
  a.n_levels = n_levels;
  a.leaf_node = leaf_node;
  a.level_start_host = level_start_host;
  a.level_nodes = level_nodes;
  a.child_ptr = child_ptr;
  a.child_idx = child_idx;
  a.out = out;
  a.out_ld = out_ld;
  // (batches below 32 rows keep the row-major kernels: the rows are too short a run to coalesce node-major)
  if (n_rows >= kTrieNodeMajorRows) {
    a.workspace = workspace;
    a.workspace_bytes = workspace_bytes;
  }
  return glb_trie_masses(&a, stream);
}

}  // extern "C"
