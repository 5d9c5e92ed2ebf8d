// glb_trie.hpp — the leaf weight of the token -> byte trie kernels (glb_api.hip: level-synchronous node-major kernels;
// glb_trie.hip: the row-resident kernel): one definition, so both paths produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/glb.h"

namespace glb {
// e^x by the hardware's 2^x (v_exp_f32, 1 ulp) on x * log2(e): relative error about |x| * 1e-7 - the input, a logit minus
// a float32 lse, is no better known - for 3 instructions instead of expf's 25 (a third of the leaves kernel's time).
__device__ __forceinline__ float trie_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// one weight from its value: from_logprobs: exp(x * scale - lse) - with the row's lse from the fused step this turns
// LOGITS into probabilities on the way in (no [B, V] log-prob matrix is ever written)
__device__ __forceinline__ float trie_weight_value(float v, int from_logprobs, float scale, float lse) {
  return from_logprobs ? trie_exp(v * scale - lse) : v;
}

template <int DT>  // GLB_BF16 / GLB_F16
__device__ __forceinline__ float trie_upcast(uint16_t h) {
  if constexpr (DT == GLB_BF16) return __uint_as_float((uint32_t)h << 16);
  else return (float)__builtin_bit_cast(_Float16, h);
}

// element idx of the weights of any element type, as float
template <int DT>  // GLB_F32 / GLB_BF16 / GLB_F16
__device__ __forceinline__ float trie_weight_load(const void *ws, int64_t idx) {
  if constexpr (DT == GLB_F32) return reinterpret_cast<const float *>(ws)[idx];
  else return trie_upcast<DT>(reinterpret_cast<const uint16_t *>(ws)[idx]);
}

template <int DT>
__device__ __forceinline__ float trie_weight(const void *ws, int64_t idx, int from_logprobs, float scale, float lse) {
  return trie_weight_value(trie_weight_load<DT>(ws, idx), from_logprobs, scale, lse);
}
}  // namespace glb
