// diagnostic: run row_kernel_v2 (fp32, no mask, stats) with s_memtime stamps and print per-phase cycles
#define GLB_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../genlm-backend_amd/csrc/glb_row_kernel_v2.hpp"
int main() {
  const int B = 1024, V = 50257;
  float *x; hipMalloc(&x, (size_t)B * V * 4 * 2);
  std::vector<float> h((size_t)B * V);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.01f - 5.f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(x + h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float *lz, *ls; hipMalloc(&lz, B * 4); hipMalloc(&ls, B * 4);
  glb::RowParams p{}; p.logits = x; p.ld = V; p.V = V; p.n_particles = B; p.out_logZ = lz; p.out_lse = ls;
  uint64_t* ws; hipMalloc(&ws, (size_t)B * 70 * 8); p.chunk_sums = ws; p.row_sums = ws + (size_t)B * 64; p.row_exps = (float*)(ws + (size_t)B * 66); p.n_chunks = 40; p.chunk_vecs = 320;
  uint32_t* mk; int W32 = (V + 31) / 32; hipMalloc(&mk, (size_t)2 * W32 * 4); hipMemset(mk, 0xB7, (size_t)2 * W32 * 4); p.mask = mk; p.mask_ld = W32; p.n_masks = 1;
  for (int it = 0; it < 6; ++it) {
    p.logits = x + (it & 1) * h.size();
    hipLaunchKernelGGL((glb::row_kernel_v2<0, MASKK, MODEE, 25, 19, 6, 512, false>), dim3(256), dim3(512), 0, 0, p);
  }
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(256 * 64);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(glb::g_stamps), st.size() * 8);
  // print for a few WGs the deltas
  for (int wg : {0, 1, 100, 255}) {
    printf("WG %d:", wg);
    unsigned long long t0 = st[wg * 64];
    for (int i = 1; i < 64 && st[wg * 64 + i]; ++i) printf(" %llu", (st[wg * 64 + i] - st[wg * 64 + i - 1]));
    printf("  total %llu\n", st[wg*64+ (int)(std::find(st.begin()+wg*64+1, st.begin()+wg*64+64, 0ull)-(st.begin()+wg*64)) -1] - t0);
  }
  // global: min start, max end
  unsigned long long mn = ~0ull, mx = 0;
  for (int wg = 0; wg < 256; ++wg) { mn = std::min(mn, st[wg*64]); for (int i = 0; i < 64; ++i) mx = std::max(mx, st[wg*64+i]); }
  printf("span %llu cycles\n", mx - mn);
  return 0;
}
