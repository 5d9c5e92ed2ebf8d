#define GLB_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../genlm-backend_amd/csrc/glb_row_kernel_v3.hpp"
int main() {
  const int B = 1024, V = 50257;
  float *x; hipMalloc(&x, (size_t)B * V * 4 * 2);
  std::vector<float> h((size_t)B * V);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.01f - 5.f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(x + h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
  glb::V3Params q{};
  q.rp.logits = x; q.rp.ld = V; q.rp.V = V; q.rp.n_particles = B;
  uint32_t* mk; int W32 = (V + 31) / 32; hipMalloc(&mk, (size_t)2 * W32 * 4); hipMemset(mk, 0xB7, (size_t)2 * W32 * 4); q.rp.mask = mk; q.rp.mask_ld = W32; q.rp.n_masks = 1;
  char* ws; size_t wsz = (size_t)B * (72 * 8 + 8) + 64; hipMalloc(&ws, wsz); hipMemset(ws, 0, wsz);
  q.xch = (unsigned long long*)ws; q.wave_sums = (uint64_t*)(ws + (size_t)B * 64); q.row_exps = (float*)(ws + (size_t)B * 72 * 8); q.timeout = (unsigned*)(ws + (size_t)B * (72 * 8 + 8)); q.n_clusters = 64;
  for (int it = 0; it < 4; ++it) {
    q.rp.logits = x + (it & 1) * h.size();
    hipMemset(ws, 0, (size_t)B * 64);
    hipLaunchKernelGGL((glb::row_kernel_v3<0, MASKK, 7, false>), dim3(256), dim3(512), 0, 0, q);
  }
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(256 * 96);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(glb::g_stamps3), st.size() * 8);
  for (int wg : {0, 101}) {
    printf("WG %d:", wg);
    for (int i = 1; i < 96 && st[wg * 96 + i]; ++i) printf(" %llu%s", (st[wg * 96 + i] - st[wg * 96 + i - 1]), (i % 6 == 0) ? " |" : "");
    printf("\n");
  }
  unsigned to; hipMemcpy(&to, q.timeout, 4, hipMemcpyDeviceToHost); printf("timeout flag %u\n", to);
  return 0;
}
