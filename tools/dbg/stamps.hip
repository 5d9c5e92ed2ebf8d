// diagnostic: run row_kernel_persist (fp32 gpt2 rows) with shader-clock stamps and print per-phase cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DV4 -DMASKK=1 -DMODEE=1 tools/dbg/stamps.hip -o tools/dbg/stamps_11
#define GLB_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../../genlm-backend_amd/csrc/glb_row_persist.hpp"
int main() {
  const int B = 1024, V = 50257;
  float *x;
  hipMalloc(&x, (size_t)B * V * 4 * 4);
  std::vector<float> h((size_t)B * V);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.01f - 5.f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int c = 1; c < 4; ++c) hipMemcpy(x + c * h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float *lz, *ls;
  hipMalloc(&lz, B * 4);
  hipMalloc(&ls, B * 4);
  glb::RowParams p{};
  p.logits = x; p.ld = V; p.V = V; p.n_particles = B; p.out_logZ = lz; p.out_lse = ls;
  uint64_t *ws;
  hipMalloc(&ws, (size_t)B * 70 * 8);
  p.row_sums = ws; p.row_exps = (float *)(ws + (size_t)B * 2);
  uint32_t *mk;
  int W32 = (V + 31) / 32;
  hipMalloc(&mk, (size_t)2 * W32 * 4);
  hipMemset(mk, 0xB7, (size_t)2 * W32 * 4);
  p.mask = mk; p.mask_ld = W32; p.n_masks = 1;
  int32_t *tok;
  hipMalloc(&tok, B * 4);
  p.out_token = tok; p.seed = 5;
  for (int skew : {0}) {
  for (int it = 0; it < 8; ++it) {
    p.logits = x + (it & 3) * h.size();
    hipLaunchKernelGGL((glb::row_kernel_persist<0, MASKK, MODEE, 25, 512, false>), dim3(256), dim3(512), 0, 0, p);
  }
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(256 * 2 * 64), rt(512);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(glb::g_stamps), st.size() * 8);
  hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(glb::g_realtime), rt.size() * 8);
  const char *names[] = {"issue0", "arrive0", "land0"};
  for (int wg : {0}) { if (skew != 0 && skew != 10000) break;
    for (int wv = 0; wv < 2; ++wv) {
      const unsigned long long *s = &st[(wg * 2 + wv) * 64];
      printf("WG %3d wave %s:", wg, wv ? "last " : "first");
      int i = 1;
#ifdef V4
      for (; i < 2 && s[i]; ++i) printf(" first row issued+mask %llu", s[i] - s[i - 1]);
#else
      for (; i < 4 && s[i]; ++i) printf(" %s %llu", names[i - 1], s[i] - s[i - 1]);
#endif
      int row = 0;
      while (i < 64 && s[i]) {
        printf(" | row%d ph1 %llu", row, s[i] - s[i - 1]); ++i;
        if (i < 64 && s[i]) { printf(" ph2 %llu", s[i] - s[i - 1]); ++i; }
        if (i < 64 && s[i]) { printf(" red %llu", s[i] - s[i - 1]); ++i; }
#ifdef V4
        if (i < 64 && s[i]) { printf(" bits/tail %llu", s[i] - s[i - 1]); ++i; }
#else
        if (i < 64 && s[i]) { printf(" wait %llu", s[i] - s[i - 1]); ++i; }
        if (i < 64 && s[i]) { printf(" land %llu", s[i] - s[i - 1]); ++i; }
#endif
        ++row;
      }
      printf("  total %llu cyc", s[i - 1] - s[0]);
      if (!wv) printf("  realtime %llu ticks (100 MHz) -> %.2f GHz", rt[wg * 2 + 1] - rt[wg * 2],
                      (double)(s[i - 1] - s[0]) / (double)(rt[wg * 2 + 1] - rt[wg * 2]) * 0.1);
      printf("\n");
    }
  }
#ifdef V4
  {  // averages over all workgroups: stamps 0 start, 1 prologue, then per row ph1 ph2 red bits (last row: tail)
    for (int wv = 0; wv < 2; ++wv) {
      double sum[32] = {0};
      for (int wg = 0; wg < 256; ++wg) {
        const unsigned long long *q = &st[(wg * 2 + wv) * 64];
        for (int i = 1; i < 20; ++i) sum[i] += (double)(q[i] - q[i - 1]);
      }
      printf("AVG wave %s: prologue %.0f |", wv ? "last " : "first", sum[1] / 256);
      for (int r = 0; r < 4; ++r)
        printf(" row%d ph1 %.0f ph2 %.0f red %.0f %s %.0f |", r, sum[2 + 4 * r] / 256, sum[3 + 4 * r] / 256,
               sum[4 + 4 * r] / 256, r == 3 ? "tail" : "bits", sum[5 + 4 * r] / 256);
      printf("\n   tail: last draw + drain %.0f, own-scale redo %.0f, lse/logZ %.0f\n", sum[17] / 256, sum[18] / 256,
             sum[19] / 256);
    }
  }
#endif
  unsigned long long mn = ~0ull, mx = 0;
  for (int wg = 0; wg < 256; ++wg) {
    mn = std::min(mn, rt[wg * 2]);
    mx = std::max(mx, rt[wg * 2 + 1]);
  }
  printf("skew %d: span %.2f us\n", skew, (double)(mx - mn) * 0.01);
  }
  return 0;
}
