// How fast do one-wave workgroups START when they stay resident (the ramp of the fused step)?  Every wave stamps its start
// (s_memrealtime, 100 MHz) and then lives ~3 us: A = sleeps only, B = first issues eight 16-byte non-temporal loads of its
// own 8 KiB (the stats wave's burst) and waits for them, C = like A with a body of ~40 KB of code behind a never-taken
// branch... (no: the I-cache only matters for code that runs).  Prints how many waves had started by t after the first.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Big { uint64_t a[40]; };
template <int MODE, int NV>
__global__ __launch_bounds__(64) void kbig(uint64_t *stamps, Big big, float *out, int never, int sleep_iters) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < sleep_iters; ++i) __builtin_amdgcn_s_sleep(16);
  if (never) out[threadIdx.x] = (float)big.a[never & 31];
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
template <int MODE, int NV>
__global__ __launch_bounds__(64) void kscr(uint64_t *stamps, const u32x4 *src, float *out, int never, int sleep_iters) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < sleep_iters; ++i) __builtin_amdgcn_s_sleep(16);
  if (never) {  // a private array indexed at run time: scratch memory, allocated for every wave whether it gets here or not
    float a[64];
    for (int i = 0; i < 64; ++i) a[i] = out[i];
    out[threadIdx.x] = a[(never * threadIdx.x) & 63];
  }
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
__global__ __launch_bounds__(64) void k16(uint64_t *stamps, const u32x4 *src, float *out, int never, int sleep_iters) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("; v119 in use" ::: "v119");
  const u32x4 *p = src + (size_t)blockIdx.x * 1024 + threadIdx.x;
  u32x4 r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = __builtin_nontemporal_load(p + i * 64);
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
  if (never) out[threadIdx.x] = acc;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + (acc == 0x12345u ? 1 : 0);
  }
}
// persistent form: `gridDim.x` resident waves, wave w takes chunks w, w + gridDim.x, ...; the NEXT chunk's eight loads are
// issued before the current chunk's BODY instructions run (the loaded words feed the chains, so nothing is reordered away)
template <int BODY, int PREFETCH>
__global__ __launch_bounds__(64) void kpers(uint64_t *stamps, const u32x4 *src, float *out, int never, int n_items) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("; v95 in use" ::: "v95");
  float a = 1.0f, b = 2.0f, c = 3.0f, d = 4.0f;
  uint32_t acc = 0;
  int item = blockIdx.x;
  u32x4 r[8], q[8];
  if (item < n_items) {
    const u32x4 *p = src + (size_t)item * 512 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = __builtin_nontemporal_load(p + i * 64);
  }
  while (item < n_items) {
    const int nxt = item + gridDim.x;
    if (PREFETCH && nxt < n_items) {
      const u32x4 *p = src + (size_t)nxt * 512 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_nontemporal_load(p + i * 64);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    a += (float)(acc & 1u);
#pragma unroll
    for (int i = 0; i < BODY / 4; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(b) : "v"(c));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c) : "v"(d));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(a));
    }
    if (!PREFETCH && nxt < n_items) {
      const u32x4 *p = src + (size_t)nxt * 512 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __builtin_nontemporal_load(p + i * 64);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = q[i];
    item = nxt;
  }
  if (never) out[threadIdx.x] = a + b + c + d + acc;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + (acc == 0x12345u ? 1 : 0);
  }
}
template <int BODY>
__global__ __launch_bounds__(64) void kbody(uint64_t *stamps, const u32x4 *src, float *out, int never, int sleep_iters) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("; v95 in use" ::: "v95");
  const u32x4 *p = src + (size_t)blockIdx.x * 512 + threadIdx.x;
  u32x4 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = __builtin_nontemporal_load(p + i * 64);
  float a = 1.0f, b = 2.0f, c = 3.0f, d = 4.0f;
#pragma unroll
  for (int i = 0; i < BODY / 4; ++i) {  // straight-line code: BODY 8-byte instructions, four independent chains
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(b) : "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c) : "v"(d));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(a));
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
  if (never) out[threadIdx.x] = a + b + c + d + acc;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + (acc == 0x12345u ? 1 : 0);
  }
}
template <int MODE, int NV>
__global__ __launch_bounds__(64) void k(uint64_t *stamps, const u32x4 *src, float *out, int never, int sleep_iters) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
  if (NV >= 120) asm volatile("; v119 in use" ::: "v119");  // (the allocation the kernel asks for: up to the highest register named)
  else if (NV >= 96) asm volatile("; v95 in use" ::: "v95");
  uint32_t acc = 0;
  if (MODE == 1) {
    const u32x4 *p = src + (size_t)blockIdx.x * 512 + threadIdx.x;
    u32x4 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = __builtin_nontemporal_load(p + i * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
  }
  for (int i = 0; i < sleep_iters; ++i) __builtin_amdgcn_s_sleep(16);
  if (never) {
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
    float s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    out[threadIdx.x] = s + acc;
  }
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + (acc == 0x12345u ? 1 : 0);
  }
}
template <int MODE, int NV>
void run(const char *name, int n, int sleep_iters, uint64_t *d_st, const u32x4 *src, float *out) {
  const int n_items = 16384;
  std::vector<uint64_t> h(2 * n);
  static char *flush = nullptr;
  if (!flush) hipMalloc(&flush, 640u << 20);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d_st, 0, 2 * n * 8);
    hipMemset(flush, rep, 640u << 20);  // (the 256 MiB Infinity Cache holds something else when the kernel starts)
    hipDeviceSynchronize();
    if (MODE == 2) hipLaunchKernelGGL((kscr<MODE, NV>), dim3(n), dim3(64), 0, 0, d_st, src, out, 0, sleep_iters);
    else if (MODE == 3) hipLaunchKernelGGL((kbig<MODE, NV>), dim3(n), dim3(64), 0, 0, d_st, Big{}, out, 0, sleep_iters);
    else if (MODE == 7) hipLaunchKernelGGL((kpers<NV, 1>), dim3(n), dim3(64), 0, 0, d_st, src, out, 0, n_items);
    else if (MODE == 8) hipLaunchKernelGGL((kpers<NV, 0>), dim3(n), dim3(64), 0, 0, d_st, src, out, 0, n_items);
    else if (MODE == 6) hipLaunchKernelGGL(k16, dim3(n), dim3(64), 0, 0, d_st, src, out, 0, sleep_iters);
    else if (MODE == 5) hipLaunchKernelGGL((kbody<NV>), dim3(n), dim3(64), 0, 0, d_st, src, out, 0, sleep_iters);
    else if (MODE == 4) hipLaunchKernelGGL((k<0, NV>), dim3(n), dim3(64), 4096, 0, d_st, src, out, 0, sleep_iters);
    else hipLaunchKernelGGL((k<MODE, NV>), dim3(n), dim3(64), 0, 0, d_st, src, out, 0, sleep_iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d_st, 2 * n * 8, hipMemcpyDeviceToHost);
  std::vector<double> st(n), life(n);
  uint64_t t0 = ~0ull;
  for (int i = 0; i < n; ++i) t0 = std::min(t0, h[2 * i]);
  for (int i = 0; i < n; ++i) st[i] = (h[2 * i] - t0) * 0.01, life[i] = (h[2 * i + 1] - h[2 * i]) * 0.01;
  std::sort(st.begin(), st.end());
  std::sort(life.begin(), life.end());
  uint64_t t1 = 0;
  for (int i = 0; i < n; ++i) t1 = std::max(t1, h[2 * i + 1]);
  printf("%-36s n=%6d  first start -> last end %6.2f us  life p50 %5.2f us | started by", name, n, (t1 - t0) * 0.01, life[n / 2]);
  for (double t : {0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0}) printf("  %.2g us: %5d", t, (int)(std::upper_bound(st.begin(), st.end(), t) - st.begin()));
  printf("\n");
}
int main() {
  const int n = 16384;
  uint64_t *st;
  u32x4 *src;
  float *out;
  hipMalloc(&st, 2 * n * 8);
  hipMalloc(&src, (size_t)n * 16384);
  hipMemset(src, 1, (size_t)n * 16384);
  hipMalloc(&out, 4096);
  run<0, 8>("sleep, 8 VGPRs", n, 60, st, src, out);
  run<0, 96>("sleep, 96 VGPRs (5 waves/SIMD)", n, 60, st, src, out);
  run<0, 120>("sleep, 120 VGPRs (4 waves/SIMD)", n, 60, st, src, out);
  run<1, 96>("8 KiB of loads + sleep, 96 VGPRs", n, 40, st, src, out);
  run<1, 96>("8 KiB of loads, 96 VGPRs", n, 0, st, src, out);
  run<0, 96>("exit at once, 96 VGPRs", n, 0, st, src, out);
  run<6, 120>("16 KiB loads, 120 VGPRs, 12544 waves", 12544, 0, st, src, out);
  run<6, 120>("16 KiB loads, 120 VGPRs, 16384 waves", n, 0, st, src, out);
  run<5, 64>("8 KiB loads + 64 instr (0.5 KB)", n, 0, st, src, out);
  run<5, 512>("8 KiB loads + 512 instr (4 KB)", n, 0, st, src, out);
  run<5, 1024>("8 KiB loads + 1024 instr (8 KB)", n, 0, st, src, out);
  run<5, 2048>("8 KiB loads + 2048 instr (16 KB)", n, 0, st, src, out);
  run<5, 4096>("8 KiB loads + 4096 instr (32 KB)", n, 0, st, src, out);
  for (int waves : {4096, 5120}) {
    printf("persistent, %d waves over 16384 chunks of 8 KiB:\n", waves);
    run<7, 64>("  next chunk's loads ahead, 64 instr", waves, 0, st, src, out);
    run<7, 512>("  next chunk's loads ahead, 512 instr", waves, 0, st, src, out);
    run<7, 1024>("  next chunk's loads ahead, 1024 instr", waves, 0, st, src, out);
    run<8, 1024>("  loads after the body, 1024 instr", waves, 0, st, src, out);
  }
  run<2, 8>("sleep, scratch allocated", n, 60, st, src, out);
  run<3, 8>("sleep, 320-byte kernarg", n, 60, st, src, out);
  run<4, 8>("sleep, 4 KB of dynamic LDS", n, 60, st, src, out);
  return 0;
}
