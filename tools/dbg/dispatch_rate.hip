// How fast does the chip start one-wave workgroups, and does it depend on the registers a wave asks for?
//   empty kernels (exit at once) of 64 threads with 8 / 32 / 64 / 96 / 128 VGPRs, 16384 .. 131072 workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NV>
__global__ __launch_bounds__(64) void k(float *out, int never) {
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
  if (never) {  // keeps NV registers allocated without executing anything
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
    float s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    out[threadIdx.x] = s;
  }
}
template <int NV>
void run(float *d) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int n : {16384, 65536, 131072}) {
    hipLaunchKernelGGL(k<NV>, dim3(n), dim3(64), 0, 0, d, 0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<NV>, dim3(n), dim3(64), 0, 0, d, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("VGPR array %3d: %6d one-wave workgroups in %7.2f us = %6.0f waves/us\n", NV, n, ms * 100, n / (ms * 100));
  }
}
int main() {
  float *d;
  hipMalloc(&d, 4096);
  run<4>(d); run<32>(d); run<64>(d); run<96>(d); run<120>(d);
  return 0;
}
