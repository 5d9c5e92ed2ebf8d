#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../genlm-backend_amd/csrc/glb_math.hpp"
__global__ void k(const uint64_t* in, uint64_t* out, uint64_t* tot, int cond_wave) {
  int tid = threadIdx.x, wave = tid >> 6;
  uint64_t v = in[tid];
  if (wave == cond_wave) {
    int lo = 0, hi = 5; uint64_t Tw = 3;
    while (hi - lo > 1) { int mid = (lo+hi)>>1; uint64_t c = glb::wave_sum_u64(v); if (Tw < c) hi = mid; else {Tw -= c; lo = mid;} }
    out[tid] = glb::wave_scan_u64(v);
    tot[tid] = glb::wave_sum_u64(v);
  }
}
int main() {
  uint64_t h[256], *d, *o, *t; 
  for (int i=0;i<256;i++) h[i] = (uint64_t)(i+1) * 0x100000001ull;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h); hipMalloc(&t, sizeof h);
  hipMemcpy(d,h,sizeof h,hipMemcpyHostToDevice); hipMemset(o,0,sizeof h);
  for (int w=0; w<4; ++w) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, t, w);
  uint64_t r[256], tt[256]; hipMemcpy(r,o,sizeof h,hipMemcpyDeviceToHost); hipMemcpy(tt,t,sizeof h,hipMemcpyDeviceToHost);
  int bad=0;
  for (int w=0;w<4;++w){ uint64_t run=0; for(int l=0;l<64;++l){ run+=h[w*64+l]; if(r[w*64+l]!=run){ if(bad<10) printf("scan mismatch w%d l%d got %llx want %llx\n",w,l,(unsigned long long)r[w*64+l],(unsigned long long)run); bad++; } }
    for(int l=0;l<64;++l) if(tt[w*64+l]!=run){ if(bad<20) printf("tot mismatch w%d l%d got %llx want %llx\n",w,l,(unsigned long long)tt[w*64+l],(unsigned long long)run); bad++; } }
  printf("bad=%d\n",bad); return bad!=0;
}
