// prototype of the chunked streaming statistics kernel ("K1") + host check of its integer sums.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dbg/chunk_probe.hip -o tools/dbg/chunk_probe
// Contract under test: a row is cut in chunks of 4096 elements; per chunk c: N_c = n(max x), every element gives
//   t = ldexp(P(f), n - N_c) in fp32 (P = 2^(f-1) by degree-5 Horner, clamped to [0,1]) and q = floor(t * 2^36);
//   S_c = sum q (all), S_c^m = sum q over allowed.  The kernel gets the same integers from fp32 adds in
//   round-toward-zero mode on two grids (2^-18, 2^-36).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int CH = 4096;
constexpr float kLog2e = 1.44269502f;
constexpr float kMagic = 12582912.0f;
constexpr uint32_t kMagicBits = 0x4B400000u;
// 2^(f-1) on |f|<=1/2: the 2^30-scaled coefficients of glb_math.hpp times 2^-31
__host__ __device__ inline float c_of(uint32_t bits) { union { uint32_t u; float f; } v; v.u = bits - (31u << 23); return v.f; }
#define C0 0x4e800000u
#define C1 0x4e317216u
#define C2 0x4d75fcd9u
#define C3 0x4c635b16u
#define C4 0x4b1e7722u
#define C5 0x49adfe07u

struct Rec { float Nc; uint32_t pA, pB, pAm, pBm; uint32_t pad[3]; };

__device__ __forceinline__ float fma_clamp(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ float wave_max(float v) {
#define STEP(CTRL, RM) { float o = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)0xff800000u, (int)__float_as_uint(v), CTRL, RM, 0xf, false)); v = fmaxf(v, o); }
  STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
#undef STEP
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#define STEP(CTRL, RM) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xf, false);
  STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
#undef STEP
  return v;  // lane 63 holds the total
}

// accumulate 4 terms: elements 0,2 -> set 0, elements 1,3 -> set 1.  fp32 round-toward-zero inside.
template <bool MASKED>
__device__ __forceinline__ void acc4(float t0, float t1, float t2, float t3, float &A0, float &B0, float &A1, float &B1,
                                     float &Am0, float &Bm0, float &Am1, float &Bm1, uint64_t M0, uint64_t M1,
                                     uint64_t M2, uint64_t M3) {
  float x0, x1, d0, d1, l0, l1;
  if constexpr (MASKED) {
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
        "v_add_f32 %8, %0, %14\n\t"
        "v_add_f32 %9, %2, %15\n\t"
        "v_sub_f32 %10, %8, %0\n\t"
        "v_sub_f32 %11, %9, %2\n\t"
        "v_sub_f32 %12, %14, %10\n\t"
        "v_sub_f32 %13, %15, %11\n\t"
        "v_add_f32 %1, %1, %12\n\t"
        "v_add_f32 %3, %3, %13\n\t"
        "s_mov_b64 exec, %18\n\t"
        "v_add_f32 %4, %4, %10\n\t"
        "v_add_f32 %5, %5, %12\n\t"
        "s_mov_b64 exec, %19\n\t"
        "v_add_f32 %6, %6, %11\n\t"
        "v_add_f32 %7, %7, %13\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_add_f32 %0, %8, %16\n\t"
        "v_add_f32 %2, %9, %17\n\t"
        "v_sub_f32 %10, %0, %8\n\t"
        "v_sub_f32 %11, %2, %9\n\t"
        "v_sub_f32 %12, %16, %10\n\t"
        "v_sub_f32 %13, %17, %11\n\t"
        "v_add_f32 %1, %1, %12\n\t"
        "v_add_f32 %3, %3, %13\n\t"
        "s_mov_b64 exec, %20\n\t"
        "v_add_f32 %4, %4, %10\n\t"
        "v_add_f32 %5, %5, %12\n\t"
        "s_mov_b64 exec, %21\n\t"
        "v_add_f32 %6, %6, %11\n\t"
        "v_add_f32 %7, %7, %13\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
        : "+v"(A0), "+v"(B0), "+v"(A1), "+v"(B1), "+v"(Am0), "+v"(Bm0), "+v"(Am1), "+v"(Bm1), "=&v"(x0), "=&v"(x1),
          "=&v"(d0), "=&v"(d1), "=&v"(l0), "=&v"(l1)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(M0), "s"(M1), "s"(M2), "s"(M3));
  } else {
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
        "v_add_f32 %4, %0, %10\n\t"
        "v_add_f32 %5, %2, %11\n\t"
        "v_sub_f32 %6, %4, %0\n\t"
        "v_sub_f32 %7, %5, %2\n\t"
        "v_sub_f32 %8, %10, %6\n\t"
        "v_sub_f32 %9, %11, %7\n\t"
        "v_add_f32 %1, %1, %8\n\t"
        "v_add_f32 %3, %3, %9\n\t"
        "v_add_f32 %0, %4, %12\n\t"
        "v_add_f32 %2, %5, %13\n\t"
        "v_sub_f32 %6, %0, %4\n\t"
        "v_sub_f32 %7, %2, %5\n\t"
        "v_sub_f32 %8, %12, %6\n\t"
        "v_sub_f32 %9, %13, %7\n\t"
        "v_add_f32 %1, %1, %8\n\t"
        "v_add_f32 %3, %3, %9\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
        : "+v"(A0), "+v"(B0), "+v"(A1), "+v"(B1), "=&v"(x0), "=&v"(x1), "=&v"(d0), "=&v"(d1), "=&v"(l0), "=&v"(l1)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
  }
}

__device__ __forceinline__ float term(float x, float magicN) {
  const float tm = __builtin_fmaf(x, kLog2e, magicN);
  const float negn = magicN - tm;
  const int np = (int)(__float_as_uint(tm) - kMagicBits);
  const float f = __builtin_fmaf(x, kLog2e, negn);
  float p = c_of(C5);
  p = __builtin_fmaf(p, f, c_of(C4));
  p = __builtin_fmaf(p, f, c_of(C3));
  p = __builtin_fmaf(p, f, c_of(C2));
  p = __builtin_fmaf(p, f, c_of(C1));
  p = fma_clamp(p, f, c_of(C0));
  return __builtin_amdgcn_ldexpf(p, np);
}

template <int DT> struct Tr { static constexpr int EPV = DT == 0 ? 4 : 8, ES = DT == 0 ? 4 : 2, NVC = 64 / EPV; };

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// MODE: 0 = loads + max only, 1 = full statistics
template <int DT, bool MASKED, int MODE>
__global__ __launch_bounds__(256) void k1(const char *__restrict__ logits, int64_t ld, int V, int n_pairs, int nch,
                                          const int *__restrict__ pair_row, const int *__restrict__ pair_mask,
                                          const uint64_t *__restrict__ maskT, Rec *__restrict__ out) {
  constexpr int EPV = Tr<DT>::EPV, ES = Tr<DT>::ES, NVC = Tr<DT>::NVC;
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int n_items = n_pairs * nch;
  for (int it = gw; it < n_items; it += nw) {
    const int pr = __builtin_amdgcn_readfirstlane(it / nch), c = __builtin_amdgcn_readfirstlane(it % nch);
    const int row = pair_row[pr];
    const char *rowp = logits + (int64_t)row * ld * ES;
    const int e_base = c * CH;
    const int nv_valid = min(NVC, (V - e_base + 64 * EPV - 1) / (64 * EPV));  // vectors with at least one valid lane
    float x[64];
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int e0 = e_base + (i * 64 + lane) * EPV;
      u32x4_t r;
      if (e0 + EPV <= V) {
        r = *reinterpret_cast<const u32x4_t *>(rowp + (int64_t)e0 * ES);
      } else {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = DT == 0 ? 0xff800000u : 0xff80ff80u;
        if (e0 < V) {
          for (int k = 0; k < V - e0; ++k) {
            if (DT == 0) w[k] = *reinterpret_cast<const uint32_t *>(rowp + (int64_t)(e0 + k) * 4);
            else {
              const uint32_t h = *reinterpret_cast<const uint16_t *>(rowp + (int64_t)(e0 + k) * 2);
              w[k >> 1] = (k & 1) ? ((w[k >> 1] & 0xffffu) | (h << 16)) : ((w[k >> 1] & 0xffff0000u) | h);
            }
          }
        }
        r = u32x4_t{w[0], w[1], w[2], w[3]};
      }
      if constexpr (DT == 0) {
        x[i * 4 + 0] = __uint_as_float(r.x); x[i * 4 + 1] = __uint_as_float(r.y);
        x[i * 4 + 2] = __uint_as_float(r.z); x[i * 4 + 3] = __uint_as_float(r.w);
      } else {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          x[i * 8 + 2 * k] = __uint_as_float(w[k] << 16);
          x[i * 8 + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 64; j += 2) m = fmaxf(m, fmaxf(x[j], x[j + 1]));
    m = wave_max(m);
    const float Nc = __builtin_fmaf(m, kLog2e, kMagic) - kMagic;
    if constexpr (MODE == 0) {
      if (lane == 63) out[it].Nc = Nc;
      continue;
    }
    const float magicN = kMagic - Nc;
    const uint64_t *mt = nullptr;
    if constexpr (MASKED) mt = maskT + ((int64_t)pair_mask[pr] * nch + c) * 64;
    float A0 = 32.f, A1 = 32.f, B0 = 0x1p-13f, B1 = 0x1p-13f, Am0 = 32.f, Am1 = 32.f, Bm0 = 0x1p-13f, Bm1 = 0x1p-13f;
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      if (i < nv_valid) {
        float t[EPV];
#pragma unroll
        for (int k = 0; k < EPV; ++k) t[k] = term(x[i * EPV + k], magicN);
#pragma unroll
        for (int h = 0; h < EPV / 4; ++h) {
          uint64_t M0 = 0, M1 = 0, M2 = 0, M3 = 0;
          if constexpr (MASKED) {
            M0 = mt[i * EPV + 4 * h + 0]; M1 = mt[i * EPV + 4 * h + 1];
            M2 = mt[i * EPV + 4 * h + 2]; M3 = mt[i * EPV + 4 * h + 3];
          }
          acc4<MASKED>(t[4 * h], t[4 * h + 1], t[4 * h + 2], t[4 * h + 3], A0, B0, A1, B1, Am0, Bm0, Am1, Bm1, M0,
                       M1, M2, M3);
        }
      }
    }
    uint32_t pA = (__float_as_uint(A0) - 0x42000000u) + (__float_as_uint(A1) - 0x42000000u);
    uint32_t pB = (__float_as_uint(B0) - 0x39000000u) + (__float_as_uint(B1) - 0x39000000u);
    uint32_t pAm = (__float_as_uint(Am0) - 0x42000000u) + (__float_as_uint(Am1) - 0x42000000u);
    uint32_t pBm = (__float_as_uint(Bm0) - 0x39000000u) + (__float_as_uint(Bm1) - 0x39000000u);
    pA = wave_sum_u32(pA);
    pB = wave_sum_u32(pB);
    if constexpr (MASKED) {
      pAm = wave_sum_u32(pAm);
      pBm = wave_sum_u32(pBm);
    }
    if (lane == 63) {
      Rec r;
      r.Nc = Nc; r.pA = pA; r.pB = pB; r.pAm = pAm; r.pBm = pBm;
      out[it] = r;
    }
  }
}

// ---------------------------------------------------------------- host reference
static float bf16_to_f(uint16_t h) { union { uint32_t u; float f; } v; v.u = (uint32_t)h << 16; return v.f; }
static uint16_t f_to_bf16(float f) { union { uint32_t u; float f; } v; v.f = f; return (uint16_t)((v.u + 0x7fff + ((v.u >> 16) & 1)) >> 16); }

static void host_chunk(const float *x, int n, const uint8_t *allow, float &Nc, uint64_t &S, uint64_t &Sm) {
  float m = -INFINITY;
  for (int j = 0; j < n; ++j) m = fmaxf(m, x[j]);
  Nc = fmaf(m, kLog2e, kMagic) - kMagic;
  const float magicN = kMagic - Nc;
  S = Sm = 0;
  for (int j = 0; j < n; ++j) {
    const float tm = fmaf(x[j], kLog2e, magicN);
    const float negn = magicN - tm;
    union { float f; uint32_t u; } b; b.f = tm;
    const int np = (int)(b.u - kMagicBits);
    const float f = fmaf(x[j], kLog2e, negn);
    float p = c_of(C5);
    p = fmaf(p, f, c_of(C4)); p = fmaf(p, f, c_of(C3)); p = fmaf(p, f, c_of(C2)); p = fmaf(p, f, c_of(C1));
    p = fmaf(p, f, c_of(C0));
    if (!(p > 0.f)) p = 0.f;
    if (p > 1.f) p = 1.f;
    const float t = ldexpf(p, np < -200 ? -200 : np);
    const uint64_t q = (uint64_t)floor(ldexp((double)t, 36));
    S += q;
    if (allow[j]) Sm += q;
  }
}

template <int DT, bool MASKED, int MODE>
static float run(const char *name, const void *d_logits, int64_t ld, int V, int n, int nch, const int *d_row,
                 const int *d_mid, const uint64_t *d_mt, Rec *d_out, int wgs, int reps, size_t copy_stride, int ncopies) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL((k1<DT, MASKED, MODE>), dim3(wgs), dim3(256), 0, 0, (const char *)d_logits + (w % ncopies) * copy_stride,
                       ld, V, n, nch, d_row, d_mid, d_mt, d_out);
  CK(hipDeviceSynchronize());
  float best = 1e9f, tot = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k1<DT, MASKED, MODE>), dim3(wgs), dim3(256), 0, 0, (const char *)d_logits + (r % ncopies) * copy_stride,
                       ld, V, n, nch, d_row, d_mid, d_mt, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms);
    tot += ms;
  }
  const double bytes = (double)n * V * Tr<DT>::ES;
  printf("%-34s wgs %5d: best %7.1f us  mean %7.1f us  -> %.2f TB/s (best)\n", name, wgs, best * 1e3, tot / reps * 1e3,
         bytes / (best * 1e-3) / 1e12);
  return best;
}

template <int DT>
static void bench(int n, int V, int64_t ld) {
  constexpr int ES = Tr<DT>::ES, EPV = Tr<DT>::EPV;
  const int nch = (V + CH - 1) / CH;
  const size_t elems = (size_t)n * ld;
  const int ncopies = 3;
  std::vector<float> hx(elems);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < elems; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    float g = 0;
    for (int k = 0; k < 4; ++k) g += (float)((s >> (16 * k)) & 0xffff) / 65536.f;
    hx[i] = (g - 2.f) * 5.2f;
    if ((s & 0xfff) == 7) hx[i] += 25.f;
    if ((s & 0xffff) == 9) hx[i] = -INFINITY;
  }
  std::vector<uint16_t> hb;
  if (DT == 1) {
    hb.resize(elems);
    for (size_t i = 0; i < elems; ++i) { hb[i] = f_to_bf16(hx[i]); hx[i] = bf16_to_f(hb[i]); }
  }
  char *d_logits;
  const size_t stride = (elems * ES + 255) & ~(size_t)255;
  CK(hipMalloc(&d_logits, stride * ncopies + 64));
  for (int c = 0; c < ncopies; ++c)
    CK(hipMemcpy(d_logits + c * stride, DT == 0 ? (const void *)hx.data() : (const void *)hb.data(), elems * ES, hipMemcpyHostToDevice));
  // two masks, ~1/3 forbidden; transposed layout [mask][chunk][vector][comp] of 64-bit lane words
  const int K = 2;
  std::vector<uint8_t> allow((size_t)K * V);
  for (size_t i = 0; i < allow.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; allow[i] = (s % 3) != 0; }
  std::vector<uint64_t> mt((size_t)K * nch * 64, 0);
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < V; ++j)
      if (allow[(size_t)k * V + j]) {
        const int c = j / CH, r = j % CH, i = r / (64 * EPV), l = (r % (64 * EPV)) / EPV, kk = r % EPV;
        mt[((size_t)k * nch + c) * 64 + i * EPV + kk] |= 1ull << l;
      }
  uint64_t *d_mt;
  CK(hipMalloc(&d_mt, mt.size() * 8));
  CK(hipMemcpy(d_mt, mt.data(), mt.size() * 8, hipMemcpyHostToDevice));
  std::vector<int> row(n), mid(n);
  for (int i = 0; i < n; ++i) { row[i] = i; mid[i] = i & 1; }
  int *d_row, *d_mid;
  CK(hipMalloc(&d_row, n * 4)); CK(hipMalloc(&d_mid, n * 4));
  CK(hipMemcpy(d_row, row.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_mid, mid.data(), n * 4, hipMemcpyHostToDevice));
  Rec *d_out;
  CK(hipMalloc(&d_out, (size_t)n * nch * sizeof(Rec)));
  printf("== %s n=%d V=%d ld=%lld  (%d chunks/row, %.1f MB)\n", DT == 0 ? "f32" : "bf16", n, V, (long long)ld, nch,
         (double)n * V * ES / 1e6);
  for (int wgs : {1024, 2048, 4096}) run<DT, false, 0>("loads+max", d_logits, ld, V, n, nch, d_row, d_mid, d_mt, d_out, wgs, 20, stride, ncopies);
  for (int wgs : {1024, 2048, 4096}) run<DT, false, 1>("stats unmasked", d_logits, ld, V, n, nch, d_row, d_mid, d_mt, d_out, wgs, 20, stride, ncopies);
  for (int wgs : {1024, 2048, 4096}) run<DT, true, 1>("stats masked", d_logits, ld, V, n, nch, d_row, d_mid, d_mt, d_out, wgs, 20, stride, ncopies);
  // check (masked run is the last one; logits copy index of the last rep = (reps-1) % ncopies, all copies equal)
  std::vector<Rec> out((size_t)n * nch);
  CK(hipMemcpy(out.data(), d_out, out.size() * sizeof(Rec), hipMemcpyDeviceToHost));
  int bad = 0, checked = 0;
  for (int p = 0; p < n; p += (n / 16 ? n / 16 : 1))
    for (int c = 0; c < nch; ++c) {
      const int e0 = c * CH, cnt = std::min(CH, V - e0);
      float Nc; uint64_t S, Sm;
      host_chunk(&hx[(size_t)p * ld + e0], cnt, &allow[(size_t)mid[p] * V + e0], Nc, S, Sm);
      const Rec &r = out[(size_t)p * nch + c];
      const uint64_t gS = ((uint64_t)r.pA << 18) + r.pB, gSm = ((uint64_t)r.pAm << 18) + r.pBm;
      ++checked;
      if (Nc != r.Nc || S != gS || Sm != gSm) {
        if (bad < 8) printf("  MISMATCH row %d chunk %d: N %g/%g S %llu/%llu Sm %llu/%llu\n", p, c, Nc, r.Nc,
                            (unsigned long long)S, (unsigned long long)gS, (unsigned long long)Sm, (unsigned long long)gSm);
        ++bad;
      }
    }
  printf("  check: %d chunks, %d mismatches\n", checked, bad);
  hipFree(d_logits); hipFree(d_mt); hipFree(d_row); hipFree(d_mid); hipFree(d_out);
}

int main(int argc, char **argv) {
  bench<0>(1024, 50257, 50257);
  bench<1>(512, 128256, 128256);
  bench<1>(1024, 50257, 50257);
  bench<0>(1024, 50257, 50304);
  return 0;
}
