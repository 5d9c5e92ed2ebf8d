// glb_mt.hip — torch's CPU generator on the device: MT19937 -> exponential_ rows for the parity draw (include/glb.h:
// glb_mt19937_*).  The reference draws every token with torch.multinomial on the CPU (README.md:87, base.py:136-141):
// V exponentials per particle from ONE serial MT19937 stream, 2 V words each, particles in resolution order.  To draw the
// same tokens without generating 2 V N words serially on the host (and shipping N V floats over PCIe), the stream is
// entered at N places at once:
//
//   * The generator's untempered words x[i] obey a linear recurrence over GF(2): the window W(o) = x[o .. o+623] moves by
//     T, W(o + 1) = T W(o), and T^J = g_J(T) with g_J(t) = t^J mod phi(t), phi the minimal polynomial of T (degree 19937).
//     With g_J = sum c_i t^i:  x[o + J + j] = XOR over {i : c_i = 1} of x[o + i + j]  - a window J words ahead is a
//     GF(2) convolution of the sequence that starts at the window (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer:
//     "Efficient jump ahead for F2-linear random number generators", 2008).
//   * Host, once per vocabulary (glb_mt19937_jump_polys): phi by Berlekamp-Massey, g_{r s} for r < n_small and
//     g_{m n_small s} for m < n_big, s = 2 V words = one particle's stride.
//   * mt_jump_kernel: a workgroup runs the recurrence from its source window through a ring in LDS (227 words per step:
//     the recurrence reaches back 227) up to the part of the sequence it convolves (a quarter; a sixteenth in the first
//     launch), and each of its four waves convolves that part with the matching part of one polynomial - lane l keeps 11
//     consecutive output words, a 42-word sliding window of the sequence serves 32 coefficients; a window is the XOR of its
//     partial planes.  Two launches: the base window -> every n_small-th row's window, those -> every row's window.
//   * mt_rows_kernel: a workgroup per output row runs the recurrence from its window (454 words per round through a
//     2048-word ring in LDS), tempers, pairs the words as random64() and turns them into E = (float)(-log1p(-u)): glibc's
//     log1p restated operation for operation (glb_log1p.hpp) where a cheaper double cannot tell the float - one value in
//     10^5 -, a table and a series everywhere else (the same float, by construction: exponential_fast); one more block
//     leaves the window after the last consumed row as the stream's new position.
// Everything is integer work except the logarithm; results are bit-identical to the host's serial
// glb_mt19937_exponential_f32 (tests/test_mt_gpu.py), which is pinned against torch (tests/test_oracle.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/glb.h"
#include "glb_common.hpp"
#include "glb_log1p.hpp"

namespace {

using glb::api_fail;
using glb::api_hip_fail;

constexpr int kN = 624, kM = 397, kDeg = 19937;
constexpr int kPW = GLB_MT_POLY_WORDS;  // 312 64-bit words: coefficients 0 .. 19967
constexpr int kStep = kN - kM;          // 227: how many new words one step of the recurrence can make at once
constexpr int kPerLane = 11;             // output words a lane of mt_jump_kernel keeps (57 lanes x 11 >= 624; stride 11: no bank conflicts)
// A polynomial's coefficients are applied in PARTS parts by as many workgroups that never meet (a window is the XOR of the
// PARTS planes they leave).  kParts = 4 for the launch that makes every row's window (a thousand windows: throughput - more
// parts would run the recurrence up to their range more often than they save: 178 -> 240 us with 16); kPartsFew = 16 for the
// launch before it, which makes the 32 windows of every n_small-th row from the stream's position (32 workgroups with four
// parts: the latency of ONE workgroup's 156 coefficient words, 178 us; 128 workgroups of 39 words: 50 us).
constexpr int kParts = 4, kPartsFew = 16;
template <int PARTS>
struct JumpShape {
  static_assert(kN % PARTS == 0, "a part is a whole number of coefficient words");
  static constexpr int kPartWords = kN / PARTS;  // 32-bit coefficient words a part: 156 (4992 coefficients) / 39
  static constexpr int kPartSeq = kPartWords * 32 + 56 * kPerLane + 32 + kPerLane - 1;  // sequence words a part reads: 5650 / 1906
  static constexpr int kRingWords = PARTS >= 16 ? 4096 : 8192;  // LDS words of a jumping workgroup
  static_assert(kRingWords >= kPartSeq + kN + kStep, "the ring holds a part's sequence and the recurrence's reach");
};
constexpr int kRing = 2048;

__host__ __device__ static inline uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__host__ __device__ static inline uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

static void mt_seed_words(uint32_t *mt, uint64_t seed) {
  mt[0] = (uint32_t)seed;
  for (int i = 1; i < kN; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

// ---- GF(2)[t] mod phi on the host ------------------------------------------------------------------------------------
struct Gf2 {
  // phi << p for p = 0 .. 63, kPW + 1 words each (bit 19937 + p is the top one)
  std::vector<uint64_t> phi_sh;
  bool ok = false;
};

static inline int get_bit(const uint64_t *a, int i) { return (int)((a[i >> 6] >> (i & 63)) & 1ull); }

// Berlekamp-Massey over GF(2) on the lowest bit of x[n + 1]: the sequence's minimal polynomial is that of T (phi is
// irreducible, so every non-zero component sequence has it).
static bool min_poly(std::vector<uint64_t> &phi) {
  const int NB = 2 * kDeg + 192;
  std::vector<uint32_t> x(NB + 2 + kN);
  mt_seed_words(x.data(), 5489u);
  for (int i = 0; i + kN < (int)x.size(); ++i) x[i + kN] = mt_twist(x[i], x[i + 1], x[i + kM]);
  const int W = (NB + 63) / 64 + 2;
  // rs bit (NB - 1 - n) = s[n]: the window s[n - L .. n] read backwards is a run of ascending bits of rs
  std::vector<uint64_t> rs(W + 1, 0), C(W, 0), B(W, 0), T(W, 0);
  for (int n = 0; n < NB; ++n)
    if (x[n + 1] & 1u) rs[(NB - 1 - n) >> 6] |= 1ull << ((NB - 1 - n) & 63);
  C[0] = B[0] = 1;
  int L = 0, m = 1;
  for (int n = 0; n < NB; ++n) {
    const int off = NB - 1 - n, wo = off >> 6, bo = off & 63;
    uint64_t acc = 0;
    for (int w = 0; w <= (L >> 6); ++w) {
      const uint64_t lo = rs[wo + w] >> bo;
      const uint64_t hi = bo ? (rs[wo + w + 1] << (64 - bo)) : 0ull;
      acc ^= C[w] & (lo | hi);
    }
    if (!(__builtin_popcountll(acc) & 1)) {
      ++m;
      continue;
    }
    const bool grow = 2 * L <= n;
    if (grow) T = C;
    const int ws = m >> 6, bs = m & 63;  // C ^= B << m
    for (int w = W - 1; w >= ws; --w) {
      uint64_t v = B[w - ws] << bs;
      if (bs && w - ws - 1 >= 0) v |= B[w - ws - 1] >> (64 - bs);
      C[w] ^= v;
    }
    if (grow) {
      L = n + 1 - L;
      B.swap(T);
      m = 1;
    } else {
      ++m;
    }
  }
  if (L != kDeg) return false;
  phi.assign(kPW + 1, 0);
  for (int k = 0; k <= kDeg; ++k)  // phi_k = C_{L - k}
    if (get_bit(C.data(), kDeg - k)) phi[k >> 6] |= 1ull << (k & 63);
  return true;
}

static const Gf2 &field() {
  static Gf2 f;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<uint64_t> phi;
    if (!min_poly(phi)) return;
    f.phi_sh.assign((size_t)64 * (kPW + 2), 0);
    for (int p = 0; p < 64; ++p) {
      uint64_t *d = &f.phi_sh[(size_t)p * (kPW + 2)];
      for (int w = 0; w <= kPW; ++w) {
        d[w] |= phi[w] << p;
        if (p) d[w + 1] |= phi[w] >> (64 - p);
      }
    }
    f.ok = true;
  });
  return f;
}

// out = a b mod phi (kPW words each; out may alias a or b)
static void mulmod(const Gf2 &F, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  std::vector<uint64_t> prod(2 * kPW + 4, 0), bs(kPW + 2);
  for (int p = 0; p < 64; ++p) {
    bs[kPW] = bs[kPW + 1] = 0;
    for (int w = 0; w < kPW; ++w) bs[w] = b[w] << p;
    if (p)
      for (int w = 0; w < kPW; ++w) bs[w + 1] |= b[w] >> (64 - p);
    for (int wi = 0; wi < kPW; ++wi)
      if ((a[wi] >> p) & 1ull) {
        uint64_t *d = &prod[wi];
        for (int w = 0; w <= kPW; ++w) d[w] ^= bs[w];
      }
  }
  for (int i = 2 * kDeg - 2; i >= kDeg; --i)
    if (get_bit(prod.data(), i)) {
      const int sh = i - kDeg;
      const uint64_t *ph = &F.phi_sh[(size_t)(sh & 63) * (kPW + 2)];
      uint64_t *d = &prod[sh >> 6];
      for (int w = 0; w <= kPW + 1; ++w) d[w] ^= ph[w];
    }
  memcpy(out, prod.data(), kPW * sizeof(uint64_t));
}

// r = t^J mod phi
static void pow_t(const Gf2 &F, uint64_t J, uint64_t *r) {
  memset(r, 0, kPW * sizeof(uint64_t));
  r[0] = 1;
  int top = 63;
  while (top > 0 && !((J >> top) & 1ull)) --top;
  for (int b = top; b >= 0; --b) {
    mulmod(F, r, r, r);
    if ((J >> b) & 1ull) {  // times t: shift by one, reduce when the degree reaches 19937
      uint64_t carry = 0;
      for (int w = 0; w < kPW; ++w) {
        const uint64_t nc = r[w] >> 63;
        r[w] = (r[w] << 1) | carry;
        carry = nc;
      }
      if (get_bit(r, kDeg)) {
        const uint64_t *ph = &F.phi_sh[0];
        for (int w = 0; w < kPW; ++w) r[w] ^= ph[w];
      }
    }
  }
}

// window_out = g(T) window_in on the host, the way the kernel does it (validation and tests)
static void jump_host(const uint32_t *win, const uint64_t *poly, uint32_t *out) {
  std::vector<uint32_t> seq(kN + kDeg + 1);
  memcpy(seq.data(), win, kN * sizeof(uint32_t));
  for (int i = 0; i + kN < (int)seq.size(); ++i) seq[i + kN] = mt_twist(seq[i], seq[i + 1], seq[i + kM]);
  memset(out, 0, kN * sizeof(uint32_t));
  for (int i = 0; i < kDeg; ++i)
    if (get_bit(poly, i))
      for (int j = 0; j < kN; ++j) out[j] ^= seq[i + j];
}

// ---- device ------------------------------------------------------------------------------------------------------------
// Block (src, quad, part): waves 0..3 apply coefficients [part * 4992, (part + 1) * 4992) of polynomials 4 quad + wave to
// source window src.  A window is kept as PARTS PLANES whose XOR it is (dst[part][src * n_poly + p]): the parts of a
// polynomial are workgroups that never meet, each with its part of the sequence in LDS (PARTS = 4: 32 KB, five workgroups a CU,
// so that one wave's LDS waits and taken branches are another's issue slots; the whole sequence in one workgroup - 82 KB,
// one wave a SIMD - took 577 us a launch, VALU idle two thirds of the time).  The source window is the XOR of its own planes.
// need_rows (nullable): one word per destination window, nonzero = somebody reads it.  A sharded population enters ONE global
// stream (every rank deals the whole population's rows, DeviceSIS._parity_noise) but reads only its own particles' rows:
// a block none of whose four windows is needed leaves before it has done anything.
template <int PARTS>
__global__ __launch_bounds__(256) void mt_jump_kernel(const uint32_t *__restrict__ src, int src_planes, int64_t src_plane_stride,
                                                      const uint64_t *__restrict__ polys, int n_poly, uint32_t *__restrict__ dst,
                                                      int n_dst, int64_t dst_plane_stride, const int32_t *__restrict__ need_rows,
                                                      int need_group) {
  constexpr int kPartWords = JumpShape<PARTS>::kPartWords, kPartSeq = JumpShape<PARTS>::kPartSeq, kJumpRing = JumpShape<PARTS>::kRingWords;
  __shared__ uint32_t ring[kJumpRing];
  const int quads = (n_poly + 3) >> 2;
  const int part = (int)blockIdx.x % PARTS, sq = (int)blockIdx.x / PARTS;
  const int s = sq / quads, q = sq % quads;
  const int t = (int)threadIdx.x;
  if (need_rows) {  // (block-uniform: decided before the first barrier)
    // destination window d stands for the need_group consecutive rows [d * need_group, (d + 1) * need_group)
    int any = 0;
    for (int w = 0; w < 4; ++w) {
      const int d = s * n_poly + q * 4 + w;
      if (q * 4 + w >= n_poly || d >= n_dst) continue;
      for (int g = 0; g < need_group; ++g) any |= need_rows[(size_t)d * need_group + g];
    }
    if (!any) return;
  }
  // word n of the sequence lives at ring[(n - lo) & (kJumpRing - 1)]: the part's range [lo, lo + kPartSeq) is contiguous
  const int lo = part * kPartWords * 32;
  for (int i = t; i < kN; i += 256) {  // (the planes' loads go out together: one after the other, sixteen planes are sixteen trips)
    uint32_t v = 0;
    for (int pl0 = 0; pl0 < src_planes; pl0 += 8) {
      uint32_t x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = pl0 + k < src_planes ? src[(size_t)(pl0 + k) * src_plane_stride + (size_t)s * kN + i] : 0u;
#pragma unroll
      for (int k = 0; k < 8; ++k) v ^= x[k];
    }
    ring[(i - lo) & (kJumpRing - 1)] = v;
  }
  __syncthreads();
  const int need = lo + kPartSeq;  // words [0, need) have to exist
  for (int base = 0; base + kN < need; base += kStep) {
    if (t < kStep) {
      const int i = base + t - lo;
      ring[(i + kN) & (kJumpRing - 1)] = mt_twist(ring[i & (kJumpRing - 1)], ring[(i + 1) & (kJumpRing - 1)], ring[(i + kM) & (kJumpRing - 1)]);
    }
    __syncthreads();
  }
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int p = q * 4 + wave;
  const int d = s * n_poly + p;
  if (p >= n_poly || d >= n_dst) return;  // (no barrier below)
  // coefficient i: 32-bit word i / 32, bit i % 32; this part's words
  const uint32_t *c = reinterpret_cast<const uint32_t *>(polys + (size_t)p * kPW) + part * kPartWords;
  const int j0 = lane * kPerLane;
  if (j0 >= kN) return;
  uint32_t acc[kPerLane];
#pragma unroll
  for (int k = 0; k < kPerLane; ++k) acc[k] = 0;
  for (int wi = 0; wi < kPartWords; ++wi) {
    const uint32_t cw = __builtin_amdgcn_readfirstlane(c[wi]);
    if (cw == 0) continue;
    uint32_t win[32 + kPerLane - 1];
    const uint32_t *sp = ring + 32 * wi + j0;
#pragma unroll
    for (int u = 0; u < 32 + kPerLane - 1; ++u) win[u] = sp[u];
#pragma unroll
    for (int b = 0; b < 32; ++b)
      if ((cw >> b) & 1u) {
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) acc[k] ^= win[b + k];
      }
  }
  uint32_t *o = dst + (size_t)part * dst_plane_stride + (size_t)d * kN;
#pragma unroll
  for (int k = 0; k < kPerLane; ++k)
    if (j0 + k < kN) o[j0 + k] = acc[k];
}

// need[k] = 1 for every stream row an output row takes, and for the row the stream stands at afterwards (window_out)
__global__ __launch_bounds__(256) void mt_need_kernel(const int32_t *__restrict__ row_slot, int n_out, const int32_t *__restrict__ n_draw_dev,
                                                      int n_draw_host, int want_window_out, int32_t *__restrict__ need, int n_need) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i < n_out) {
    const int sl = row_slot[i];
    if (sl >= 0 && sl < n_need) need[sl] = 1;
  }
  if (i == 0 && want_window_out) {
    int nd = n_draw_dev ? *n_draw_dev : n_draw_host;
    nd = nd < 0 ? 0 : (nd > n_draw_host ? n_draw_host : nd);
    if (nd < n_need) need[nd] = 1;
  }
}

// Block r < n_out: output row r = the V exponentials of stream row row_slot[r] (window windows[slot]); ones for slot < 0.
// Block n_out: window_out = windows[n_draw].
__global__ __launch_bounds__(256) void mt_rows_kernel(const uint32_t *__restrict__ windows, int64_t plane_stride, int n_win,
                                                      const int32_t *__restrict__ row_slot,
                                                      int n_out, int64_t V, float *__restrict__ out, int64_t ld,
                                                      const int32_t *__restrict__ n_draw_dev, int n_draw_host, uint32_t *window_out) {
  __shared__ uint32_t st[kRing];
  __shared__ double tab_inv[glb::kLog1pEntries], tab_hi[glb::kLog1pEntries];  // exponential_fast's table (glb_log1p.hpp)
  const int r = (int)blockIdx.x, t = (int)threadIdx.x;
  if (r == n_out) {
    int nd = n_draw_dev ? *n_draw_dev : n_draw_host;
    nd = nd < 0 ? 0 : (nd > n_win - 1 ? n_win - 1 : nd);
    const uint32_t *w = windows + (size_t)nd * kN;
    for (int i = t; i < kN; i += 256) {
      uint32_t v = 0;
#pragma unroll
      for (int pl = 0; pl < kParts; ++pl) v ^= w[(size_t)pl * plane_stride + i];
      window_out[i] = v;
    }
    return;
  }
  const int slot = row_slot ? row_slot[r] : r;
  float *o = out + (size_t)r * ld;
  if (slot < 0 || slot >= n_win) {  // a particle that draws nothing: any finite noise will do (a slot past the plan: NaN, loudly)
    const float fill = slot < 0 ? 1.0f : __builtin_nanf("");
    for (int64_t v = t; v < V; v += 256) o[v] = fill;
    return;
  }
  const uint32_t *w = windows + (size_t)slot * kN;
  for (int i = t; i < kN; i += 256) {
    uint32_t v = 0;
#pragma unroll
    for (int pl = 0; pl < kParts; ++pl) v ^= w[(size_t)pl * plane_stride + i];
    st[i] = v;
  }
  if (t < glb::kLog1pEntries) glb::log1p_table_entry(t, &tab_inv[t], &tab_hi[t]);
  __syncthreads();
  int pos = 0;  // st[(pos + i) & (kRing - 1)] = x[row start + consumed + i]
  for (int64_t v0 = 0; v0 < V; v0 += kStep) {
    if (t < kStep) {
      const int i = pos + t;
      st[(i + kN) & (kRing - 1)] = mt_twist(st[i & (kRing - 1)], st[(i + 1) & (kRing - 1)], st[(i + kM) & (kRing - 1)]);
    }
    __syncthreads();
    if (t < kStep) {
      const int i = pos + kStep + t;
      st[(i + kN) & (kRing - 1)] = mt_twist(st[i & (kRing - 1)], st[(i + 1) & (kRing - 1)], st[(i + kM) & (kRing - 1)]);
    }
    __syncthreads();
    if (t < kStep && v0 + t < V) {
      const int i = pos + kN + 2 * t;
      const uint32_t first = mt_temper(st[i & (kRing - 1)]), second = mt_temper(st[(i + 1) & (kRing - 1)]);
      // the float by the short form; the one value in 10^5 whose double lies too near a rounding boundary for it to tell (one
      // wave in 1 500) by glibc's own form
      bool redo;
      float e = glb::exponential_fast(first, second, tab_inv, tab_hi, &redo);
      if (redo) e = glb::exponential_from_words(first, second);
      o[v0 + t] = e;
    }
    pos = (pos + 2 * kStep) & (kRing - 1);
    // (the next round writes ring positions pos + 624 + [0, 454) of the NEW pos - 454 past what this round's readers
    // touch - and its first barrier comes before anything reads them: no barrier needed here)
  }
}

// out[i, 0..V) = src[row_slot[i], 0..V) (ones for a negative slot, NaN for one past the rows that exist): the rows were
// generated ahead of time in stream order (glb_mt_rows_args.rows_from)
__global__ __launch_bounds__(256) void mt_rows_copy_kernel(const float *__restrict__ src, int64_t src_ld, int n_src, const int32_t *__restrict__ row_slot,
                                                           int64_t V, float *__restrict__ out, int64_t ld) {
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // (a row starts on any word)
  const int r = (int)blockIdx.x, t = (int)threadIdx.x;
  const int slot = row_slot ? row_slot[r] : r;
  float *o = out + (size_t)r * ld;
  const int64_t v4 = V & ~(int64_t)3;
  if (slot < 0 || slot >= n_src) {
    const float fill = slot < 0 ? 1.0f : __builtin_nanf("");
    for (int64_t v = t; v < V; v += 256) o[v] = fill;
    return;
  }
  const float *s = src + (size_t)slot * src_ld;
  for (int64_t v = 4 * (int64_t)t; v < v4; v += 4 * 256 * 4) {
    f32x4_u x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v + k * 1024 < v4) x[k] = *reinterpret_cast<const f32x4_u *>(s + v + k * 1024);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v + k * 1024 < v4) *reinterpret_cast<f32x4_u *>(o + v + k * 1024) = x[k];
  }
  for (int64_t v = v4 + t; v < V; v += 256) o[v] = s[v];
}

template <int PARTS>
int launch_jump(const uint32_t *src, int src_planes, int64_t src_plane_stride, int n_src, const uint64_t *polys, int n_poly,
                uint32_t *dst, int n_dst, int64_t dst_plane_stride, hipStream_t st, const int32_t *need = nullptr, int need_group = 1) {
  const int quads = (n_poly + 3) / 4;
  hipLaunchKernelGGL(mt_jump_kernel<PARTS>, dim3((unsigned)(n_src * quads * PARTS)), dim3(256), 0, st, src, src_planes, src_plane_stride, polys,
                     n_poly, dst, n_dst, dst_plane_stride, need, need_group);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? GLB_OK : api_hip_fail(e, "mt_jump_kernel");
}

}  // namespace

extern "C" {

// ---- host stream (the serial form the reference runs) -----------------------------------------------------------------
void glb_mt19937_seed(glb_mt19937 *st, uint64_t seed) {
  mt_seed_words(st->mt, seed);
  st->idx = kN;
}

static inline void mt_refill(glb_mt19937 *st) {
  uint32_t *mt = st->mt;
  int i = 0;
  for (; i < kN - kM; ++i) mt[i] = mt_twist(mt[i], mt[i + 1], mt[i + kM]);
  for (; i < kN - 1; ++i) mt[i] = mt_twist(mt[i], mt[i + 1], mt[i - (kN - kM)]);
  mt[kN - 1] = mt_twist(mt[kN - 1], mt[0], mt[kM - 1]);
  st->idx = 0;
}

static inline uint32_t mt_next(glb_mt19937 *st) {
  if (st->idx >= kN) mt_refill(st);
  return mt_temper(st->mt[st->idx++]);
}

int glb_mt19937_exponential_f32(glb_mt19937 *st, float *out, int64_t n) {
  if (!st || !out || n < 0) return api_fail(GLB_EINVAL, "bad arguments");
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t first = mt_next(st), second = mt_next(st);
    out[i] = glb::exponential_from_words(first, second);
  }
  return GLB_OK;
}

// ---- the stream's position as a window, and the polynomials that move it -----------------------------------------------
int glb_mt19937_window(uint64_t seed, uint32_t *out_window) {
  if (!out_window) return api_fail(GLB_EINVAL, "null pointer");
  mt_seed_words(out_window, seed);
  return GLB_OK;
}

int glb_mt19937_jump_polys(int64_t stride_words, int32_t n_small, int32_t n_big, uint64_t *out_polys) {
  if (!out_polys || stride_words <= 0 || n_small < 2 || n_big < 1) return api_fail(GLB_EINVAL, "bad arguments");
  const Gf2 &F = field();
  if (!F.ok) return api_fail(GLB_EUNSUPPORTED, "Berlekamp-Massey did not find a minimal polynomial of degree 19937");
  uint64_t *small = out_polys, *big = out_polys + (size_t)n_small * kPW;
  memset(out_polys, 0, (size_t)(n_small + n_big) * kPW * sizeof(uint64_t));
  small[0] = 1;
  pow_t(F, (uint64_t)stride_words, small + kPW);
  for (int r = 2; r < n_small; ++r) mulmod(F, small + (size_t)(r - 1) * kPW, small + kPW, small + (size_t)r * kPW);
  big[0] = 1;
  if (n_big > 1) {
    mulmod(F, small + (size_t)(n_small - 1) * kPW, small + kPW, big + kPW);
    for (int m = 2; m < n_big; ++m) mulmod(F, big + (size_t)(m - 1) * kPW, big + kPW, big + (size_t)m * kPW);
  }
  return GLB_OK;
}

int glb_mt19937_jump_host(const uint32_t *window_in, const uint64_t *poly, uint32_t *window_out) {
  if (!window_in || !poly || !window_out) return api_fail(GLB_EINVAL, "null pointer");
  jump_host(window_in, poly, window_out);
  return GLB_OK;
}

size_t glb_mt19937_rows_workspace(int64_t max_draw_rows, int32_t n_small) {
  if (max_draw_rows < 0 || n_small < 2) return 0;
  const int64_t n_big = (max_draw_rows + 1 + n_small - 1) / n_small;  // windows are kept as planes: kPartsFew / kParts (csrc/glb_mt.hip)
  // ... and one word per row window behind them: which windows this call's output rows read (mt_need_kernel)
  return (size_t)(n_big * kPartsFew + n_big * n_small * kParts) * kN * sizeof(uint32_t) + (size_t)(n_big * n_small) * sizeof(int32_t);
}

int glb_mt19937_exponential_rows(const glb_mt_rows_args *a, void *hip_stream) {
  if (!a || a->struct_size != sizeof(glb_mt_rows_args)) return api_fail(GLB_EINVAL, "glb_mt_rows_args: struct_size mismatch");
  if (!a->window || !a->polys || a->n_small < 2 || a->n_big < 1 || a->vocab <= 0 || a->max_draw_rows < 0 || a->n_out_rows < 0)
    return api_fail(GLB_EINVAL, "bad arguments");
  if (a->n_out_rows && (!a->out || a->out_ld < a->vocab)) return api_fail(GLB_EINVAL, "out / out_ld");
  if (a->rows_from && a->rows_from_ld < a->vocab) return api_fail(GLB_EINVAL, "rows_from_ld");
  if (a->max_draw_rows + 1 > (int64_t)a->n_big * a->n_small)
    return api_fail(GLB_EINVAL, "max_draw_rows %lld needs more than the %d x %d polynomials handed over", (long long)a->max_draw_rows,
                    a->n_big, a->n_small);
  if (a->n_out_rows > INT32_MAX - 1 || a->max_draw_rows > INT32_MAX - 1) return api_fail(GLB_EINVAL, "too many rows");
  const size_t need = glb_mt19937_rows_workspace(a->max_draw_rows, a->n_small);
  if (!a->workspace || a->workspace_bytes < need) return api_fail(GLB_ENOSPC, "workspace: %zu bytes needed", need);
  hipStream_t st = (hipStream_t)hip_stream;
  const int n_win = (int)a->max_draw_rows + 1;
  const int m_need = (n_win + a->n_small - 1) / a->n_small;
  // (window_out may be `window`: the launches below read `window` before the last one writes window_out)
  const int64_t big_plane = (int64_t)m_need * kN, all_plane = (int64_t)m_need * a->n_small * kN;
  uint32_t *big = (uint32_t *)a->workspace, *all = big + big_plane * kPartsFew;
  const uint32_t *lvl1 = a->window;
  int lvl1_planes = 1;
  int64_t lvl1_stride = 0;
  if (!a->reuse_windows) {  // (reuse_windows: the workspace holds them since the call that generated rows_from)
    // Output rows that take only SOME of the stream's rows (a rank's particles in a sharded population's one global
    // stream: n_out_rows of max_draw_rows): the windows nobody reads are not made - a word per row says which are.
    const int32_t *need = nullptr;
    if (a->row_slot && !a->rows_from && a->n_out_rows * 2 <= a->max_draw_rows) {
      int32_t *flags = (int32_t *)(all + all_plane * kParts);
      const size_t n_flags = (size_t)m_need * a->n_small;
      hipError_t e = hipMemsetAsync(flags, 0, n_flags * sizeof(int32_t), st);
      if (e != hipSuccess) return api_hip_fail(e, "need flags");
      hipLaunchKernelGGL(mt_need_kernel, dim3((unsigned)((a->n_out_rows + 255) / 256 + (a->n_out_rows ? 0 : 1))), dim3(256), 0, st, a->row_slot,
                         (int)a->n_out_rows, a->n_draw, (int)a->max_draw_rows, a->window_out ? 1 : 0, flags, (int)n_flags);
      e = hipGetLastError();
      if (e != hipSuccess) return api_hip_fail(e, "mt_need_kernel");
      need = flags;
    }
    if (m_need > 1) {  // the base window -> the window of every n_small-th row (of the groups of n_small rows that are read)
      int rc = launch_jump<kPartsFew>(a->window, 1, 0, 1, a->polys + (size_t)a->n_small * kPW, m_need, big, m_need, big_plane, st, need, a->n_small);
      if (rc) return rc;
      lvl1 = big;
      lvl1_planes = kPartsFew;
      lvl1_stride = big_plane;
    }
    // -> every row's window (and the one after the last)
    int rc = launch_jump<kParts>(lvl1, lvl1_planes, lvl1_stride, m_need, a->polys, a->n_small, all, n_win, all_plane, st, need, 1);
    if (rc) return rc;
  }
  if (a->rows_from) {  // the rows exist: copy them into place; the stream's new position by the rows kernel's last block alone
    if (a->n_out_rows) {
      hipLaunchKernelGGL(mt_rows_copy_kernel, dim3((unsigned)a->n_out_rows), dim3(256), 0, st, a->rows_from, a->rows_from_ld, (int)a->max_draw_rows,
                         a->row_slot, a->vocab, a->out, a->out_ld);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return api_hip_fail(e, "mt_rows_copy_kernel");
    }
    if (a->window_out) {
      hipLaunchKernelGGL(mt_rows_kernel, dim3(1), dim3(256), 0, st, all, all_plane, n_win, a->row_slot, 0, a->vocab, a->out, a->out_ld, a->n_draw,
                         (int)a->max_draw_rows, a->window_out);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return api_hip_fail(e, "mt_rows_kernel");
    }
    return GLB_OK;
  }
  if (a->n_out_rows || a->window_out) {
    uint32_t *wo = a->window_out;
    const int n_blocks = (int)a->n_out_rows + (wo ? 1 : 0);
    hipLaunchKernelGGL(mt_rows_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, all, all_plane, n_win, a->row_slot, (int)a->n_out_rows,
                       a->vocab, a->out, a->out_ld, a->n_draw, (int)a->max_draw_rows, wo);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return api_hip_fail(e, "mt_rows_kernel");
  }
  return GLB_OK;
}

}  // extern "C"
