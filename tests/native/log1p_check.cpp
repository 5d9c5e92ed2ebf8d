// Host check of csrc/glb_log1p.hpp against the C library (built and run by tests/test_mt_cpu.py with g++; no GPU):
//   - exponential_from_words (glibc's log1p restated, without the division that is 0 / u for the stream's arguments) gives
//     the float of (float)(-log1p(-u)) for every argument tried;
//   - exponential_fast (table + series) gives the same float whenever it does not ask for the exact form, and asks rarely.
// Arguments: uniform 53-bit k, k shifted down (u small, every magnitude), 2^53 - 1 - small (u next to 1), and the neighbours
// of the table's interval ends and of the small-u switch.  Prints one line: n redo undetected exact_bad.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../genlm-backend_amd/csrc/glb_log1p.hpp"

static long redo_n = 0, bad = 0, exact_bad = 0, n_done = 0;
static double tinv[glb::kLog1pEntries], thi[glb::kLog1pEntries];

static void one(uint64_t k, uint32_t junk) {
  k &= (1ull << 53) - 1;
  const uint32_t first = (uint32_t)(k >> 32) | (junk << 21), second = (uint32_t)k;  // (the top 11 bits of random64() are not used)
  const double u = (double)k * (1.0 / 9007199254740992.0);
  const float want = (float)(-log1p(-u));
  const float ex = glb::exponential_from_words(first, second);
  if (ex != want || std::signbit(ex) != std::signbit(want)) ++exact_bad;
  bool redo;
  const float f = glb::exponential_fast(first, second, tinv, thi, &redo);
  if (redo) ++redo_n;
  else if (f != want || std::signbit(f) != std::signbit(want)) ++bad;
  ++n_done;
}

int main(int argc, char **argv) {
  for (int i = 0; i < glb::kLog1pEntries; ++i) glb::log1p_table_entry(i, &tinv[i], &thi[i]);
  const long n = argc > 1 ? atol(argv[1]) : 10000000;
  uint64_t s = 88172645463325252ull;
  for (long it = 0; it < n; ++it) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    uint64_t k = s & ((1ull << 53) - 1);
    if (it % 7 == 0) k >>= (s >> 58) % 53;
    if (it % 11 == 0) k = ((1ull << 53) - 1) - (k >> ((s >> 57) % 53));
    one(k, (uint32_t)(s >> 40));
  }
  // w = 1 - u at the ends of the table's intervals (every exponent of w that occurs), and u round 2^-7
  for (int e = 0; e <= 52; ++e)
    for (int i = 0; i <= 128; ++i)
      for (int d = -3; d <= 3; ++d) {
        const double w = std::ldexp(1.0 + i / 128.0, -e) + d * std::ldexp(1.0, -53 - (e > 0 ? e - 1 : 0));
        if (w <= 0.0 || w > 1.0) continue;
        const double k = (1.0 - w) * 9007199254740992.0;
        if (k >= 0 && k < 9007199254740992.0) one((uint64_t)k, 0);
      }
  for (int d = -64; d <= 64; ++d) one((1ull << 46) + (uint64_t)(int64_t)d, 0);
  for (uint64_t k = 0; k < 4096; ++k) one(k, 0), one(((1ull << 53) - 1) - k, 0);
  std::printf("%ld %ld %ld %ld\n", n_done, redo_n, bad, exact_bad);
  return 0;
}
