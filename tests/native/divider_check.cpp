// Checks rt::make_divider / rt::fast_div (ray_tracing_amd/csrc/rt_internal.h) on the CPU against
// the machine's own division: every divisor in a list that covers 1, powers of two and their
// neighbours, the slice counts and image widths the library uses, and large values, against
// numerators that cover the edges of each quotient step and a pseudo-random sweep.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../ray_tracing_amd/csrc/rt_internal.h"

int main() {
  std::vector<uint32_t> divisors;
  for (uint32_t d = 1; d <= 4100; ++d) divisors.push_back(d);
  for (int b = 12; b < 32; ++b)
    for (int k = -2; k <= 2; ++k) divisors.push_back((uint32_t)((1ull << b) + k));
  const uint32_t extra[] = {7680, 3840, 1920, 65535, 65537, 1000003, 0x7FFFFFFFu, 0x80000001u,
                            0xFFFFFFFEu, 0xFFFFFFFFu, 2073600u, 8294400u};
  for (uint32_t d : extra) divisors.push_back(d);
  unsigned long long checked = 0;
  for (uint32_t d : divisors) {
    const rt::Divider v = rt::make_divider(d);
    auto check = [&](uint32_t n) {
      ++checked;
      if (rt::fast_div(n, v) != n / d) {
        printf("FAIL n=%u d=%u got %u want %u\n", n, d, rt::fast_div(n, v), n / d);
        return false;
      }
      return true;
    };
    // edges of quotient steps near both ends of the range, and a sweep
    for (uint64_t q = 0; q < 40; ++q)
      for (int k = -1; k <= 1; ++k) {
        const int64_t n = (int64_t)(q * d) + k;
        if (n >= 0 && n <= 0xFFFFFFFFll && !check((uint32_t)n)) return 1;
      }
    const uint64_t qmax = 0xFFFFFFFFull / d;
    for (uint64_t q = qmax > 40 ? qmax - 40 : 0; q <= qmax; ++q)
      for (int k = -1; k <= 1; ++k) {
        const int64_t n = (int64_t)(q * d) + k;
        if (n >= 0 && n <= 0xFFFFFFFFll && !check((uint32_t)n)) return 1;
      }
    uint32_t x = 0x9E3779B9u * (d + 1);
    for (int i = 0; i < 2000; ++i) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      if (!check(x)) return 1;
    }
    if (!check(0) || !check(0xFFFFFFFFu)) return 1;
  }
  printf("ok %llu\n", checked);
  return 0;
}
