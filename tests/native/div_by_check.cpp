// Checks rt::div_by (ray_tracing_amd/csrc/rt_internal.h) on the CPU against the machine's own
// division: the quotients of the sample loop, (pixel + jitter) / (image extent as f32), for image
// extents from 1 to 2^31 - 1 - every float between 2^-24 and the extent for two of them, a
// pseudo-random sweep for the rest - and mantissa pairs at large.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../ray_tracing_amd/csrc/rt_internal.h"

static uint64_t state = 88172645463325252ull;
static inline uint64_t next() {
  state ^= state << 13; state ^= state >> 7; state ^= state << 17;
  return state;
}
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  unsigned long long checked = 0;
  auto check = [&](float a, float b) {
    ++checked;
    const float want = a / b, got = rt::div_by(a, b, 1.0f / b);
    if (bits_of(want) != bits_of(got)) {
      printf("FAIL %a / %a: got %a want %a\n", a, b, got, want);
      return false;
    }
    return true;
  };
  const uint32_t extents[] = {1, 2, 3, 7, 27, 48, 225, 400, 1080, 1920, 2160, 3840, 7680, 65535, 65536,
                              1000003, 16777215u, 16777217u, 0x7FFFFFFFu};
  for (uint32_t e : extents) {
    const float b = (float)e;
    if (!check(0.0f, b)) return 1;
    const uint32_t lo = bits_of(0x1p-24f), hi = bits_of(b);
    if (e == 27 || e == 1920) {
      for (uint32_t u = lo; u <= hi; ++u)
        if (!check(from_bits(u), b)) return 1;
    } else {
      for (int k = 0; k < 2000000; ++k)
        if (!check(from_bits(lo + (uint32_t)(next() % (hi - lo + 1u))), b)) return 1;
      for (uint32_t u = hi - 4096; u <= hi; ++u)
        if (!check(from_bits(u), b)) return 1;
    }
  }
  // any mantissas, with many divisors near the top of the binade (where a * y is furthest off)
  for (int k = 0; k < 40000000; ++k) {
    const uint64_t r = next();
    uint32_t ub = 0x3F800000u | (uint32_t)((r >> 23) & 0x7FFFFFu);
    if ((k & 7) == 0) ub |= 0x7FFF00u;
    if (!check(from_bits(0x3F800000u | (uint32_t)(r & 0x7FFFFFu)), from_bits(ub))) return 1;
  }
  printf("ok %llu quotients\n", checked);
  return 0;
}
