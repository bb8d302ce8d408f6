/* Keyed random stream shared by the HIP kernels, the C++ host mirror and the CPU oracle (called
 * the counter stream elsewhere: every sample's draws are a function of counters, not of history).
 *
 * The reference draws every random number from rand::random::<f32>() (src/cpu/job.rs:50-51,
 * src/main.rs:215,267-269): a thread-local, OS-seeded ChaCha12 stream mapped to
 * {k * 2^-24 : 0 <= k < 2^24}.  That stream is unseeded and therefore not reproducible, so this
 * build keeps the DISTRIBUTION (uniform 24-bit floats in [0,1)) and the ORDER in which draws are
 * consumed, and replaces the generator by a stream that is a pure function of
 * (seed, pixel index, sample index, draw number).  A render is then bit-reproducible for any
 * tiling, any lane assignment and any number of GPUs.
 *
 * stream(seed, pixel, sample):   state0 = mix(mix(mix(seed ^ A) ^ pixel) + sample)   (never 0)
 *                                inc    = (state0 rotated right by 11 bits) ^ B | 1
 * draw number c (0-based):       bits = x(c + 1) + inc, with x(0) = state0 and x(k + 1) = the
 *                                xorshift32 step (13, 17, 5) of x(k);  xi = (bits >> 8) * 2^-24
 * mix() is the 32-bit finaliser "lowbias32" (two multiplies, three xor-shifts).  It seeds every
 * sample's stream - one hash per pixel and one per sample; the draws inside a sample come from
 * the shift-and-xor recurrence, because a 32-bit integer multiply runs at a quarter of the vector
 * ALU's rate on the GPU and a path makes four draws per segment (one hash per draw, and four
 * per sample: about 2 % of the benchmark frame's time).  Marsaglia's xorshift32 has the full
 * period 2^32 - 1 over non-zero states; `inc` keeps the streams of two samples that meet on that
 * cycle from handing out the same values (their draws differ by a constant modulo 1).
 *
 * Scene construction (the reference's create_random_world draws from the same rand::random)
 * uses the counter form: state0 as above, inc = mix(mix(mix(seed + B) + pixel) ^ sample) | 1,
 * bits = mix(state0 + (c + 1) * inc): rt_rng_counter_init, rt_rng_counter_next_*.
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

typedef struct rt_rng {
  uint32_t state;
  uint32_t inc; /* always odd */
} rt_rng;

RT_HD uint32_t rt_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

/* Per-render constant: computed once, not per sample. */
RT_HD uint32_t rt_rng_seed_a(uint32_t seed) { return rt_mix32(seed ^ 0xA511E9B3U); }
/* Per-pixel constant: computed once per pixel (or per work item). */
RT_HD uint32_t rt_rng_pixel(uint32_t seed_a, uint32_t pixel) { return rt_mix32(seed_a ^ pixel); }

RT_HD rt_rng rt_rng_from_pixel(uint32_t pixel_hash, uint32_t sample) {
  rt_rng r;
  r.state = rt_mix32(pixel_hash + sample);
  if (r.state == 0U) r.state = 0x9E3779B9U; /* the one state the shift-and-xor recurrence cannot leave */
  r.inc = (((r.state >> 11) | (r.state << 21)) ^ 0x632BE5ABU) | 1U;
  return r;
}

RT_HD rt_rng rt_rng_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
  return rt_rng_from_pixel(rt_rng_pixel(rt_rng_seed_a(seed), pixel), sample);
}

RT_HD uint32_t rt_rng_next_u32(rt_rng* r) {
  uint32_t x = r->state;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  r->state = x;
  return x + r->inc;
}

/* Same mapping as rand 0.8.5's Standard distribution for f32: 24 random bits * 2^-24. */
RT_HD float rt_rng_next_f32(rt_rng* r) {
  return (float)(rt_rng_next_u32(r) >> 8) * (1.0f / 16777216.0f);
}

/* The counter form, for scene construction. */
RT_HD rt_rng rt_rng_counter_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
  rt_rng r;
  r.state = rt_mix32(rt_mix32(rt_rng_seed_a(seed) ^ pixel) + sample);
  r.inc = rt_mix32(rt_mix32(rt_mix32(seed + 0x632BE5ABU) + pixel) ^ sample) | 1U;
  return r;
}
RT_HD uint32_t rt_rng_counter_next_u32(rt_rng* r) {
  r->state += r->inc;
  return rt_mix32(r->state);
}
RT_HD float rt_rng_counter_next_f32(rt_rng* r) {
  return (float)(rt_rng_counter_next_u32(r) >> 8) * (1.0f / 16777216.0f);
}

/* Pixel index reserved for streams that are not tied to a pixel (scene construction). */
#define RT_RNG_SCENE_STREAM 0xFFFFFFF1U

#endif /* RT_RNG_H */
