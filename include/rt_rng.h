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
 * stream(seed, pixel, sample):   state0 = mix(mix(mix(seed ^ A) ^ pixel) + sample)
 *                                inc    = (state0 rotated right by 11 bits) ^ B | 1
 * draw number c (0-based):       bits = x(c + 1), with x(0) = state0 and
 *                                x(k + 1) = x(k) * 747796405 + inc mod 2^32;  xi = (bits >> 8) * 2^-24
 * mix() is the 32-bit finaliser "lowbias32" (two multiplies, three xor-shifts).  It seeds every
 * sample's stream - one hash per pixel and one per sample.  The draws inside a sample come from
 * a linear congruential step, the cheapest recurrence the GPU's vector ALU has (one integer
 * multiply and one add: a third of the issue time of a shift-and-xor recurrence with three
 * shifts, a sixth of a hash per draw), because the kernels draw in a rejection loop that a whole
 * wave repeats until its unluckiest lane is done: 4 % of the benchmark frame's time lay between
 * this recurrence and the shift-and-xor one.  The multiplier is 1 mod 4 and inc is odd, so every
 * stream has the full period 2^32 (Hull and Dobell); a draw takes the upper 24 bits of the state,
 * whose periods are the longest (bit b of such a state repeats after 2^(b+1) steps: 512 for the
 * lowest bit a draw uses, and a sample draws a few dozen numbers); `inc` keeps two samples
 * whose states meet from handing out the same values from there on.
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
  r.inc = (((r.state >> 11) | (r.state << 21)) ^ 0x632BE5ABU) | 1U;
  return r;
}

RT_HD rt_rng rt_rng_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
  return rt_rng_from_pixel(rt_rng_pixel(rt_rng_seed_a(seed), pixel), sample);
}

RT_HD uint32_t rt_rng_next_u32(rt_rng* r) {
  r->state = r->state * 747796405U + r->inc;
  return r->state;
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
