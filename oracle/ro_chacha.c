/* ro_chacha.c - an independent random stream for the oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference draws every random number from rand 0.8.5's thread_rng (Cargo.lock:616-618):
 * ChaCha with 12 rounds (rand_chacha 0.3.1, ChaCha12Core behind a BlockRng; the 64 KiB
 * reseeding of ReseedingRng does not change the distribution), seeded per thread from the
 * operating system, and `rand::random::<f32>()` maps one 32-bit output to
 * (word >> 8) * 2^-24 (call sites src/cpu/job.rs:50-51, src/main.rs:215, 267-269).  This file
 * restates that generator from the published ChaCha definition so that the oracle's estimator
 * can be driven by a stream that shares nothing with include/rt_rng.h, the counter stream the
 * GPU kernel uses.  A render driven by it is the closest obtainable stand-in for a render of
 * the reference program: same generator family, same float mapping, same order of consumption;
 * only the seed differs, as it does between any two runs of the reference.
 *
 * State words: "expand 32-byte k", 8 key words, a 64-bit block counter in words 12-13 and a
 * 64-bit stream id in words 14-15 (rand_chacha's layout); output words are handed out in
 * order, block after block.
 */
#include "rt_oracle.h"

#include <string.h>

static inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

#define RO_QR(a, b, c, d)                 \
  do {                                    \
    a += b; d ^= a; d = rotl32(d, 16);    \
    c += d; b ^= c; b = rotl32(b, 12);    \
    a += b; d ^= a; d = rotl32(d, 8);     \
    c += d; b ^= c; b = rotl32(b, 7);     \
  } while (0)

void ro_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                     uint32_t out[16]) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
  for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
  s[12] = (uint32_t)counter;
  s[13] = (uint32_t)(counter >> 32);
  s[14] = (uint32_t)stream;
  s[15] = (uint32_t)(stream >> 32);
  memcpy(x, s, sizeof x);
  for (int r = 0; r < rounds; r += 2) {
    RO_QR(x[0], x[4], x[8], x[12]);
    RO_QR(x[1], x[5], x[9], x[13]);
    RO_QR(x[2], x[6], x[10], x[14]);
    RO_QR(x[3], x[7], x[11], x[15]);
    RO_QR(x[0], x[5], x[10], x[15]);
    RO_QR(x[1], x[6], x[11], x[12]);
    RO_QR(x[2], x[7], x[8], x[13]);
    RO_QR(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

void ro_chacha12_seed(ro_chacha* c, const uint32_t key[8], uint64_t stream) {
  memcpy(c->key, key, sizeof c->key);
  c->counter = 0;
  c->stream = stream;
  c->index = 16; /* buffer empty */
}

uint32_t ro_chacha12_next_u32(ro_chacha* c) {
  if (c->index >= 16) {
    ro_chacha_block(c->key, c->counter, c->stream, 12, c->buffer);
    c->counter++;
    c->index = 0;
  }
  return c->buffer[c->index++];
}

/* rand 0.8.5, `Standard` for f32: the top 24 bits of one u32, times 2^-24 */
float ro_chacha12_next_f32(ro_chacha* c) {
  return (float)(ro_chacha12_next_u32(c) >> 8) * (1.0f / 16777216.0f);
}

/* A key for (seed, a, b): the seed words pushed through one ChaCha20 block, so that nearby
 * (seed, a, b) give unrelated keys.  (The reference seeds each thread from the OS.) */
void ro_chacha12_seed_from(ro_chacha* c, uint32_t seed, uint32_t a, uint32_t b) {
  uint32_t k0[8] = {seed, a, b, 0x243F6A88u, 0x85A308D3u, 0x13198A2Eu, 0x03707344u, 0xA4093822u};
  uint32_t out[16];
  ro_chacha_block(k0, 0, 0, 20, out);
  ro_chacha12_seed(c, out, 0);
}
