// mfma_scan.hip - can the matrix pipe run World::hit's discriminant scan?
//
// The sign of the reference's discriminant (src/main.rs:131-135), divided by |d|^2 and halved, is a
// bilinear form between ten numbers of the sphere and ten of the ray:
//     disc / (2 |d|^2) = 1/2 (c.dn)^2 + c.op - k/2 - |op|^2 / 2,
//     dn = d / |d|,  op = o - (o.dn) dn  (the part of the origin across the ray),  k = |c|^2 - r^2
//   sphere: cx^2 cy^2 cz^2 cxcy cxcz cycz cx cy cz k
//   ray:    dx^2/2 dy^2/2 dz^2/2 dxdy dxdz dydz opx opy opz -1/2        threshold |op|^2 / 2
// so 32 spheres x 32 rays are one 32x32 tile of a matrix product with K = 10.  In f16 with every
// number split in two halves (hi + lo, 22 bits together) and the three products hi.hi, hi.lo,
// lo.hi per term, K = 30: two v_mfma_f32_32x32x16_f16 per tile, then ONE compare per
// (sphere, ray) on the vector ALU instead of 7 fused multiply-adds and a compare.
//
// This program measures, on real geometry (the benchmark scene's sphere sizes and camera):
//   1. the arithmetic of the instruction: error of a K = 32 f16 dot product with f32 accumulation
//      against the exact sum, in units of 2^-24 of the sum of the products' magnitudes, with
//      normal and with subnormal f16 inputs;
//   2. that the filter never drops a sphere whose exact discriminant is >= 0, and how many it
//      passes compared with the exact count (false positives cost an exact replay each);
//   3. tests per second of the loop as the kernel would run it (sphere rows from LDS, ray
//      columns in registers, candidates appended to a list in LDS).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mfma_scan.hip -o mfma_scan_bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kSlots = 32;       // K of a tile: two MFMAs of 16
constexpr int kTerms = 10;

// ---- host: features and their f16 halves ---------------------------------------------------
static _Float16 to_h(double v) { return (_Float16)v; }   // round to nearest even

struct SphereRow { _Float16 a[kSlots]; };
struct RayCol { _Float16 b[kSlots]; float thr; };

static void split2(double v, _Float16* hi, _Float16* lo) {
  *hi = to_h(v);
  *lo = to_h(v - (double)*hi);
}

// `scale` (a power of two) brings the scene into the range of f16: features are formed from
// scale * c, scale * o.
static SphereRow sphere_row(const double c[3], double r, double scale, double rel_slack, double abs_slack) {
  const double x = c[0] * scale, y = c[1] * scale, z = c[2] * scale, rr = r * scale;
  const double c2 = x * x + y * y + z * z;
  const double k = c2 - rr * rr - rel_slack * (c2 + rr * rr) - abs_slack;
  const double S[9] = {x * x, y * y, z * z, x * y, x * z, y * z, x, y, z};
  SphereRow row;
  memset(&row, 0, sizeof row);
  for (int t = 0; t < 9; ++t) {
    _Float16 hi, lo;
    split2(S[t], &hi, &lo);
    row.a[3 * t] = hi; row.a[3 * t + 1] = hi; row.a[3 * t + 2] = lo;
  }
  // k in three parts, rounded so that the sum does not exceed k
  _Float16 kh = to_h(k), km = to_h(k - (double)kh), kl = to_h(k - (double)kh - (double)km);
  row.a[27] = kh; row.a[28] = km; row.a[29] = kl;
  row.a[30] = (_Float16)1.0f; row.a[31] = (_Float16)1.0f;   // times the two halves of -threshold
  return row;
}

// The ray's side is what the kernel would compute, so it is formed in f32 here.
static float g_thr_bias = 0.0f;   // argv[2]: added to every threshold (3e4: nothing passes, the loop's fast path alone)
static RayCol ray_col(const float o_in[3], const float d[3], float scale, float rel_slack) {
  const float o[3] = {o_in[0] * scale, o_in[1] * scale, o_in[2] * scale};
  const float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const float s = 1.0f / std::sqrt(a);
  const float n[3] = {d[0] * s, d[1] * s, d[2] * s};
  const float od = o[0] * n[0] + o[1] * n[1] + o[2] * n[2];
  const float p[3] = {o[0] - od * n[0], o[1] - od * n[1], o[2] - od * n[2]};
  const float F[9] = {0.5f * n[0] * n[0], 0.5f * n[1] * n[1], 0.5f * n[2] * n[2],
                      n[0] * n[1], n[0] * n[2], n[1] * n[2], p[0], p[1], p[2]};
  RayCol col;
  memset(&col, 0, sizeof col);
  for (int t = 0; t < 9; ++t) {
    const _Float16 hi = (_Float16)F[t];
    const _Float16 lo = (_Float16)(F[t] - (float)hi);
    col.b[3 * t] = hi; col.b[3 * t + 1] = lo; col.b[3 * t + 2] = hi;
  }
  col.b[27] = col.b[28] = col.b[29] = (_Float16)(-0.5f);
  const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
  const float pp = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  col.thr = 0.5f * pp - rel_slack * oo + g_thr_bias;
  const _Float16 th = (_Float16)(-col.thr);
  col.b[30] = th;
  col.b[31] = (_Float16)(-col.thr - (float)th);
  return col;
}

// ---- 1. arithmetic of the instruction --------------------------------------------------------
// One wave, one tile: D = A (32 x 32) * B (32 x 32), A row-major [m][k], B stored [n][k].
__global__ void k_tile(const _Float16* A, const _Float16* B, float* D, int tiles) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const _Float16* a = A + (size_t)t * 32 * kSlots + r * kSlots;
    const _Float16* b = B + (size_t)t * 32 * kSlots + r * kSlots;
    h8 a0, a1, b0, b1;
    for (int j = 0; j < 8; ++j) {
      a0[j] = a[8 * h + j]; a1[j] = a[16 + 8 * h + j];
      b0[j] = b[8 * h + j]; b1[j] = b[16 + 8 * h + j];
    }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
      D[(size_t)t * 1024 + m * 32 + r] = acc[i];
    }
  }
}

// ---- 3. the loop ---------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_f16: 16 spheres x 16 rays x all 32 slots in one instruction (16 cycles), four
// results per lane.  The threshold rides in the last two slots (sphere side 1, 1; ray side the
// two halves of -threshold), so a candidate is simply a result that is not negative, whichever
// ray it belongs to: the 32 results a lane holds for 16 spheres x 128 rays are folded with
// v_max3_i32 (a non-negative float is a non-negative integer) and ONE compare and branch decide
// whether the block has any candidate at all.  Only then does a lane collect the sign bits of
// its 32 results in a mask and append an entry per clear bit to the wave's list.
// (A first version on 32x32x16 tiles with one compare per result and a branch ladder over the
// 16 compare masks ran 6.3 Ttests/s: the ladder, taken for most tiles, cost more than the MFMAs.)
// LDS: sphere rows as four quarters of 8 slots, [quarter][sphere][8 f16]: lane l reads quarter
// l >> 4 of sphere l & 15 of the block, 16 contiguous bytes, neighbours contiguous.
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kListEntries = 512;
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_scan(const _Float16* rows /* [4][n_pad][8] */, int n_pad,
                                                    const _Float16* cols /* [rays][32] */, const float* thrs,
                                                    int iters, unsigned long long* found, uint32_t* lists_out) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  h8* sph = reinterpret_cast<h8*>(lds_raw);                    // [4][n_pad]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  uint32_t* list = reinterpret_cast<uint32_t*>(sph + 4 * n_pad) + wave * kListEntries;
  const h8* rows8 = reinterpret_cast<const h8*>(rows);
  for (int i = threadIdx.x; i < 4 * n_pad; i += blockDim.x) sph[i] = rows8[i];
  __syncthreads();
  // this wave's 128 rays: block j holds rays 16 j .. 16 j + 15 of the wave
  const size_t ray0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 128;
  h8 b[8];
  const h8* cols8 = reinterpret_cast<const h8*>(cols);
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = cols8[(ray0 + 16 * j + c) * 4 + g];
  uint32_t count = 0;
  unsigned long long total = 0;
  const f4v zero = {};
  for (int it = 0; it < iters; ++it) {
    h8 a = sph[g * n_pad + c];
    for (int blk = 0; blk < n_pad; blk += 16) {
      f4v acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[j], zero, 0, 0, 0);
      if (blk + 16 < n_pad) a = sph[g * n_pad + blk + 16 + c];   // read ahead
      // bit q = 8 i + j of a lane's mask stands for result i of ray block j, so that an entry is
      // the lane's base entry plus q << 4:  sphere (blk + 4 g + i) << 7 | ray 16 j + c
      int v[32];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[8 * i + j] = __float_as_int(acc[j][i]);
      int m = v[0];
#pragma unroll
      for (int k = 1; k + 1 < 32; k += 2) m = max(max(m, v[k]), v[k + 1]);
      m = max(m, v[31]);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) {
        uint32_t neg = 0;   // bit q = sign of v[q]
#pragma unroll
        for (int q = 31; q >= 0; --q) neg = __builtin_amdgcn_alignbit(neg, (uint32_t)v[q], 31);
        uint32_t cand = ~neg;
        // every lane's entries behind those of the lanes below it: prefix sum of the bit counts
        const uint32_t p = (uint32_t)__builtin_popcount(cand);
        uint32_t inc = p;
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);   // row_shr:1
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);   // row_shr:2
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);   // row_shr:4
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true);   // row_shr:8
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);  // row_bcast:15
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);  // row_bcast:31
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (count + all > kListEntries) { total += count; count = 0; }   // (the kernel replays here)
        uint32_t at = count + inc - p;
        const uint32_t base = ((uint32_t)(blk + 4 * g) << 7) | (uint32_t)c;
        while (cand != 0) {
          const uint32_t q = (uint32_t)__builtin_ctz(cand);
          cand &= cand - 1;
          if (at < kListEntries) list[at] = base + (q << 4);
          ++at;
        }
        count += all < kListEntries ? all : kListEntries;
      }
    }
    // the first pass of the first wave leaves its list behind for the host's check
    if (it == 0 && blockIdx.x == 0 && wave == 0) {
      __builtin_amdgcn_wave_barrier();
      for (uint32_t e = lane; e < count; e += 64) lists_out[1 + e] = list[e];
      if (lane == 0) lists_out[0] = count;
    }
    total += count;
    count = 0;
  }
  if (lane == 0) atomicAdd(found, total);
}


// ---- the instruction's rate on its own ------------------------------------------------------------
// Eight independent accumulation chains written as inline assembly (the compiler's own code for
// such a loop shuffles accumulators through v_accvgpr moves).  MODE 0: 16x16x32, 1: 32x32x16.
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, int seed) {
  const int lane = threadIdx.x & 63;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.001f * (float)((lane + j + seed) & 15));
    b[j] = (_Float16)(0.002f * (float)((lane * 3 + j + seed) & 7));
  }
  float s = 0.0f;
  if (MODE == 0) {
    f4v c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    for (int it = 0; it < iters; ++it) {
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n"
          "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n"
          "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n"
          "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
          : "v"(a), "v"(b));
    }
    s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  } else {
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
      asm volatile(
          "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
          "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
          : "v"(a), "v"(b));
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- do vector and matrix instructions share a SIMD's issue? ----------------------------------------
// Per trip 8 independent MFMAs and/or 32 independent FMAs.  WHAT 0: the MFMAs only; 1: the FMAs
// only; 2: both, interleaved, in every wave; 3: waves 0 and 2 of a workgroup (one per SIMD pair)
// run the MFMAs, waves 1 and 3 the FMAs.  If the two kinds overlapped, 2 and 3 would take as long
// as the slower of 0 and 1; if they share the issue, as long as both together.
template <int WHAT>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, int seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.001f * (float)((lane + j + seed) & 15));
    b[j] = (_Float16)(0.002f * (float)((lane * 3 + j + seed) & 7));
  }
  f4v c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = 0.001f * (float)(lane + j + seed);
  const float m = 1.0001f, k = 0.0001f;
  const bool do_m = WHAT == 0 || WHAT == 2 || (WHAT == 3 && (wave & 1) == 0);
  const bool do_v = WHAT == 1 || WHAT == 2 || (WHAT == 3 && (wave & 1) == 1);
  const int trips = WHAT == 3 ? 2 * iters : iters;   // each kind of wave does both waves' share
  for (int it = 0; it < trips; ++it) {
    if (do_m && do_v) {
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %16, %17, %0\n v_fma_f32 %8, %8, %18, %19\n v_fma_f32 %9, %9, %18, %19\n v_fma_f32 %10, %10, %18, %19\n v_fma_f32 %11, %11, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %1, %16, %17, %1\n v_fma_f32 %12, %12, %18, %19\n v_fma_f32 %13, %13, %18, %19\n v_fma_f32 %14, %14, %18, %19\n v_fma_f32 %15, %15, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %2, %16, %17, %2\n v_fma_f32 %8, %8, %18, %19\n v_fma_f32 %9, %9, %18, %19\n v_fma_f32 %10, %10, %18, %19\n v_fma_f32 %11, %11, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %3, %16, %17, %3\n v_fma_f32 %12, %12, %18, %19\n v_fma_f32 %13, %13, %18, %19\n v_fma_f32 %14, %14, %18, %19\n v_fma_f32 %15, %15, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %4, %16, %17, %4\n v_fma_f32 %8, %8, %18, %19\n v_fma_f32 %9, %9, %18, %19\n v_fma_f32 %10, %10, %18, %19\n v_fma_f32 %11, %11, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %5, %16, %17, %5\n v_fma_f32 %12, %12, %18, %19\n v_fma_f32 %13, %13, %18, %19\n v_fma_f32 %14, %14, %18, %19\n v_fma_f32 %15, %15, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %6, %16, %17, %6\n v_fma_f32 %8, %8, %18, %19\n v_fma_f32 %9, %9, %18, %19\n v_fma_f32 %10, %10, %18, %19\n v_fma_f32 %11, %11, %18, %19\n"
          "v_mfma_f32_16x16x32_f16 %7, %16, %17, %7\n v_fma_f32 %12, %12, %18, %19\n v_fma_f32 %13, %13, %18, %19\n v_fma_f32 %14, %14, %18, %19\n v_fma_f32 %15, %15, %18, %19\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7),
            "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
          : "v"(a), "v"(b), "v"(m), "v"(k));
    } else if (do_m) {
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n"
          "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n"
          "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n"
          "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
          : "v"(a), "v"(b));
    } else if (do_v) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
            : "v"(m), "v"(k));
    }
  }
  float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  for (int j = 0; j < 8; ++j) s += f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WHAT>
static float time_mix(int cus, int wgs_per_cu, float* d_out, int n_it) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(cus * wgs_per_cu);
  hipLaunchKernelGGL(k_mix<WHAT>, grid, dim3(256), 0, 0, d_out, 10, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_mix<WHAT>, grid, dim3(256), 0, 0, d_out, n_it, 1);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 40;
  if (argc > 2) g_thr_bias = (float)atof(argv[2]);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device CUs %d  %s\n", prop.multiProcessorCount, prop.gcnArchName);
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);

  // ---- 1. arithmetic ----
  {
    const int tiles = 256;
    std::vector<_Float16> A((size_t)tiles * 32 * kSlots), B(A.size());
    // tile classes: 0 = normal values over 2^-6..2^6 with random signs; 1 = same, with every third
    // entry a subnormal-sized "lo" part (2^-11 of its neighbour, below 2^-14); 2 = pure subnormal
    // times large
    for (int t = 0; t < tiles; ++t) {
      const int cls = t % 3;
      for (int i = 0; i < 32 * kSlots; ++i) {
        double va = std::ldexp(1.0 + U(rng), (int)(U(rng) * 12) - 6) * (U(rng) < 0.5 ? -1 : 1);
        double vb = std::ldexp(1.0 + U(rng), (int)(U(rng) * 12) - 6) * (U(rng) < 0.5 ? -1 : 1);
        if (cls == 1 && i % 3 == 2) va = std::ldexp(va, -17);
        if (cls == 1 && i % 3 == 1) vb = std::ldexp(vb, -17);
        if (cls == 2) { va = std::ldexp(1.0 + U(rng), -20 + (int)(U(rng) * 5)); vb = std::ldexp(1.0 + U(rng), 8); }
        A[(size_t)t * 32 * kSlots + i] = to_h(va);
        B[(size_t)t * 32 * kSlots + i] = to_h(vb);
      }
    }
    _Float16 *dA, *dB;
    float* dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, (size_t)tiles * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_tile, dim3(64), dim3(64), 0, 0, dA, dB, dD, tiles);
    CK(hipDeviceSynchronize());
    std::vector<float> D((size_t)tiles * 1024);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double worst[3] = {0, 0, 0}, worst_rel_result[3] = {0, 0, 0};
    for (int t = 0; t < tiles; ++t)
      for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
          double exact = 0, mag = 0;
          for (int k = 0; k < kSlots; ++k) {
            const double p = (double)A[(size_t)t * 32 * kSlots + m * kSlots + k] *
                             (double)B[(size_t)t * 32 * kSlots + n * kSlots + k];
            exact += p; mag += std::fabs(p);
          }
          const double err = std::fabs((double)D[(size_t)t * 1024 + m * 32 + n] - exact);
          const double e = err / mag * 16777216.0;
          if (e > worst[t % 3]) worst[t % 3] = e;
          const double er = err / std::fmax(std::fabs(exact), 1e-300) * 16777216.0;
          if (er > worst_rel_result[t % 3]) worst_rel_result[t % 3] = er;
        }
    printf("K=32 f16 dot, f32 accumulate: worst |error| / sum|products| in units of 2^-24:\n"
           "  normal inputs %.3f   with subnormal-sized low parts %.3f   subnormal x large %.3f\n",
           worst[0], worst[1], worst[2]);
    printf("  (relative to the result itself: %.3f / %.3f / %.3f; a flushed subnormal operand would show as 2^24 = 1.7e7 in the third)\n",
           worst_rel_result[0], worst_rel_result[1], worst_rel_result[2]);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dD));
  }


  // ---- vector and matrix issue, alone and together ----
  if (argc > 1 && !strcmp(argv[1], "mix")) {
    float* d_out;
    CK(hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
      const int n_it = 4000;
      const float t0 = time_mix<0>(prop.multiProcessorCount, wgs_per_cu, d_out, n_it);
      const float t1 = time_mix<1>(prop.multiProcessorCount, wgs_per_cu, d_out, n_it);
      const float t2 = time_mix<2>(prop.multiProcessorCount, wgs_per_cu, d_out, n_it);
      const float t3 = time_mix<3>(prop.multiProcessorCount, wgs_per_cu, d_out, n_it);
      printf("%d wave(s)/SIMD, per trip 8 MFMA 16x16x32 f16 / 32 FMA f32: MFMA alone %.3f ms, FMA alone %.3f ms, "
             "both in every wave %.3f ms, in alternate waves %.3f ms (sum %.3f, larger %.3f)\n",
             wgs_per_cu, t0, t1, t2, t3, t0 + t1, t0 > t1 ? t0 : t1);
    }
    CK(hipFree(d_out));
    return 0;
  }

  // ---- the instruction's rate ----
  {
    float* d_out;
    CK(hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    for (int mode = 0; mode < 2; ++mode)
      for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {
        const int n_it = 4000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const dim3 grid(prop.multiProcessorCount * wgs_per_cu);
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, grid, dim3(256), 0, 0, d_out, 10, 1);
        else hipLaunchKernelGGL(k_rate<1>, grid, dim3(256), 0, 0, d_out, 10, 1);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, grid, dim3(256), 0, 0, d_out, n_it, 1);
        else hipLaunchKernelGGL(k_rate<1>, grid, dim3(256), 0, 0, d_out, n_it, 1);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s, %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD at 2.4 GHz\n",
               mode == 0 ? "v_mfma_f32_16x16x32_f16, eight chains" : "v_mfma_f32_32x32x16_f16, four chains ", wgs_per_cu,
               ms * 1e-3 * 2.4e9 / ((double)n_it * (mode == 0 ? 8 : 4) * wgs_per_cu));
      }
    CK(hipFree(d_out));
  }

  // ---- 2 + 3. geometry ----
  const int n_sph = 488, n_pad = 512;
  const double scale = 8.0;                       // |c| <= 15.6 -> 125
  const double rel_slack = std::ldexp(1.0, -18), abs_slack = std::ldexp(1.0, -20);
  std::vector<double> cs(3 * n_sph), rs(n_sph);
  for (int i = 0; i < n_sph; ++i) {
    // the benchmark scene's population: small spheres on a 22 x 22 grid at y = 0.2, three of radius 1
    if (i < 484) {
      const int a = i / 22 - 11, b = i % 22 - 11;
      cs[3 * i] = a + 0.9 * U(rng); cs[3 * i + 1] = 0.2; cs[3 * i + 2] = b + 0.9 * U(rng);
      rs[i] = 0.2;
    } else {
      cs[3 * i] = (i - 485) * 4.0; cs[3 * i + 1] = 1.0; cs[3 * i + 2] = 0.0;
      rs[i] = 1.0;
    }
  }
  std::vector<_Float16> rows((size_t)4 * n_pad * 8, (_Float16)0);
  for (int i = 0; i < n_pad; ++i) {
    SphereRow row;
    if (i < n_sph) row = sphere_row(&cs[3 * i], rs[i], scale, rel_slack, abs_slack);
    else {   // padding: a row no ray can pass (k huge)
      memset(&row, 0, sizeof row);
      row.a[27] = (_Float16)60000.0f;
    }
    for (int q = 0; q < 4; ++q)
      for (int j = 0; j < 8; ++j) rows[((size_t)q * n_pad + i) * 8 + j] = row.a[8 * q + j];
  }
  const int wgs = prop.multiProcessorCount * 2;   // 256 threads, 2 per CU at 4 waves/SIMD; more are queued
  const int grid = wgs * 2;
  const size_t n_rays = (size_t)grid * 4 * 128;
  std::vector<_Float16> cols(n_rays * 32);
  std::vector<float> thrs(n_rays);
  std::vector<float> ro(3 * n_rays), rd(3 * n_rays);
  for (size_t i = 0; i < n_rays; ++i) {
    float o[3], d[3];
    // a wave's 128 rays: 64 camera rays through one pixel's footprint (a wave works on the samples
    // of one or two pixels), 64 secondary rays from points on the ground in random directions
    static double tx = 0, tz = 0;
    if (i % 128 == 0) { tx = -11 + 22 * U(rng); tz = -11 + 22 * U(rng); }
    if (i % 128 < 64) {
      o[0] = 13; o[1] = 2; o[2] = 3;
      d[0] = (float)(tx + 0.02 * U(rng) - 13); d[1] = (float)(0.2 - 2 + 0.02 * U(rng)); d[2] = (float)(tz + 0.02 * U(rng) - 3);
    } else {
      o[0] = (float)(-11 + 22 * U(rng)); o[1] = (float)(1e-3 + 0.4 * U(rng)); o[2] = (float)(-11 + 22 * U(rng));
      d[0] = (float)(2 * U(rng) - 1); d[1] = (float)(U(rng) * 0.6 - 0.1); d[2] = (float)(2 * U(rng) - 1);
    }
    const RayCol col = ray_col(o, d, (float)scale, (float)rel_slack);
    memcpy(&cols[i * 32], col.b, 64);
    thrs[i] = col.thr;
    memcpy(&ro[3 * i], o, 12); memcpy(&rd[3 * i], d, 12);
  }
  _Float16 *d_rows, *d_cols;
  float* d_thr;
  unsigned long long* d_found;
  uint32_t* d_lists;
  CK(hipMalloc(&d_rows, rows.size() * 2)); CK(hipMalloc(&d_cols, cols.size() * 2));
  CK(hipMalloc(&d_thr, thrs.size() * 4)); CK(hipMalloc(&d_found, 8)); CK(hipMalloc(&d_lists, 4 * (1 + 128 * 512)));
  CK(hipMemcpy(d_rows, rows.data(), rows.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cols, cols.data(), cols.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_thr, thrs.data(), thrs.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)4 * n_pad * 16 + 4 * kListEntries * 4;

  auto run = [&](auto kernel, const char* name, int it, size_t lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d_found, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, d_rows, n_pad, d_cols, d_thr, 2, d_found, d_lists);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_found, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, d_rows, n_pad, d_cols, d_thr, it, d_found, d_lists);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long found = 0;
    CK(hipMemcpy(&found, d_found, 8, hipMemcpyDeviceToHost));
    const double tests = (double)n_rays * n_pad * it;
    const double mfmas = tests / 256.0;
    printf("%s: %.3f ms  %.2f Ttests/s (%.2f counting the %d real spheres)  %.1f cycles per MFMA per SIMD at 2.4 GHz  "
           "candidates per ray %.3f\n",
           name, ms, tests / ms * 1e-9, tests / ms * 1e-9 * n_sph / n_pad, n_sph,
           ms * 1e-3 * 2.4e9 * prop.multiProcessorCount * 4 / mfmas, (double)found / ((double)n_rays * it));
  };
  // a workgroup is one wave per SIMD; the LDS it asks for sets how many share a CU
  run(k_scan<4>, "scan on MFMA, 4 waves/SIMD", iters, lds);
  run(k_scan<2>, "scan on MFMA, 2 waves/SIMD", iters, 80 * 1024);
  run(k_scan<1>, "scan on MFMA, 1 wave/SIMD ", iters, 160 * 1024);

  // ---- 2. no false negatives: the first wave's 128 rays against all spheres, exactly ----
  {
    std::vector<uint32_t> lists(1 + 128 * 512);
    CK(hipMemcpy(lists.data(), d_lists, lists.size() * 4, hipMemcpyDeviceToHost));
    std::vector<char> flagged((size_t)128 * n_pad, 0);
    for (uint32_t e = 0; e < lists[0]; ++e) flagged[(size_t)(lists[1 + e] & 127) * n_pad + (lists[1 + e] >> 7)] = 1;
    long exact_hits = 0, missed = 0, listed = lists[0], f32_listed = 0;
    for (int ray = 0; ray < 128; ++ray) {
      const float* o = &ro[3 * ray];
      const float* d = &rd[3 * ray];
      for (int s = 0; s < n_sph; ++s) {
        // the discriminant as the reference computes it (f32, src/main.rs:131-135) and exactly
        const float ocx = o[0] - (float)cs[3 * s], ocy = o[1] - (float)cs[3 * s + 1], ocz = o[2] - (float)cs[3 * s + 2];
        const float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float hb = ocx * d[0] + ocy * d[1] + ocz * d[2];
        const float c = ocx * ocx + ocy * ocy + ocz * ocz - (float)(rs[s] * rs[s]);
        const float disc = hb * hb - a * c;
        if (disc >= 0.0f) {
          ++exact_hits;
          if (!flagged[(size_t)ray * n_pad + s]) ++missed;
        }
        // what the present f32 scan would list: slack 2^-17 (|o|^2 + |c|^2)
        const double c2 = cs[3 * s] * cs[3 * s] + cs[3 * s + 1] * cs[3 * s + 1] + cs[3 * s + 2] * cs[3 * s + 2];
        const double oo = (double)o[0] * o[0] + (double)o[1] * o[1] + (double)o[2] * o[2];
        if ((double)disc / a >= -std::ldexp(c2 + oo, -17)) ++f32_listed;
      }
    }
    printf("first wave, 128 rays x %d spheres: reference discriminant >= 0 for %ld pairs, the MFMA filter listed %ld, "
           "missed %ld (must be 0); the f32 scan's slack would list about %ld\n",
           n_sph, exact_hits, listed, missed, f32_listed);
  }
  return 0;
}
