// clock_probe.hip - is the scan loop held by VALU issue or by the clock?
//
// Every wave stamps the shader-clock counter (clock64 = s_memtime) and the constant 100 MHz
// counter (wall_clock64 = s_memrealtime) around its loop.  From them:
//   actual shader clock while the loop runs  = d(clock64) / d(wall_clock64) * 100 MHz
//   cycles per VALU instruction per SIMD     = launch time * that clock / (waves per SIMD * instructions per wave)
// The second figure does not depend on what the clock does, so it separates "the SIMD issues
// one wave64 fma every N cycles" from "the chip runs slower than its nominal 2.4 GHz under this
// load".  Loops: plain v_fma_f32 chains (operands spread over registers), v_pk_fma_f32 chains,
// and the production scan block (8 fma per ray/sphere test, LDS broadcast read-ahead, sign fold).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

struct Stamp {
  unsigned long long cycles, ticks;
};

__device__ __forceinline__ void stamp_begin(unsigned long long& c, unsigned long long& t) {
  t = wall_clock64();
  c = clock64();
}
__device__ __forceinline__ void stamp_end(Stamp* out, unsigned long long c, unsigned long long t) {
  const unsigned long long c1 = clock64();
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    Stamp s;
    s.cycles = c1 - c;
    s.ticks = t1 - t;
    out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
  }
}

// 16 independent chains, each with its own multiplier and addend register
template <bool LITERAL>
__global__ void k_fma(Stamp* st, float* out, int iters, float seed, float mul) {
  float a[16], m[16], c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + threadIdx.x + i;
    // LITERAL: the multiplier is a compile-time constant (v_fmamk_f32 with a 32-bit literal);
    // otherwise it lives in a register (v_fma_f32 with three VGPR operands, the scan's form)
    m[i] = LITERAL ? 0.999f - 1e-4f * i : mul - 1e-4f * i - 1e-9f * threadIdx.x;
    c[i] = LITERAL ? 0.001f + 1e-4f * i : (1.0f - mul) + 1e-4f * i + 1e-9f * threadIdx.x;
  }
  unsigned long long c0, t0;
  stamp_begin(c0, t0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], m[i], c[i]);
    }
  }
  stamp_end(st, c0, t0);
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// 16 independent v_fma_f32 chains with the registers chosen by hand: accumulators v40, v44, ...
// (all in VGPR bank 0 if banks go by register number mod 4), multiplier and addend in registers of
// the banks given by MB and CB.  MB = CB = 0: all three source operands of every instruction sit in
// one bank; MB = 1, CB = 2: in three different banks.
// FMA3: v_fma_f32 acc, acc, m, c (src0 = acc, src1 = m, src2 = c); FMAC: v_fmac_f32 acc, m, c
// (acc += m * c: src0 = m, src1 = c, and the accumulator is the third operand read)
#define FMA3(a, m, c) "v_fma_f32 v" #a ", v" #a ", v" #m ", v" #c "\n\t"
#define FMAC(a, m, c) "v_fmac_f32 v" #a ", v" #m ", v" #c "\n\t"
#define F64(F4, m, c) \
  F4(40, m, c) F4(44, m, c) F4(48, m, c) F4(52, m, c) F4(56, m, c) F4(60, m, c) F4(64, m, c) F4(68, m, c) \
  F4(72, m, c) F4(76, m, c) F4(80, m, c) F4(84, m, c) F4(88, m, c) F4(92, m, c) F4(96, m, c) F4(100, m, c)
#define FMA_BANK_KERNEL(name, F4, m, c)                                                                \
  __global__ void name(Stamp* st, float* out, int iters, float seed) {                               \
    unsigned long long c0, t0;                                                                       \
    float r = 0.0f;                                                                                  \
    stamp_begin(c0, t0);                                                                             \
    asm volatile(                                                                                    \
        "v_mov_b32 v40, %[seed]\n\tv_mov_b32 v44, %[seed]\n\tv_mov_b32 v48, %[seed]\n\tv_mov_b32 v52, %[seed]\n\t"  \
        "v_mov_b32 v56, %[seed]\n\tv_mov_b32 v60, %[seed]\n\tv_mov_b32 v64, %[seed]\n\tv_mov_b32 v68, %[seed]\n\t"  \
        "v_mov_b32 v72, %[seed]\n\tv_mov_b32 v76, %[seed]\n\tv_mov_b32 v80, %[seed]\n\tv_mov_b32 v84, %[seed]\n\t"  \
        "v_mov_b32 v88, %[seed]\n\tv_mov_b32 v92, %[seed]\n\tv_mov_b32 v96, %[seed]\n\tv_mov_b32 v100, %[seed]\n\t" \
        "v_mov_b32 v" #m ", 0x3f7fbe77\n\tv_mov_b32 v" #c ", 0x3a83126f\n\t"                         \
        "s_mov_b32 s40, %[iters]\n"                                                                  \
        "1:\n\t" F64(F4, m, c) F64(F4, m, c) F64(F4, m, c) F64(F4, m, c)                                             \
        "s_sub_u32 s40, s40, 1\n\t"                                                                  \
        "s_cmp_lg_u32 s40, 0\n\t"                                                                    \
        "s_cbranch_scc1 1b\n\t"                                                                      \
        "v_add_f32 %[r], v40, v100\n\t"                                                              \
        : [r] "=v"(r)                                                                                \
        : [seed] "v"(seed + threadIdx.x), [iters] "s"(iters)                                         \
        : "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68", "v72", "v76", "v80", "v84", "v88", \
          "v92", "v96", "v100", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",      \
          "s40", "scc");                                                                             \
    stamp_end(st, c0, t0);                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                  \
  }
FMA_BANK_KERNEL(k_fma_banks_000, FMA3, 104, 108)
FMA_BANK_KERNEL(k_fma_banks_002, FMA3, 104, 110)
FMA_BANK_KERNEL(k_fma_banks_012, FMA3, 105, 110)
FMA_BANK_KERNEL(k_fma_banks_010, FMA3, 105, 108)
FMA_BANK_KERNEL(k_fma_banks_011, FMA3, 105, 109)
FMA_BANK_KERNEL(k_fmac_acc0_01, FMAC, 104, 109)   // accumulator and src0 in bank 0
FMA_BANK_KERNEL(k_fmac_acc0_10, FMAC, 105, 108)   // accumulator and src1 in bank 0
FMA_BANK_KERNEL(k_fmac_acc0_11, FMAC, 105, 109)   // src0 and src1 in bank 1
FMA_BANK_KERNEL(k_fmac_acc0_12, FMAC, 105, 110)   // three banks

// The production scan block as the compiler emits it today (four sphere records in v[0:15],
// two paths, 64 fma + fold), registers chosen by hand.  Per path: multipliers of x, y, z for the
// hb chain (NX, NY, NZ) and the qh chain (OX, OY, OZ), the addends OD and NOO.  The records'
// x, y, z, w sit in banks 0, 1, 2, 3, so a multiplier of x in bank 0 (of y in bank 1, of z in
// bank 2) makes src0 and src1 of its instruction share a bank.
#define SPH(x, y, z, w, t1, t2, ANX, ANY, ANZ, AOD, AOX, AOY, AOZ, ANOO, BNX, BNY, BNZ, BOD, BOX, BOY, BOZ, BNOO, D0, D1) \
  "v_fma_f32 v" #t1 ", v" #x ", v" #AOX ", v" #w "\n\t"      /* qh A = x*ox + w   */ \
  "v_fmac_f32 v" #w ", v" #x ", v" #BOX "\n\t"               /* qh B (in place)   */ \
  "v_fma_f32 v" #t2 ", v" #x ", v" #ANX ", v" #AOD "\n\t"    /* hb A              */ \
  "v_fma_f32 v" #x ", v" #x ", v" #BNX ", v" #BOD "\n\t"     /* hb B (in place)   */ \
  "v_fmac_f32 v" #w ", v" #y ", v" #BOY "\n\t"                                       \
  "v_fmac_f32 v" #x ", v" #y ", v" #BNY "\n\t"                                       \
  "v_fmac_f32 v" #w ", v" #z ", v" #BOZ "\n\t"                                       \
  "v_fmac_f32 v" #t2 ", v" #y ", v" #ANY "\n\t"                                      \
  "v_fmac_f32 v" #t1 ", v" #y ", v" #AOY "\n\t"                                      \
  "v_fmac_f32 v" #x ", v" #z ", v" #BNZ "\n\t"                                       \
  "v_fma_f32 v" #D1 ", 2.0, v" #w ", v" #BNOO "\n\t"                                 \
  "v_fmac_f32 v" #D1 ", v" #x ", v" #x "\n\t"                                        \
  "v_fmac_f32 v" #t2 ", v" #z ", v" #ANZ "\n\t"                                      \
  "v_fmac_f32 v" #t1 ", v" #z ", v" #AOZ "\n\t"                                      \
  "v_fma_f32 v" #D0 ", 2.0, v" #t1 ", v" #ANOO "\n\t"                                \
  "v_fmac_f32 v" #D0 ", v" #t2 ", v" #t2 "\n\t"
#define SCAN_BLOCK_KERNEL(name, ANX, ANY, ANZ, AOD, AOX, AOY, AOZ, ANOO, BNX, BNY, BNZ, BOD, BOX, BOY, BOZ, BNOO) \
  __global__ void name(Stamp* st, float* out, int iters, float seed) {                               \
    unsigned long long c0, t0;                                                                       \
    float r = 0.0f;                                                                                  \
    stamp_begin(c0, t0);                                                                             \
    asm volatile(                                                                                    \
        "v_mov_b32 v" #ANX ", 0.5\n\tv_mov_b32 v" #ANY ", 0.5\n\tv_mov_b32 v" #ANZ ", 0.5\n\tv_mov_b32 v" #AOD ", 0.5\n\t" \
        "v_mov_b32 v" #AOX ", 0.5\n\tv_mov_b32 v" #AOY ", 0.5\n\tv_mov_b32 v" #AOZ ", 0.5\n\tv_mov_b32 v" #ANOO ", 0.5\n\t" \
        "v_mov_b32 v" #BNX ", 0.5\n\tv_mov_b32 v" #BNY ", 0.5\n\tv_mov_b32 v" #BNZ ", 0.5\n\tv_mov_b32 v" #BOD ", 0.5\n\t" \
        "v_mov_b32 v" #BOX ", 0.5\n\tv_mov_b32 v" #BOY ", 0.5\n\tv_mov_b32 v" #BOZ ", 0.5\n\tv_mov_b32 v" #BNOO ", 0.5\n\t" \
        "s_mov_b32 s40, %[iters]\n"                                                                  \
        "1:\n\t"                                                                                     \
        "v_mov_b32 v0, %[seed]\n\tv_mov_b32 v1, %[seed]\n\tv_mov_b32 v2, %[seed]\n\tv_mov_b32 v3, %[seed]\n\t"     \
        "v_mov_b32 v4, %[seed]\n\tv_mov_b32 v5, %[seed]\n\tv_mov_b32 v6, %[seed]\n\tv_mov_b32 v7, %[seed]\n\t"     \
        "v_mov_b32 v8, %[seed]\n\tv_mov_b32 v9, %[seed]\n\tv_mov_b32 v10, %[seed]\n\tv_mov_b32 v11, %[seed]\n\t"   \
        "v_mov_b32 v12, %[seed]\n\tv_mov_b32 v13, %[seed]\n\tv_mov_b32 v14, %[seed]\n\tv_mov_b32 v15, %[seed]\n\t" \
        SPH(0, 1, 2, 3, 16, 17, ANX, ANY, ANZ, AOD, AOX, AOY, AOZ, ANOO, BNX, BNY, BNZ, BOD, BOX, BOY, BOZ, BNOO, 24, 25) \
        SPH(4, 5, 6, 7, 18, 19, ANX, ANY, ANZ, AOD, AOX, AOY, AOZ, ANOO, BNX, BNY, BNZ, BOD, BOX, BOY, BOZ, BNOO, 26, 27) \
        SPH(8, 9, 10, 11, 20, 21, ANX, ANY, ANZ, AOD, AOX, AOY, AOZ, ANOO, BNX, BNY, BNZ, BOD, BOX, BOY, BOZ, BNOO, 28, 29) \
        SPH(12, 13, 14, 15, 22, 23, ANX, ANY, ANZ, AOD, AOX, AOY, AOZ, ANOO, BNX, BNY, BNZ, BOD, BOX, BOY, BOZ, BNOO, 30, 31) \
        "v_max3_i32 v32, v24, v25, v26\n\tv_max3_i32 v32, v32, v27, v28\n\t"                         \
        "v_max3_i32 v32, v32, v29, v30\n\tv_max_i32 v32, v32, v31\n\t"                               \
        "s_sub_u32 s40, s40, 1\n\t"                                                                  \
        "s_cmp_lg_u32 s40, 0\n\t"                                                                    \
        "s_cbranch_scc1 1b\n\t"                                                                      \
        "v_mov_b32 %[r], v32\n\t"                                                                    \
        : [r] "=v"(r)                                                                                \
        : [seed] "v"(seed + threadIdx.x), [iters] "s"(iters)                                         \
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",    \
          "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", \
          "v27", "v28", "v29", "v30", "v31", "v32", "v80", "v81", "v82", "v83", "v84", "v85", "v86", \
          "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
          "v100", "v101", "v102", "v103", "s40", "scc");                                             \
    stamp_end(st, c0, t0);                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                  \
  }
// as the compiler allocates them today (path A's three hb multipliers collide with x, y, z)
SCAN_BLOCK_KERNEL(k_block_today, 96, 97, 90, 95, 98, 99, 100, 92, 87, 88, 81, 82, 83, 84, 85, 86)
// every multiplier in a bank other than its sphere component's
SCAN_BLOCK_KERNEL(k_block_free, 97, 98, 99, 95, 81, 82, 83, 92, 85, 86, 87, 88, 89, 90, 91, 93)
// every multiplier in the bank of its sphere component
SCAN_BLOCK_KERNEL(k_block_worst, 96, 97, 98, 95, 80, 81, 82, 92, 84, 85, 86, 88, 100, 101, 102, 93)

__global__ void k_pkfma(Stamp* st, float* out, int iters, float seed) {
  f2 a[8], m[8], c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = f2{seed + threadIdx.x + i, seed + i};
    m[i] = f2{0.999f - 1e-4f * i, 0.998f - 1e-4f * i};
    c[i] = f2{0.001f + 1e-4f * i, 0.002f};
  }
  unsigned long long c0, t0;
  stamp_begin(c0, t0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m[i], c[i]);
    }
  }
  stamp_end(st, c0, t0);
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// the production scan: blocks of four spheres, two paths per lane, records read one block ahead
template <int R>
__global__ void k_scan(Stamp* st, const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < n + 8; i += blockDim.x) lds[i] = spheres[i < n ? i : 0];
  __syncthreads();
  float ox[R], oy[R], oz[R], nx[R], ny[R], nz[R], od[R], noo[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f + (h & 7) * 1e-3f;
    nx[r] = 0.96f - ((h >> 16) & 255) * 1e-3f; ny[r] = 0.15f + (h & 3) * 1e-3f; nz[r] = 0.22f - (h >> 24) * 1e-3f;
    od[r] = -(ox[r] * nx[r] + oy[r] * ny[r] + oz[r] * nz[r]);
    noo[r] = -(ox[r] * ox[r] + oy[r] * oy[r] + oz[r] * oz[r]);
  }
  int cnt = 0;
  auto block = [&](const float4 (&s)[4]) {
    int m = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float hb = __builtin_fmaf(s[u].z, nz[r], __builtin_fmaf(s[u].y, ny[r], __builtin_fmaf(s[u].x, nx[r], od[r])));
        float qh = __builtin_fmaf(s[u].z, oz[r], __builtin_fmaf(s[u].y, oy[r], __builtin_fmaf(s[u].x, ox[r], s[u].w)));
        float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(qh, 2.0f, noo[r]));
        m = max(m, __float_as_int(disc));
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
  unsigned long long c0, t0;
  stamp_begin(c0, t0);
  for (int rep = 0; rep < reps; ++rep) {
    float4 A[4], B[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) A[u] = lds[u];
    for (int i = 0; i + 8 <= n; i += 8) {
#pragma unroll
      for (int u = 0; u < 4; ++u) B[u] = lds[i + 4 + u];
      __builtin_amdgcn_sched_barrier(0);
      block(A);
#pragma unroll
      for (int u = 0; u < 4; ++u) A[u] = lds[i + 8 + u];
      __builtin_amdgcn_sched_barrier(0);
      block(B);
    }
    ox[0] += 1e-6f;
  }
  stamp_end(st, c0, t0);
  out[tid] = (float)cnt;
}

// The production scan with the sphere records read through the scalar cache into SGPRs, four
// records (64 bytes) per s_load_dwordx16, requested one block ahead by hand: the load of block
// k+1 is issued before block k is evaluated and waited for only when block k+1 begins, so its
// latency hides behind a whole block of arithmetic (the compiler, left to itself, waits right
// after issuing).  Every fma then reads two VGPRs and one SGPR instead of three VGPRs.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int R>
__global__ void k_scan_smem(Stamp* st, const float4* __restrict__ spheres, int n, int reps, float* out) {
  float ox[R], oy[R], oz[R], nx[R], ny[R], nz[R], od[R], hnoo[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f + (h & 7) * 1e-3f;
    nx[r] = 0.96f - ((h >> 16) & 255) * 1e-3f; ny[r] = 0.15f + (h & 3) * 1e-3f; nz[r] = 0.22f - (h >> 24) * 1e-3f;
    od[r] = -(ox[r] * nx[r] + oy[r] * ny[r] + oz[r] * nz[r]);
    hnoo[r] = -0.5f * (ox[r] * ox[r] + oy[r] * oy[r] + oz[r] * oz[r]);
  }
  int cnt = 0;
  auto block = [&](const f16v& s) {
    int m = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float hb = __builtin_fmaf(s[4 * u + 2], nz[r], __builtin_fmaf(s[4 * u + 1], ny[r], __builtin_fmaf(s[4 * u], nx[r], od[r])));
        float t = __builtin_fmaf(s[4 * u + 2], oz[r], __builtin_fmaf(s[4 * u + 1], oy[r], __builtin_fmaf(s[4 * u], ox[r], hnoo[r])));
        float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(t, 2.0f, s[4 * u + 3]));
        m = max(m, __float_as_int(disc));
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
  unsigned long long c0, t0;
  stamp_begin(c0, t0);
  for (int rep = 0; rep < reps; ++rep) {
    const float4* base = spheres;
    f16v A, B;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(A) : "s"(base) : "memory");
    for (int i = 0; i + 8 <= n; i += 8) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(A) : : "memory");
      asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(B) : "s"(base) : "memory");
      block(A);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(B) : : "memory");
      asm volatile("s_load_dwordx16 %0, %1, 0x80" : "=s"(A) : "s"(base) : "memory");
      block(B);
      base += 8;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(A) : : "memory");
    ox[0] += 1e-6f;
  }
  stamp_end(st, c0, t0);
  out[tid] = (float)cnt;
}

static void report(const char* name, int wps, double instr_per_wave, const std::vector<Stamp>& st,
                   float ms, double flop_per_instr_lane) {
  double cyc = 0, ticks = 0, cmax = 0;
  for (const Stamp& s : st) {
    cyc += (double)s.cycles;
    ticks += (double)s.ticks;
    if ((double)s.cycles > cmax) cmax = (double)s.cycles;
  }
  cyc /= st.size();
  ticks /= st.size();
  const double ghz = cyc / ticks * 0.1;                 // ticks are 10 ns
  // cycles per VALU instruction per SIMD over the whole launch (the waves of a launch do not all
  // run at the same time, so a wave's own duration would flatter the figure)
  const double cpi = ms * 1e-3 * ghz * 1e9 / (wps * instr_per_wave);
  const double tflops = instr_per_wave * st.size() * 64.0 * flop_per_instr_lane / (ms * 1e-3) * 1e-12;
  printf("%-10s waves/SIMD %d : %8.3f ms  wave loop %.0f kcycles (max %.0f) at %.3f GHz  "
         "%.3f cycles per VALU instr per SIMD  %.1f TFLOP/s executed\n",
         name, wps, ms, cyc / 1e3, cmax / 1e3, ghz, cpi, tflops);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device CUs %d  nominal clock %d kHz  %s\n", prop.multiProcessorCount, prop.clockRate,
         prop.gcnArchName);
  const int cus = prop.multiProcessorCount;
  const int n = 488;
  std::vector<float4> sph(n + 16);
  for (int i = 0; i < n + 16; ++i) {
    const float x = (float)(i % 22) - 11.f + 0.3f, z = (float)(i / 22) - 11.f + 0.4f;
    const float k = x * x + 0.04f + z * z - 0.04f;
    sph[i] = make_float4(x, 0.2f, z, -0.5f * k);
  }
  float4* d_sph;
  CK(hipMalloc(&d_sph, sph.size() * sizeof(float4)));
  CK(hipMemcpy(d_sph, sph.data(), sph.size() * sizeof(float4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {
    const int threads = 256;                       // one wave per SIMD per workgroup
    const int blocks = cus * wps;
    const size_t waves = (size_t)blocks * threads / 64;
    Stamp* d_st;
    float* d_out;
    CK(hipMalloc(&d_st, waves * sizeof(Stamp)));
    CK(hipMalloc(&d_out, (size_t)blocks * threads * sizeof(float)));
    std::vector<Stamp> st(waves);
    float ms;
    const int iters = 20000;
    for (int pass = 0; pass < 2; ++pass) {         // the second pass is the one reported
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fma<true>, dim3(blocks), dim3(threads), 0, 0, d_st, d_out, iters, 1.0f, 0.999f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    report("fmamk", wps, (double)iters * 64, st, ms, 2.0);
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fma<false>, dim3(blocks), dim3(threads), 0, 0, d_st, d_out, iters, 1.0f, 0.999f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    report("fma 3vgpr", wps, (double)iters * 64, st, ms, 2.0);
    struct Variant { const char* name; void (*fn)(Stamp*, float*, int, float); };
    const Variant variants[] = {
        {"fma 000", k_fma_banks_000}, {"fma 002", k_fma_banks_002}, {"fma 012", k_fma_banks_012},
        {"fma 010", k_fma_banks_010}, {"fma 011", k_fma_banks_011}, {"fmac a=s0", k_fmac_acc0_01},
        {"fmac a=s1", k_fmac_acc0_10}, {"fmac s0=s1", k_fmac_acc0_11}, {"fmac 3bank", k_fmac_acc0_12},
        {"blk today", k_block_today}, {"blk free", k_block_free}, {"blk worst", k_block_worst}};
    for (const Variant& v : variants) {
      if (wps != 4 && wps != 8) continue;
      for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(threads), 0, 0, d_st, d_out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
      }
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
      // the block kernels run 16 moves + 64 fma + 4 max per iteration
      const bool blk = strncmp(v.name, "blk", 3) == 0;
      report(v.name, wps, (double)iters * (blk ? 84 : 64), st, ms, blk ? 2.0 * 64 / 84 : 2.0);
    }
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(threads), 0, 0, d_st, d_out, iters, 1.0f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    report("pk_fma", wps, (double)iters * 32, st, ms, 4.0);
    const int reps = 400;
    const size_t lds = (n + 8) * sizeof(float4);
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_scan<2>), dim3(blocks), dim3(threads), lds, 0, d_st, d_sph, n, reps, d_out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    // per wave: reps * (488 / 8) iterations * 2 blocks * (64 fma + 4 max3 + 1 cmp)
    report("scan R=2", wps, (double)reps * (n / 8) * 2 * 69, st, ms, 2.0 * 64.0 / 69.0);
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_scan_smem<2>), dim3(blocks), dim3(threads), 0, 0, d_st, d_sph, n, reps, d_out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    report("scan smem", wps, (double)reps * (n / 8) * 2 * 69, st, ms, 2.0 * 64.0 / 69.0);
    CK(hipFree(d_st));
    CK(hipFree(d_out));
  }
  return 0;
}
