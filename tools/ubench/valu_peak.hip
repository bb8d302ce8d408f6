// Microbenchmarks that fix the inner-loop design of the ray/sphere kernel on gfx950:
//   1. plain v_fma_f32 vs v_pk_fma_f32 issue rate at 1..8 waves per SIMD
//   2. a brute-force ray/sphere discriminant loop with the sphere record (cx,cy,cz,r^2)
//      delivered three ways: LDS broadcast read, scalar loads (SGPR operands), and
//      with 1/2/4 rays blocked per lane.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// ---- 1a. plain fma: 8 independent chains per lane
__global__ void k_fma(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 0.001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // inline asm so the SLP vectoriser cannot fuse pairs into v_pk_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t"
                   "v_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                   "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                   "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// ---- 1b. packed fma: 8 independent float2 chains per lane
__global__ void k_pkfma(float* out, int iters, float seed) {
  f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  f2 a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const f2 m = {0.999f, 0.998f}, c = {0.001f, 0.002f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c);
      a2 = __builtin_elementwise_fma(a2, m, c); a3 = __builtin_elementwise_fma(a3, m, c);
      a4 = __builtin_elementwise_fma(a4, m, c); a5 = __builtin_elementwise_fma(a5, m, c);
      a6 = __builtin_elementwise_fma(a6, m, c); a7 = __builtin_elementwise_fma(a7, m, c);
    }
  }
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// ---- 2. discriminant loop. R rays per lane; sphere record from LDS (broadcast) or SGPR.
// Counts (per lane) spheres with disc >= 0 so nothing is dead code; mirrors the real
// hot loop: 3 sub, 3 (mul,fma,fma), 3 fma, mul, fma, cmp.
template <int R, bool SMEM>
__global__ void k_disc(const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];
  if (!SMEM) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = spheres[i];
    __syncthreads();
  }
  float ox[R], oy[R], oz[R], dx[R], dy[R], dz[R], a[R];
  int cnt[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f;
    dx[r] = -0.96f + ((h >> 16) & 255) * 1e-3f; dy[r] = -0.15f; dz[r] = -0.22f + (h >> 24) * 1e-3f;
    a[r] = dx[r] * dx[r] + dy[r] * dy[r] + dz[r] * dz[r];
    cnt[r] = 0;
  }
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      float4 s = SMEM ? spheres[i] : lds[i];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float ocx = ox[r] - s.x, ocy = oy[r] - s.y, ocz = oz[r] - s.z;
        float hb = __builtin_fmaf(ocz, dz[r], __builtin_fmaf(ocy, dy[r], ocx * dx[r]));
        float cc = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, -s.w)));
        float disc = __builtin_fmaf(hb, hb, -(a[r] * cc));
        if (disc >= 0.f) cnt[r]++;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ox[r] += 1e-6f;  // keep reps from being merged
  }
  int tot = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) tot += cnt[r];
  out[tid] = (float)tot;
}

// ---- 3. two spheres per packed instruction.  LDS holds sphere PAIRS:
//   pairs[2k]   = (cx0, cx1, cy0, cy1),  pairs[2k+1] = (cz0, cz1, r2_0, r2_1)
// Ray components are scalars splat into both halves (free: op_sel).  NORM drops the a*c
// multiply (direction pre-normalised): 10 packed instructions per 2 tests, else 11.
template <int R, bool NORM>
__global__ void k_disc_pk(const float4* __restrict__ pairs, int npairs, int reps, float* out) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < 2 * npairs; i += blockDim.x) lds[i] = pairs[i];
  __syncthreads();
  float ox[R], oy[R], oz[R], dx[R], dy[R], dz[R], a[R];
  int cnt[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f;
    dx[r] = -0.96f + ((h >> 16) & 255) * 1e-3f; dy[r] = -0.15f; dz[r] = -0.22f + (h >> 24) * 1e-3f;
    a[r] = dx[r] * dx[r] + dy[r] * dy[r] + dz[r] * dz[r];
    cnt[r] = 0;
  }
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 2
    for (int k = 0; k < npairs; ++k) {
      const float4 A = lds[2 * k], B = lds[2 * k + 1];
      const f2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, r2 = {B.z, B.w};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f2 ocx = f2(ox[r]) - cx, ocy = f2(oy[r]) - cy, ocz = f2(oz[r]) - cz;
        const f2 hb = __builtin_elementwise_fma(ocz, f2(dz[r]),
                      __builtin_elementwise_fma(ocy, f2(dy[r]), ocx * f2(dx[r])));
        const f2 cc = __builtin_elementwise_fma(ocz, ocz,
                      __builtin_elementwise_fma(ocy, ocy, __builtin_elementwise_fma(ocx, ocx, -r2)));
        f2 disc;
        if (NORM) disc = __builtin_elementwise_fma(hb, hb, -cc);
        else disc = __builtin_elementwise_fma(hb, hb, -(f2(a[r]) * cc));
        if (!(disc.x < 0.f)) cnt[r]++;
        if (!(disc.y < 0.f)) cnt[r]++;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ox[r] += 1e-6f;
  }
  int tot = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) tot += cnt[r];
  out[tid] = (float)tot;
}


// ---- 4. the conservative 8-instruction discriminant of the production scan, plain and packed
// over PATH pairs.  Record = (cx, cy, cz, w).  Per path: nd = -d/|d|, od = o.d/|d|, o, noo.
//   hb = fma(cz,ndz, fma(cy,ndy, fma(cx,ndx, od)));  qh = fma(cz,oz, fma(cy,oy, fma(cx,ox, w)))
//   disc = fma(hb,hb, fma(qh,2,noo))
// Plain: 8 VALU per test.  Packed: the two paths of a lane sit in the halves of a register pair
// and the record's component is broadcast to both halves with op_sel: 8 v_pk_fma_f32 per sphere
// per path PAIR.  Blocks of 4 spheres, sign bits folded with integer max like the real kernel.
template <int R>
__global__ void k_cons(const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = spheres[i];
  __syncthreads();
  float ox[R], oy[R], oz[R], nx[R], ny[R], nz[R], od[R], noo[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f;
    nx[r] = 0.96f - ((h >> 16) & 255) * 1e-3f; ny[r] = 0.15f; nz[r] = 0.22f - (h >> 24) * 1e-3f;
    od[r] = -(ox[r] * nx[r] + oy[r] * ny[r] + oz[r] * nz[r]);
    noo[r] = -(ox[r] * ox[r] + oy[r] * oy[r] + oz[r] * oz[r]);
  }
  int cnt = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int i = 0; i + 4 <= n; i += 4) {
      float4 s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] = lds[i + u];
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
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ox[r] += 1e-6f;
  }
  out[tid] = (float)cnt;
}

// D = S.{lo|hi} (broadcast) * B + C       (C a per-path pair)
__device__ __forceinline__ f2 pk_lo(f2 s, f2 b, f2 c) {
  f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f2 pk_hi(f2 s, f2 b, f2 c) {
  f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(s), "v"(b), "v"(c)); return d;
}
// D = S.lo (broadcast) * B + T.hi (broadcast)
__device__ __forceinline__ f2 pk_lo_chi(f2 s, f2 b, f2 t) {
  f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s), "v"(b), "v"(t)); return d;
}

template <int RP, bool ASM>
__global__ void k_cons_pk(const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = spheres[i];
  __syncthreads();
  f2 ox[RP], oy[RP], oz[RP], nx[RP], ny[RP], nz[RP], od[RP], noo[RP];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    float t[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned h = (tid * 2 * RP + 2 * p + k) * 2654435761u;
      t[k][0] = 13.f + (h & 255) * 1e-3f; t[k][1] = 2.f + ((h >> 8) & 255) * 1e-3f; t[k][2] = 3.f;
      t[k][3] = 0.96f - ((h >> 16) & 255) * 1e-3f; t[k][4] = 0.15f; t[k][5] = 0.22f - (h >> 24) * 1e-3f;
      t[k][6] = -(t[k][0] * t[k][3] + t[k][1] * t[k][4] + t[k][2] * t[k][5]);
      t[k][7] = -(t[k][0] * t[k][0] + t[k][1] * t[k][1] + t[k][2] * t[k][2]);
    }
    ox[p] = f2{t[0][0], t[1][0]}; oy[p] = f2{t[0][1], t[1][1]}; oz[p] = f2{t[0][2], t[1][2]};
    nx[p] = f2{t[0][3], t[1][3]}; ny[p] = f2{t[0][4], t[1][4]}; nz[p] = f2{t[0][5], t[1][5]};
    od[p] = f2{t[0][6], t[1][6]}; noo[p] = f2{t[0][7], t[1][7]};
  }
  int cnt = 0;
  const f2 two = {2.0f, 2.0f};
  for (int rep = 0; rep < reps; ++rep) {
    for (int i = 0; i + 4 <= n; i += 4) {
      float4 s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] = lds[i + u];
      int m = (int)0x80000000;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f2 sxy = {s[u].x, s[u].y}, szw = {s[u].z, s[u].w};
#pragma unroll
        for (int p = 0; p < RP; ++p) {
          f2 hb, qh, disc;
          if (ASM) {
            hb = pk_lo(szw, nz[p], pk_hi(sxy, ny[p], pk_lo(sxy, nx[p], od[p])));
            qh = pk_lo(szw, oz[p], pk_hi(sxy, oy[p], pk_lo_chi(sxy, ox[p], szw)));
          } else {
            hb = __builtin_elementwise_fma(f2(s[u].z), nz[p], __builtin_elementwise_fma(f2(s[u].y), ny[p],
                 __builtin_elementwise_fma(f2(s[u].x), nx[p], od[p])));
            qh = __builtin_elementwise_fma(f2(s[u].z), oz[p], __builtin_elementwise_fma(f2(s[u].y), oy[p],
                 __builtin_elementwise_fma(f2(s[u].x), ox[p], f2(s[u].w))));
          }
          disc = __builtin_elementwise_fma(hb, hb, __builtin_elementwise_fma(qh, two, noo[p]));
          m = max(m, max(__float_as_int(disc.x), __float_as_int(disc.y)));
        }
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) ox[p] += 1e-6f;
  }
  out[tid] = (float)cnt;
}


// ---- 5. the same discriminant with the matrix pipe doing part of the work.
// v_mfma_f32_16x16x4_f32 is bit-for-bit the k-ordered fmaf chain, so with A = (1|w, cx, cy, cz)
// per sphere and B = (od|1, ndx|ox, ndy|oy, ndz|oz) per ray it produces exactly the hb and qh of
// the plain form for a 16-sphere x 16-ray tile, in the C/D layout (col = ray = lane & 15,
// row = sphere = 4 * (lane >> 4) + reg).  MODE 0: every ray through MFMA (NT tiles of 16 rays per
// wave).  MODE 1: additionally one ray per lane through the plain VALU form (the two pipes run
// side by side).
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT, int MODE>
__global__ void k_mix(const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];
  float* ah = reinterpret_cast<float*>(lds + n);        // [block][64]
  float* aq = ah + (n / 16) * 64;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = spheres[i];
  for (int e = threadIdx.x; e < (n / 16) * 64; e += blockDim.x) {
    const int b = e / 64, l = e % 64, i = b * 16 + (l & 15), k = l >> 4;
    const float4 s = spheres[i];
    const float comp = k == 1 ? s.x : k == 2 ? s.y : s.z;
    ah[e] = k == 0 ? 1.0f : comp;
    aq[e] = k == 0 ? s.w : comp;
  }
  __syncthreads();
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned lane = threadIdx.x & 63;
  // classic ray of this lane
  float ox, oy, oz, nx, ny, nz, od, noo;
  {
    unsigned h = tid * 2654435761u;
    ox = 13.f + (h & 255) * 1e-3f; oy = 2.f + ((h >> 8) & 255) * 1e-3f; oz = 3.f;
    nx = 0.96f - ((h >> 16) & 255) * 1e-3f; ny = 0.15f; nz = 0.22f - (h >> 24) * 1e-3f;
    od = -(ox * nx + oy * ny + oz * nz);
    noo = -(ox * ox + oy * oy + oz * oz);
  }
  // B operands of the NT ray tiles: component (lane >> 4) of ray (lane & 15) of the tile
  float bh[NT], bq[NT], tn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    unsigned h = ((tid & ~63u) * NT + t * 16 + (lane & 15)) * 2246822519u + 12345u;
    float rx = 13.f + (h & 255) * 1e-3f, ry = 2.f + ((h >> 8) & 255) * 1e-3f, rz = 3.f;
    float mx = 0.96f - ((h >> 16) & 255) * 1e-3f, my = 0.15f, mz = 0.22f - (h >> 24) * 1e-3f;
    float rod = -(rx * mx + ry * my + rz * mz);
    const int k = lane >> 4;
    bh[t] = k == 0 ? rod : k == 1 ? mx : k == 2 ? my : mz;
    bq[t] = k == 0 ? 1.0f : k == 1 ? rx : k == 2 ? ry : rz;
    tn[t] = -(rx * rx + ry * ry + rz * rz);
  }
  int cnt = 0;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int rep = 0; rep < reps; ++rep) {
    for (int b = 0; b + 1 <= n / 16; ++b) {
      const float a_h = ah[b * 64 + lane], a_q = aq[b * 64 + lane];
      int m = (int)0x80000000;
      f4 dh[NT], dq[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        dh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_h, bh[t], zero, 0, 0, 0);
        dq[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_q, bq[t], zero, 0, 0, 0);
      }
      if (MODE == 1) {
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
          float4 s[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) s[u] = lds[b * 16 + u4 * 4 + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float hb = __builtin_fmaf(s[u].z, nz, __builtin_fmaf(s[u].y, ny, __builtin_fmaf(s[u].x, nx, od)));
            float qh = __builtin_fmaf(s[u].z, oz, __builtin_fmaf(s[u].y, oy, __builtin_fmaf(s[u].x, ox, s[u].w)));
            float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(qh, 2.0f, noo));
            m = max(m, __float_as_int(disc));
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float disc = __builtin_fmaf(dh[t][e], dh[t][e], __builtin_fmaf(dq[t][e], 2.0f, tn[t]));
          m = max(m, __float_as_int(disc));
        }
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
    }
    ox += 1e-6f;
#pragma unroll
    for (int t = 0; t < NT; ++t) tn[t] -= 1e-6f;
  }
  out[tid] = (float)cnt;
}


// ---- 6. the same two forms with the production loop's read-ahead: the next block's records are
// read from LDS (into a second register set) while the current block is evaluated.
template <int R, bool PK>
__global__ void k_cons_ahead(const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < n + 8; i += blockDim.x) lds[i] = spheres[i < n ? i : 0];
  __syncthreads();
  constexpr int RP = R / 2;
  float ox[R], oy[R], oz[R], nx[R], ny[R], nz[R], od[R], noo[R];
  f2 pox[RP], poy[RP], poz[RP], pnx[RP], pny[RP], pnz[RP], pod[RP], pnoo[RP];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f + (h & 7) * 1e-3f;
    nx[r] = 0.96f - ((h >> 16) & 255) * 1e-3f; ny[r] = 0.15f + (h & 3) * 1e-3f; nz[r] = 0.22f - (h >> 24) * 1e-3f;
    od[r] = -(ox[r] * nx[r] + oy[r] * ny[r] + oz[r] * nz[r]);
    noo[r] = -(ox[r] * ox[r] + oy[r] * oy[r] + oz[r] * oz[r]);
  }
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    pox[p] = f2{ox[2 * p], ox[2 * p + 1]}; poy[p] = f2{oy[2 * p], oy[2 * p + 1]}; poz[p] = f2{oz[2 * p], oz[2 * p + 1]};
    pnx[p] = f2{nx[2 * p], nx[2 * p + 1]}; pny[p] = f2{ny[2 * p], ny[2 * p + 1]}; pnz[p] = f2{nz[2 * p], nz[2 * p + 1]};
    pod[p] = f2{od[2 * p], od[2 * p + 1]}; pnoo[p] = f2{noo[2 * p], noo[2 * p + 1]};
  }
  int cnt = 0;
  const f2 two = {2.0f, 2.0f};
  auto block = [&](const float4 (&s)[4]) {
    int m = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (PK) {
#pragma unroll
        for (int p = 0; p < RP; ++p) {
          f2 hb = __builtin_elementwise_fma(f2(s[u].z), pnz[p], __builtin_elementwise_fma(f2(s[u].y), pny[p],
                  __builtin_elementwise_fma(f2(s[u].x), pnx[p], pod[p])));
          f2 qh = __builtin_elementwise_fma(f2(s[u].z), poz[p], __builtin_elementwise_fma(f2(s[u].y), poy[p],
                  __builtin_elementwise_fma(f2(s[u].x), pox[p], f2(s[u].w))));
          f2 disc = __builtin_elementwise_fma(hb, hb, __builtin_elementwise_fma(qh, two, pnoo[p]));
          m = max(m, max(__float_as_int(disc.x), __float_as_int(disc.y)));
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float hb = __builtin_fmaf(s[u].z, nz[r], __builtin_fmaf(s[u].y, ny[r], __builtin_fmaf(s[u].x, nx[r], od[r])));
          float qh = __builtin_fmaf(s[u].z, oz[r], __builtin_fmaf(s[u].y, oy[r], __builtin_fmaf(s[u].x, ox[r], s[u].w)));
          float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(qh, 2.0f, noo[r]));
          m = max(m, __float_as_int(disc));
        }
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
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
    if (PK) pox[0] += 1e-6f;
  }
  out[tid] = (float)cnt;
}


// ---- 7. sphere records through the scalar cache instead of LDS: the records are wave-uniform,
// so they can live in SGPRs (read-ahead included) and leave the vector registers to the paths.
// One SGPR operand per instruction: the per-sphere constant joins in its own fma.
template <int R>
__global__ void k_cons_smem(const float4* __restrict__ spheres, int n, int reps, float* out) {
  typedef float cf4 __attribute__((ext_vector_type(4)));
  const __attribute__((address_space(4))) cf4* sp =
      (const __attribute__((address_space(4))) cf4*)spheres;
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
  auto block = [&](const cf4 (&s)[4]) {
    int m = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float hb = __builtin_fmaf(s[u].z, nz[r], __builtin_fmaf(s[u].y, ny[r], __builtin_fmaf(s[u].x, nx[r], od[r])));
        float t = __builtin_fmaf(s[u].z, oz[r], __builtin_fmaf(s[u].y, oy[r], __builtin_fmaf(s[u].x, ox[r], hnoo[r])));
        float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(t, 2.0f, s[u].w));
        m = max(m, __float_as_int(disc));
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
  for (int rep = 0; rep < reps; ++rep) {
    cf4 A[4], B[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) A[u] = sp[u];
    for (int i = 0; i + 8 <= n; i += 8) {
#pragma unroll
      for (int u = 0; u < 4; ++u) B[u] = sp[i + 4 + u];
      __builtin_amdgcn_sched_barrier(0);
      block(A);
#pragma unroll
      for (int u = 0; u < 4; ++u) A[u] = sp[i + 8 + u];
      __builtin_amdgcn_sched_barrier(0);
      block(B);
    }
    ox[0] += 1e-6f;
  }
  out[tid] = (float)cnt;
}

// ---- 8. the production form with two SPHERES per packed instruction: LDS holds sphere pairs as
// (x0,x1,y0,y1), (z0,z1,w0,w1); the per-path constants sit duplicated in register pairs, so no
// operand needs a lane select.  Blocks of two pairs (four spheres), read ahead like the others.
template <int R>
__global__ void k_cons_sphere_pairs(const float4* __restrict__ spheres, int n, int reps, float* out) {
  extern __shared__ float4 lds[];   // 2 float4 per pair
  for (int i = threadIdx.x; i < (n + 8) / 2; i += blockDim.x) {
    const float4 a = spheres[2 * i < n ? 2 * i : 0], b = spheres[2 * i + 1 < n ? 2 * i + 1 : 0];
    lds[2 * i] = make_float4(a.x, b.x, a.y, b.y);
    lds[2 * i + 1] = make_float4(a.z, b.z, a.w, b.w);
  }
  __syncthreads();
  f2 ox[R], oy[R], oz[R], nx[R], ny[R], nz[R], od[R], noo[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    const float x = 13.f + (h & 255) * 1e-3f, y = 2.f + ((h >> 8) & 255) * 1e-3f, z = 3.f + (h & 7) * 1e-3f;
    const float a = 0.96f - ((h >> 16) & 255) * 1e-3f, b = 0.15f + (h & 3) * 1e-3f, c = 0.22f - (h >> 24) * 1e-3f;
    ox[r] = f2{x, x}; oy[r] = f2{y, y}; oz[r] = f2{z, z};
    nx[r] = f2{a, a}; ny[r] = f2{b, b}; nz[r] = f2{c, c};
    const float d = -(x * a + y * b + z * c), q = -(x * x + y * y + z * z);
    od[r] = f2{d, d}; noo[r] = f2{q, q};
  }
  int cnt = 0;
  const f2 two = {2.0f, 2.0f};
  auto block = [&](const float4 (&s)[4]) {   // s[2p], s[2p+1]: pair p
    int m = (int)0x80000000;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f2 sx = {s[2 * p].x, s[2 * p].y}, sy = {s[2 * p].z, s[2 * p].w};
      const f2 sz = {s[2 * p + 1].x, s[2 * p + 1].y}, sw = {s[2 * p + 1].z, s[2 * p + 1].w};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f2 hb = __builtin_elementwise_fma(sz, nz[r], __builtin_elementwise_fma(sy, ny[r],
                __builtin_elementwise_fma(sx, nx[r], od[r])));
        f2 qh = __builtin_elementwise_fma(sz, oz[r], __builtin_elementwise_fma(sy, oy[r],
                __builtin_elementwise_fma(sx, ox[r], sw)));
        f2 disc = __builtin_elementwise_fma(hb, hb, __builtin_elementwise_fma(qh, two, noo[r]));
        m = max(m, max(__float_as_int(disc.x), __float_as_int(disc.y)));
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
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
  out[tid] = (float)cnt;
}

// ---- 9. sphere records from LDS and from the scalar cache side by side: a ds_read_b128 costs
// the LDS pipe 8 cycles however many lanes share its address, and at two paths per lane that
// pipe is as busy as the vector ALU, so every NTH block of four spheres comes through SGPRs.
template <int R, int LDS_BLOCKS>
__global__ void k_cons_mixed(const float4* __restrict__ spheres, int n, int reps, float* out) {
  typedef float cf4 __attribute__((ext_vector_type(4)));
  const __attribute__((address_space(4))) cf4* sp =
      (const __attribute__((address_space(4))) cf4*)spheres;
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < n + 8; i += blockDim.x) lds[i] = spheres[i < n ? i : 0];
  __syncthreads();
  float ox[R], oy[R], oz[R], nx[R], ny[R], nz[R], od[R], noo[R], hnoo[R];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned h = (tid * R + r) * 2654435761u;
    ox[r] = 13.f + (h & 255) * 1e-3f; oy[r] = 2.f + ((h >> 8) & 255) * 1e-3f; oz[r] = 3.f + (h & 7) * 1e-3f;
    nx[r] = 0.96f - ((h >> 16) & 255) * 1e-3f; ny[r] = 0.15f + (h & 3) * 1e-3f; nz[r] = 0.22f - (h >> 24) * 1e-3f;
    od[r] = -(ox[r] * nx[r] + oy[r] * ny[r] + oz[r] * nz[r]);
    noo[r] = -(ox[r] * ox[r] + oy[r] * oy[r] + oz[r] * oz[r]);
    hnoo[r] = 0.5f * noo[r];
  }
  int cnt = 0;
  auto block_v = [&](const float4 (&s)[4]) {
    int m = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float hb = __builtin_fmaf(s[u].z, nz[r], __builtin_fmaf(s[u].y, ny[r], __builtin_fmaf(s[u].x, nx[r], od[r])));
        float qh = __builtin_fmaf(s[u].z, oz[r], __builtin_fmaf(s[u].y, oy[r], __builtin_fmaf(s[u].x, ox[r], s[u].w)));
        float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(qh, 2.0f, noo[r]));
        m = max(m, __float_as_int(disc));
      }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
  auto block_s = [&](const cf4 (&s)[4]) {
    int m = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float hb = __builtin_fmaf(s[u].z, nz[r], __builtin_fmaf(s[u].y, ny[r], __builtin_fmaf(s[u].x, nx[r], od[r])));
        float t = __builtin_fmaf(s[u].z, oz[r], __builtin_fmaf(s[u].y, oy[r], __builtin_fmaf(s[u].x, ox[r], hnoo[r])));
        float disc = __builtin_fmaf(hb, hb, __builtin_fmaf(t, 2.0f, s[u].w));
        m = max(m, __float_as_int(disc));
      }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= 0) != 0, 0)) cnt += (m >= 0);
  };
  constexpr int STEP = 4 * (LDS_BLOCKS + 1);
  for (int rep = 0; rep < reps; ++rep) {
    float4 A[4];
    cf4 S[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      A[u] = lds[u];
      S[u] = sp[4 * LDS_BLOCKS + u];
    }
    for (int i = 0; i + STEP <= n; i += STEP) {
      // the scalar block of the NEXT round is requested now and carried around the loop, so
      // its latency hides behind this round's arithmetic (the buffer is padded for it)
      cf4 Sn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) Sn[u] = sp[i + STEP + 4 * LDS_BLOCKS + u];
#pragma unroll
      for (int b = 0; b < LDS_BLOCKS; ++b) {
        float4 N[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) N[u] = lds[i + 4 * (b + 1) + (b + 1 == LDS_BLOCKS ? 4 : 0) + u];
        __builtin_amdgcn_sched_barrier(0);
        block_v(A);
#pragma unroll
        for (int u = 0; u < 4; ++u) A[u] = N[u];
      }
      __builtin_amdgcn_sched_barrier(0);
      block_s(S);
#pragma unroll
      for (int u = 0; u < 4; ++u) S[u] = Sn[u];
    }
    ox[0] += 1e-6f;
  }
  out[tid] = (float)cnt;
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char** argv) {
  printf("== build: %s\n", argc > 1 ? argv[1] : "?");
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s  CUs %d  clock %d kHz  lds/block %zu  gcn %s\n", p.name, cus, p.clockRate,
         p.sharedMemPerBlock, p.gcnArchName);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 1024 * 8));


  if (argc > 2 && !strcmp(argv[2], "cons")) {
    const int n = 488, reps = 400;
    std::vector<float4> hs(n);
    // w chosen so that no discriminant is ever >= 0 (pure scan cost, like most blocks)
    for (int i = 0; i < n; ++i) hs[i] = make_float4((i % 22) - 11 + 0.3f, 0.2f, (i / 22) - 11 + 0.4f, -1e6f);
    float4* ds; CK(hipMalloc(&ds, sizeof(float4) * n));
    CK(hipMemcpy(ds, hs.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    auto run = [&](const char* name, auto kern, int R, int wps, int threads) {
      int blocks = cus * wps * 256 / threads;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), n * sizeof(float4), 0, ds, n, reps, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = time_ms(e0, e1);
        if (ms < best) best = ms;
      }
      CK(hipGetLastError());
      double tests = (double)blocks * threads * R * n * reps;
      printf("%-12s paths/lane %d waves/SIMD %d : %8.3f ms  %7.2f Gtests/s  = %5.1f%% of 157.3 TF at 17 flop/test\n",
             name, R, wps, best, tests / best * 1e-6, tests * 17 / best * 1e-9 / 157.3 * 100);
    };
    for (int wps : {2, 4, 8}) {
      run("cons", k_cons<2>, 2, wps, 256);
      run("cons", k_cons<4>, 4, wps, 256);
      run("cons_pk", (k_cons_pk<1, false>), 2, wps, 256);
      run("cons_pk", (k_cons_pk<2, false>), 4, wps, 256);
      run("cons_pk_asm", (k_cons_pk<1, true>), 2, wps, 256);
      run("cons_pk_asm", (k_cons_pk<2, true>), 4, wps, 256);
    }
    return 0;
  }


  if (argc > 2 && !strcmp(argv[2], "mix")) {
    const int n = 480, reps = 400;
    std::vector<float4> hs(n);
    for (int i = 0; i < n; ++i) hs[i] = make_float4((i % 22) - 11 + 0.3f, 0.2f, (i / 22) - 11 + 0.4f, -1e6f);
    float4* ds; CK(hipMalloc(&ds, sizeof(float4) * n));
    CK(hipMemcpy(ds, hs.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    auto run = [&](const char* name, auto kern, int rays_per_wave, int wps, int threads) {
      int blocks = cus * wps * 256 / threads;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), n * sizeof(float4) * 3, 0, ds, n, reps, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = time_ms(e0, e1);
        if (ms < best) best = ms;
      }
      CK(hipGetLastError());
      double tests = (double)blocks * (threads / 64) * rays_per_wave * n * reps;
      printf("%-12s rays/wave %3d waves/SIMD %d : %8.3f ms  %7.2f Gtests/s  = %5.1f%% of 157.3 TF at 17 flop/test\n",
             name, rays_per_wave, wps, best, tests / best * 1e-6, tests * 17 / best * 1e-9 / 157.3 * 100);
    };
    for (int wps : {1, 2, 4}) {
      run("cons", k_cons<2>, 128, wps, 256);
      run("mfma4", (k_mix<4, 0>), 64, wps, 256);
      run("mfma8", (k_mix<8, 0>), 128, wps, 256);
      run("mix1+4", (k_mix<4, 1>), 128, wps, 256);
      run("mix1+8", (k_mix<8, 1>), 192, wps, 256);
      run("mix1+6", (k_mix<6, 1>), 160, wps, 256);
    }
    return 0;
  }


  if (argc > 2 && !strcmp(argv[2], "ahead")) {
    const int n = (argc > 3 && !strcmp(argv[3], "mixed")) ? 480 : 488, reps = 400;
    std::vector<float4> hs(n);
    for (int i = 0; i < n; ++i) hs[i] = make_float4((i % 22) - 11 + 0.3f, 0.2f, (i / 22) - 11 + 0.4f, -1e6f);
    float4* ds; CK(hipMalloc(&ds, sizeof(float4) * (n + 32)));   // padded for the read-ahead
    CK(hipMemset(ds, 0, sizeof(float4) * (n + 32)));
    CK(hipMemcpy(ds, hs.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    auto run = [&](const char* name, auto kern, int R, int wps, int threads) {
      int blocks = cus * wps * 256 / threads;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), (n + 8) * sizeof(float4), 0, ds, n, reps, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = time_ms(e0, e1);
        if (ms < best) best = ms;
      }
      CK(hipGetLastError());
      double tests = (double)blocks * threads * R * n * reps;
      printf("%-14s paths/lane %d waves/SIMD %d : %8.3f ms  %7.2f Gtests/s  = %5.1f%% of 157.3 TF at 17 flop/test\n",
             name, R, wps, best, tests / best * 1e-6, tests * 17 / best * 1e-9 / 157.3 * 100);
    };
    if (argc > 3 && !strcmp(argv[3], "mixed")) {
      // n is rounded down to whole rounds by each kernel; 480 = 40 rounds of 12, 30 of 16, 60 of 8
      for (int wps : {4, 8}) {
        run("ahead(lds)", (k_cons_ahead<2, false>), 2, wps, 256);
        run("mixed 1+1", (k_cons_mixed<2, 1>), 2, wps, 256);
        run("mixed 2+1", (k_cons_mixed<2, 2>), 2, wps, 256);
        run("mixed 3+1", (k_cons_mixed<2, 3>), 2, wps, 256);
      }
      return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "spairs")) {
      for (int wps : {2, 4, 8}) {
        run("ahead(plain)", (k_cons_ahead<2, false>), 2, wps, 256);
        run("sphere_pairs", (k_cons_sphere_pairs<2>), 2, wps, 256);
        run("ahead(plain)", (k_cons_ahead<4, false>), 4, wps, 256);
        run("sphere_pairs", (k_cons_sphere_pairs<4>), 4, wps, 256);
      }
      return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "smem")) {
      for (int wps : {4, 5, 6, 8}) {
        run("ahead(lds)", (k_cons_ahead<2, false>), 2, wps, 256);
        run("smem", (k_cons_smem<2>), 2, wps, 256);
      }
      return 0;
    }
    for (int wps : {2, 4, 8}) {
      run("ahead", (k_cons_ahead<2, false>), 2, wps, 256);
      run("ahead_pk", (k_cons_ahead<2, true>), 2, wps, 256);
      run("ahead", (k_cons_ahead<4, false>), 4, wps, 256);
      run("ahead_pk", (k_cons_ahead<4, true>), 4, wps, 256);
    }
    return 0;
  }

  // --- 1. fma vs pk_fma
  const int iters = 4096;
  for (int wps : {1, 2, 4, 8}) {         // waves per SIMD
    int threads = 256;                   // 4 waves per block -> one per SIMD
    int blocks = cus * wps;
    for (int pk = 0; pk < 2; ++pk) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (pk) hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
        else    hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      double flop = (double)blocks * threads * iters * 64.0 * 2.0 * (pk ? 2 : 1);
      float ms = time_ms(e0, e1);
      printf("%-8s waves/SIMD %d : %8.3f ms  %8.2f TFLOP/s\n", pk ? "pk_fma" : "fma", wps, ms, flop / ms * 1e-9);
    }
  }

  // --- 2. discriminant loop
  const int n = 488, reps = 400;
  std::vector<float4> hs(n);
  for (int i = 0; i < n; ++i) hs[i] = make_float4((i % 22) - 11 + 0.3f, 0.2f, (i / 22) - 11 + 0.4f, 0.04f);
  float4* ds; CK(hipMalloc(&ds, sizeof(float4) * n));
  CK(hipMemcpy(ds, hs.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
  auto run = [&](const char* name, auto kern, int R, int wps) {
    int threads = 256, blocks = cus * wps;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), n * sizeof(float4), 0, ds, n, reps, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    CK(hipGetLastError());
    double tests = (double)blocks * threads * R * n * reps;
    float ms = time_ms(e0, e1);
    printf("%-10s R=%d waves/SIMD %d : %8.3f ms  %7.2f Gtests/s  %6.2f TFLOP/s (17/test) = %5.1f%% of 157.3\n",
           name, R, wps, ms, tests / ms * 1e-6, tests * 17 / ms * 1e-9, tests * 17 / ms * 1e-9 / 157.3 * 100);
  };
  {
    std::vector<float4> hp(n);   // n/2 pairs, two float4 each
    for (int k = 0; k < n / 2; ++k) {
      hp[2 * k] = make_float4(hs[2 * k].x, hs[2 * k + 1].x, hs[2 * k].y, hs[2 * k + 1].y);
      hp[2 * k + 1] = make_float4(hs[2 * k].z, hs[2 * k + 1].z, hs[2 * k].w, hs[2 * k + 1].w);
    }
    float4* dp; CK(hipMalloc(&dp, sizeof(float4) * n));
    CK(hipMemcpy(dp, hp.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    auto runp = [&](const char* name, auto kern, int R, int wps) {
      int threads = 256, blocks = cus * wps;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), n * sizeof(float4), 0, dp, n / 2, reps, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      CK(hipGetLastError());
      double tests = (double)blocks * threads * R * n * reps;
      float ms = time_ms(e0, e1);
      printf("%-10s R=%d waves/SIMD %d : %8.3f ms  %7.2f Gtests/s  %6.2f TFLOP/s (17/test) = %5.1f%% of 157.3\n",
             name, R, wps, ms, tests / ms * 1e-6, tests * 17 / ms * 1e-9, tests * 17 / ms * 1e-9 / 157.3 * 100);
    };
    for (int wps : {2, 4, 8}) {
      runp("pk2", k_disc_pk<1, false>, 1, wps);
      runp("pk2", k_disc_pk<2, false>, 2, wps);
      runp("pk2norm", k_disc_pk<1, true>, 1, wps);
      runp("pk2norm", k_disc_pk<2, true>, 2, wps);
    }
  }
  for (int wps : {4}) {
    run("lds", k_disc<1, false>, 1, wps);
    run("lds", k_disc<2, false>, 2, wps);
    run("lds", k_disc<4, false>, 4, wps);
    run("smem", k_disc<1, true>, 1, wps);
    run("smem", k_disc<2, true>, 2, wps);
    run("smem", k_disc<4, true>, 4, wps);
  }
  return 0;
}
