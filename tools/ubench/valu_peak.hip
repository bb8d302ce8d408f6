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

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char** argv) {
  printf("== build: %s\n", argc > 1 ? argv[1] : "?");
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s  CUs %d  clock %d kHz  lds/block %zu  gcn %s\n", p.name, cus, p.clockRate,
         p.sharedMemPerBlock, p.gcnArchName);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 1024 * 8));

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
