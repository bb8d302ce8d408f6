// clock_probe.hip - is the scan loop held by VALU issue or by the clock?
//
// Every wave stamps the shader-clock counter (clock64 = s_memtime) and the constant 100 MHz
// counter (wall_clock64 = s_memrealtime) around its loop.  From them:
//   actual shader clock while the loop runs  = d(clock64) / d(wall_clock64) * 100 MHz
//   cycles per VALU instruction per SIMD     = d(clock64) / (waves per SIMD * instructions per wave)
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
  const double cpi = cyc / (wps * instr_per_wave);     // cycles per VALU instruction per SIMD
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
  std::vector<float4> sph(n + 8);
  for (int i = 0; i < n + 8; ++i) {
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
    CK(hipFree(d_st));
    CK(hipFree(d_out));
  }
  return 0;
}
