// sqrt_exhaustive.hip - the short correctly rounded square root of rt_kernels.hip (sqrt_rn: the
// hardware's v_sqrt_f32, its two neighbours, and the signs of their residuals) against the
// compiler's sqrtf on EVERY float: how many of the 2^32 bit patterns differ, by kind of input.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 sqrt_exhaustive.hip -o sqrt_exhaustive_bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float sqrt_short(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float below = __uint_as_float(__float_as_uint(s) - 1u);
  const float above = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_below = __builtin_fmaf(-below, s, x);
  const float r_above = __builtin_fmaf(-above, s, x);
  s = 0.0f >= r_below ? below : s;
  s = 0.0f < r_above ? above : s;
  return s;
}

// kinds: 0 = +0, 1 = positive denormal, 2 = normal below 2^-96, 3 = 2^-96 .. max, 4 = +inf,
// 5 = NaN, 6 = -0, 7 = negative (finite or infinite)
__device__ int kind_of(uint32_t u) {
  if (u == 0u) return 0;
  if (u == 0x80000000u) return 6;
  if (u & 0x80000000u) return (u & 0x7FFFFFFFu) > 0x7F800000u ? 5 : 7;
  if (u < 0x00800000u) return 1;
  if (u < 0x0F800000u) return 2;
  if (u < 0x7F800000u) return 3;
  if (u == 0x7F800000u) return 4;
  return 5;
}

__global__ void check(unsigned long long* differ, unsigned long long* nan_only, uint32_t* example) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += stride) {
    const uint32_t u = (uint32_t)v;
    float x = __uint_as_float(u);
    asm volatile("" : "+v"(x));
    const float want = __builtin_sqrtf(x), got = sqrt_short(x);
    const uint32_t bw = __float_as_uint(want), bg = __float_as_uint(got);
    if (bw != bg) {
      const int k = kind_of(u);
      if (want != want && got != got) {
        atomicAdd(&nan_only[k], 1ull);   // both NaN, different bits
      } else {
        atomicAdd(&differ[k], 1ull);
        example[k] = u;
      }
    }
  }
}

int main() {
  unsigned long long *differ, *nan_only;
  uint32_t* example;
  CK(hipMalloc(&differ, 8 * sizeof *differ));
  CK(hipMalloc(&nan_only, 8 * sizeof *nan_only));
  CK(hipMalloc(&example, 8 * sizeof *example));
  CK(hipMemset(differ, 0, 8 * sizeof *differ));
  CK(hipMemset(nan_only, 0, 8 * sizeof *nan_only));
  CK(hipMemset(example, 0, 8 * sizeof *example));
  check<<<4096, 256>>>(differ, nan_only, example);
  CK(hipDeviceSynchronize());
  unsigned long long d[8], n[8];
  uint32_t e[8];
  CK(hipMemcpy(d, differ, sizeof d, hipMemcpyDeviceToHost));
  CK(hipMemcpy(n, nan_only, sizeof n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(e, example, sizeof e, hipMemcpyDeviceToHost));
  const char* names[8] = {"+0", "positive denormal", "normal below 2^-96", "2^-96 to the largest",
                          "+infinity", "NaN", "-0", "negative"};
  for (int k = 0; k < 8; ++k)
    printf("%-22s differ %llu (e.g. 0x%08x)  both NaN with other bits %llu\n", names[k], d[k], e[k], n[k]);
  return 0;
}
