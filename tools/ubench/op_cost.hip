// op_cost.hip - what a SIMD takes to issue one wave64 instruction of each kind, with 4 waves per
// SIMD and 16 independent chains per wave (so that neither latency nor dependences limit it):
// launch time x shader clock / (waves per SIMD x instructions per wave).  The vector ALU of gfx950
// runs FP32 fma/add/mul at 32 lanes a clock (about 2 cycles a wave64 instruction); this probe
// shows which of the other instructions the kernels are made of cost the same and which cost
// twice that or more.
// Build: hipcc -O3 --offload-arch=gfx950 op_cost.hip -o op_cost_bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long cycles, ticks; };

#define CHAINS 16
#define UNROLL 8
// BODY(i): one instruction on chain i; x[i] is the chain's register, y and z are loop-invariant
#define DEFINE_KERNEL(name, BODY)                                                        \
  __global__ __launch_bounds__(256) void k_##name(Stamp* st, unsigned* out, int iters, unsigned seed) { \
    unsigned x[CHAINS];                                                                   \
    for (int i = 0; i < CHAINS; ++i) x[i] = seed * 2654435761u + threadIdx.x * 40503u + i * 977u; \
    unsigned y = seed + 0x3f800000u + threadIdx.x, z = (seed | 1u) + 3u * threadIdx.x;   \
    asm volatile("" : "+v"(y), "+v"(z));                                                  \
    const unsigned long long t0 = wall_clock64(), c0 = clock64();                        \
    for (int it = 0; it < iters; ++it) {                                                  \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { BODY(i); }                  \
      }                                                                                   \
    }                                                                                     \
    const unsigned long long c1 = clock64(), t1 = wall_clock64();                        \
    unsigned s = 0;                                                                       \
    for (int i = 0; i < CHAINS; ++i) s ^= x[i];                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
    if ((threadIdx.x & 63) == 0) {                                                        \
      Stamp r; r.cycles = c1 - c0; r.ticks = t1 - t0;                                     \
      st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;                               \
    }                                                                                     \
  }

#define OP3(ins) asm volatile(ins " %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z))
#define OP2(ins) asm volatile(ins " %0, %0, %1" : "+v"(x[i]) : "v"(y))
#define OP2R(ins) asm volatile(ins " %0, %1, %0" : "+v"(x[i]) : "v"(y))
#define OP1(ins) asm volatile(ins " %0, %0" : "+v"(x[i]))

#define B_fma(i) OP3("v_fma_f32")
#define B_mul_f32(i) OP2("v_mul_f32")
#define B_add_f32(i) OP2("v_add_f32")
#define B_max_f32(i) OP2("v_max_f32")
#define B_max3_f32(i) OP3("v_max3_f32")
#define B_max_i32(i) OP2("v_max_i32")
#define B_max3_i32(i) OP3("v_max3_i32")
#define B_min3_u32(i) OP3("v_min3_u32")
#define B_add_u32(i) OP2("v_add_u32")
#define B_xor(i) OP2("v_xor_b32")
#define B_and(i) OP2("v_and_b32")
#define B_or3(i) OP3("v_or3_b32")
#define B_lshl(i) asm volatile("v_lshlrev_b32 %0, 13, %0" : "+v"(x[i]))
#define B_lshr(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[i]))
#define B_lshl_add(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(y))
#define B_alignbit(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[i]) : "v"(y))
#define B_bitop3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36" : "+v"(x[i]) : "v"(y), "v"(z))
#define B_cndmask(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y) : )
#define B_mov(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y))
#define B_mul_lo(i) OP2("v_mul_lo_u32")
#define B_mul_hi(i) OP2("v_mul_hi_u32")
#define B_mul_u24(i) OP2("v_mul_u32_u24")
#define B_mad_u24(i) OP3("v_mad_u32_u24")
#define B_cvt_f32_u32(i) OP1("v_cvt_f32_u32")
#define B_cvt_pkrtz(i) OP2("v_cvt_pkrtz_f16_f32")
#define B_rcp(i) OP1("v_rcp_f32")
#define B_sqrt(i) OP1("v_sqrt_f32")
#define B_rsq(i) OP1("v_rsq_f32")
#define B_cmp_i32(i) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(x[i]), "v"(y) : "vcc")
#define B_cmp_f32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(y) : "vcc")
#define B_mbcnt(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x[i]) : "v"(y))
#define B_bcnt(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y))
#define B_ffbl(i) OP1("v_ffbl_b32")
#define B_perm(i) OP3("v_perm_b32")
#define B_div_scale(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x[i]) : "v"(y) : "vcc")
#define B_div_fmas(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z) : )
#define B_div_fixup(i) OP3("v_div_fixup_f32")
#define B_pk_mul(i) { }
#define B_readlane(i) { }
#define B_sub_f32(i) OP2("v_sub_f32")
#define B_fmac(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z))
#define B_or(i) OP2("v_or_b32")
#define B_sub_u32(i) OP2("v_sub_u32")
#define B_add3(i) OP3("v_add3_u32")
#define B_xad(i) OP3("v_xad_u32")
#define B_and_or(i) OP3("v_and_or_b32")
#define B_lshl_or(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(y))
#define B_add_lshl(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(x[i]) : "v"(y))
#define B_lshl1(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]))
#define B_lshlv(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(y))
#define B_lshrv(i) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(y))
#define B_ashr(i) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x[i]))
#define B_not(i) OP1("v_not_b32")
#define B_bfe(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x[i]))
#define B_bfi(i) OP3("v_bfi_b32")
#define B_min_f32(i) OP2("v_min_f32")
#define B_med3_f32(i) OP3("v_med3_f32")
#define B_cndmask_s(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(y) : "s20", "s21")
#define B_cvt_u32_f32(i) OP1("v_cvt_u32_f32")
#define B_cvt_f32_f16(i) OP1("v_cvt_f32_f16")
#define B_cvt_f16_f32(i) OP1("v_cvt_f16_f32")
#define B_pk_fma(i) { }
#define B_fmamk(i) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x[i]) : "v"(y))
#define B_fmaak(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fbe77" : "+v"(x[i]) : "v"(y))
#define B_mul_lit(i) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x[i]))
#define B_mul_sgpr(i) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(x[i]) : : "s20")
#define B_add_sgpr(i) asm volatile("v_add_u32 %0, s20, %0" : "+v"(x[i]) : : "s20")
#define B_fma_neg(i) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z))
#define B_add_abs(i) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(x[i]) : "v"(y))
#define B_fract(i) OP1("v_fract_f32")
#define B_floor(i) OP1("v_floor_f32")
#define B_ldexp(i) OP2("v_ldexp_f32")
#define B_exp(i) OP1("v_exp_f32")
#define B_cmp_sgpr(i) asm volatile("v_cmp_lt_i32_e64 s[22:23], %0, %1" : : "v"(x[i]), "v"(y) : "s22", "s23")
#define B_cmpx(i) { }
#define B_readfirst(i) asm volatile("v_readfirstlane_b32 s24, %0" : : "v"(x[i]) : "s24")
#define B_and3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(x[i]) : "v"(y), "v"(z))
#define B_mul_i32_i24(i) OP2("v_mul_i32_i24")
#define B_sad(i) OP3("v_sad_u32")
#define B_lerp(i) OP3("v_lerp_u8")
#define B_add_co(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[i]) : "v"(y) : "vcc")
#define B_sub_f32_lit(i) asm volatile("v_subrev_f32 %0, 1.0, %0" : "+v"(x[i]))
#define B_dpp(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]))

#define B_fma_smov(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_mov_b64 vcc, 0" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc")
#define B_fma_br_not(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_mov_b64 vcc, 0\n\ts_cbranch_vccnz .Lbn%=\n.Lbn%=:" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc")
#define B_fma_br_taken(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_mov_b64 vcc, 0\n\ts_cbranch_vccz .Lbt%=\n\ts_nop 0\n.Lbt%=:" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc")
#define B_fma_br_execz(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_cbranch_execz .Lbe%=\n.Lbe%=:" : "+v"(x[i]) : "v"(y), "v"(z))
#define B_fma_saveexec(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_mov_b64 vcc, exec\n\ts_and_saveexec_b64 s[22:23], vcc\n\ts_mov_b64 exec, s[22:23]" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc", "s22", "s23", "scc")
#define B_fma_nop(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(x[i]) : "v"(y), "v"(z))
#define B_fma_waitcnt(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(x[i]) : "v"(y), "v"(z))

#define LIST(X) X(fma) X(fmac) X(mul_f32) X(add_f32) X(sub_f32) X(max_f32) X(max3_f32) X(max_i32) X(max3_i32) X(min3_u32) \
  X(add_u32) X(xor) X(and) X(or3) X(lshl) X(lshr) X(lshl_add) X(alignbit) X(bitop3) X(cndmask) X(mov) \
  X(mul_lo) X(mul_hi) X(mul_u24) X(mad_u24) X(cvt_f32_u32) X(cvt_pkrtz) X(rcp) X(sqrt) X(rsq) \
  X(cmp_i32) X(cmp_f32) X(mbcnt) X(bcnt) X(ffbl) X(perm) X(div_scale) X(div_fmas) X(div_fixup) X(dpp) \
  X(or) X(sub_u32) X(add3) X(xad) X(and_or) X(lshl_or) X(add_lshl) X(lshl1) X(lshlv) X(lshrv) X(ashr) X(not) X(bfe) X(bfi) \
  X(min_f32) X(med3_f32) X(cndmask_s) X(cvt_u32_f32) X(cvt_f32_f16) X(cvt_f16_f32) X(fmamk) X(fmaak) X(mul_lit) X(mul_sgpr) \
  X(add_sgpr) X(fma_neg) X(add_abs) X(fract) X(floor) X(ldexp) X(exp) X(cmp_sgpr) X(readfirst) X(and3) X(mul_i32_i24) X(sad) \
  X(add_co) X(sub_f32_lit) X(fma_smov) X(fma_br_not) X(fma_br_taken) X(fma_br_execz) X(fma_saveexec) X(fma_nop) X(fma_waitcnt)

#define DEF(n) DEFINE_KERNEL(n, B_##n)
LIST(DEF)

typedef void (*kern_t)(Stamp*, unsigned*, int, unsigned);
struct Entry { const char* name; kern_t k; };
#define ENT(n) {#n, k_##n},
static Entry g_entries[] = { LIST(ENT) };

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
  Stamp* st; unsigned* out;
  CK(hipMalloc(&st, sizeof(Stamp) * blocks * 4));
  CK(hipMalloc(&out, sizeof(unsigned) * blocks * 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("CUs %d, %d waves per SIMD, %d instructions per wave\n", cus, waves_per_simd, iters * UNROLL * CHAINS);
  for (const Entry& e : g_entries) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, st, out, 10, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, st, out, iters, 1u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Stamp h[64];
    CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int i = 0; i < 64; ++i) ghz += (double)h[i].cycles / (double)h[i].ticks * 0.1;
    ghz /= 64;
    const double per = ms * 1e-3 * ghz * 1e9 / ((double)waves_per_simd * iters * UNROLL * CHAINS);
    printf("%-12s %7.3f ms at %.2f GHz: %5.2f cycles per instruction per SIMD\n", e.name, ms, ghz, per);
  }
  return 0;
}
