// rt_kernels.hip - hand-written gfx950 (CDNA4) kernels of the path tracer.
//
// What runs here is the reference's per-tile pixel/sample loop, execute_job
// (src/cpu/job.rs:36-74), with everything it calls: Camera::get_ray (src/main.rs:70-83),
// World::shoot_ray (src/main.rs:191-263), World::hit / Sphere::hit (src/main.rs:128-189),
// reflect / refract / reflectance (src/main.rs:162-166, 312-323) and the random helpers
// (src/main.rs:266-286).
//
// Shape of the main kernel (trace_kernel):
//  * Persistent waves.  A lane owns kRaysPerLane independent paths.  A work item is
//    (pixel, slice of that pixel's samples); a lane runs its item's samples one after the other
//    and keeps the running sum in registers, so the per-pixel sum is formed in sample order like
//    `pixel_color += color` (src/cpu/job.rs:58).  When a path ends the lane starts the next
//    sample, and when the item ends it takes the next item of its wave's reservation (one
//    global atomic per 64 items).  Lanes therefore never wait for the longest path of their
//    wave: at max_depth 50 this is what keeps the vector ALU busy.
//  * The recursion of shoot_ray is a loop with a running throughput; random numbers are drawn
//    in the reference's order from the counter stream of rt_rng.h.
//  * World::hit is split in two.  The scan evaluates, for every sphere, only the sign of the
//    discriminant half_b^2 - a*c of src/main.rs:131-135 (17 flops as written there).  It does so
//    on an expanded, pre-normalised and halved form: with hb' = (o.dn - c.dn) / sqrt 2 and
//    qh = c.o - k'/2, the sphere is a candidate unless hb'^2 + qh < |o|^2 / 2.  That needs no
//    o - c subtraction and leaves no per-ray term to add: 7 fused VALU instructions and one
//    compare per ray/sphere pair.  The sphere record (cx,cy,cz,-k'/2) is one wave-uniform
//    16-byte LDS read shared by both paths of every lane.  k' and |o|^2 carry a slack that
//    bounds the rounding of both forms, so the scan never skips a sphere the reference's own
//    arithmetic would accept (derivation: trigger_constant in rt_api.cpp).  A sphere that passes
//    is appended to the wave's candidate list in LDS.  After the scan the list is replayed
//    through the reference's exact Sphere::hit arithmetic (IEEE sqrt and divide, no
//    contraction), its entries dealt evenly to the 64 lanes whichever path they belong to; the
//    closest hit of a path is the minimum of (t, later sphere first) keys, which is the decision
//    World::hit's loop makes in storage order, so hit and t are bit-for-bit the reference's.
//  * Scenes larger than kChunkSpheres are streamed through the same LDS buffer chunk by chunk
//    by the whole workgroup (MULTI = true), carrying closest-hit across chunks.
//  * When a wave has no work left and at most kTailPaths paths are still alive it stops scanning
//    and traces them one at a time with the spheres spread across its lanes (tail_hit), so the
//    deep paths that remain when a frame drains do not hold the GPU for long.
//  * STRICT = true makes the scan evaluate the discriminant literally, in the reference's
//    operation order without fused multiply-adds (17 instructions).  Both modes give the same
//    bits; STRICT exists to check the conservative scan against the literal one.
//  * ACCEL = true (opt-in, RT_FLAG_ACCEL) replaces scan and replay by a per-lane traversal of a
//    bounding-volume hierarchy built on the host (rt_bvh.cpp), one path per lane; everything
//    around World::hit is the same code.  Same bits again: see bvh_hit.
//
// Build flags that matter: -ffp-contract=off (every fma below is written out by hand),
// -fno-slp-vectorize (auto-packing into v_pk_*_f32 costs more shuffles than it saves here).
#include "rt_internal.h"

#include <float.h>

#include "../../include/rt_rng.h"

// -DRT_ASM_MARKS (the `asm` target of the Makefile) leaves a comment in the ISA at every phase
// boundary, so that instructions can be counted per phase (tools/phase_instructions.py).
#ifdef RT_ASM_MARKS
#define RT_MARK(name) asm volatile("; RTMARK " name)
#else
#define RT_MARK(name) do { } while (0)
#endif

#ifdef RT_DIAG_PHASES
#define RT_STAMP(acc, t0)                                                   \
  do {                                                                      \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
    (acc) += now_ - (t0);                                                   \
    (t0) = now_;                                                            \
  } while (0)
#else
#define RT_STAMP(acc, t0) do { } while (0)
#endif

namespace rt {
namespace {

// Wave priority by phase.  Refill, replay and shading are chains of dependent instructions; at
// equal priority each of them waits its turn behind the other waves' scan loops, which always
// have an instruction ready.  Letting those chains go first shortens them and leaves the scan
// the same issue slots (-0.8 % kernel time on the benchmark frame, -1.1 % on the glass scene;
// the opposite assignment costs +2.8 %).
#ifndef RT_PRIO_REST_LEVEL
#define RT_PRIO_REST_LEVEL 3
#endif
#ifndef RT_PRIO_MFMA_LEVEL
#define RT_PRIO_MFMA_LEVEL 0
#endif
#define RT_PRIO_SCAN() __builtin_amdgcn_s_setprio(0)
#define RT_PRIO_REST() __builtin_amdgcn_s_setprio(RT_PRIO_REST_LEVEL)
#define RT_PRIO_MFMA() __builtin_amdgcn_s_setprio(RT_PRIO_MFMA_LEVEL)

// The lane's number again, through a statement the compiler cannot look into or move: what is
// computed from the result (LDS addresses of the lane's places in the wave's lists, mostly) is
// computed where it is used, a few instructions, instead of once before the main loop - from
// where, the registers being full, it would be spilled to scratch memory and read back with a
// round trip to device memory in the middle of a step.
__device__ __forceinline__ uint32_t here(uint32_t lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}

struct PhaseClock {
  unsigned long long t0, refill, scan, replay, shade, steps;
  unsigned long long t_start, t_exhausted, tail_steps;
  unsigned long long drain_scan_steps, tail_cycles, compact_steps;
  unsigned long long cand, replay_iters;   // queued candidates (per lane), replay loop trips (per wave)
  unsigned long long trig_cycles, prep_cycles;   // diagnostic: inside the candidate code, building columns
  unsigned long long blocks, trig_blocks, trig_pairs;   // scan blocks, those that branched to the
                                                        // queueing code, (sphere, path) pairs queued there
  unsigned long long starved_steps, starved_idle;   // steps a wave could not reserve pixels for
                                                    // want of slots; path slots left idle in them
  unsigned long long merges, merged_pixels;         // calls of merge_ready, pixels they wrote
  unsigned long long at_exhaustion[4];              // refill, scan, replay, shade when the wave ran out of work
  unsigned long long real_start, real_exhausted;    // 100 MHz clock
};

#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4   // 4 waves per SIMD = 2 workgroups per CU: at most 128 VGPRs
#endif

constexpr int R = kRaysPerLane;
constexpr int U = kScanUnroll;
// Candidates of a wave: ONE list in LDS shared by the wave's 64 lanes and R path slots.  An
// entry names the path (slot = lane + 64 * r, in its low 7 bits when R is 2) and the sphere
// (index local to the chunk in LDS).  Behind the list: one 64-bit word per path with the best
// hit found so far, as a key that an integer minimum orders like World::hit does (smaller t
// first, the later sphere at equal t), and the paths' rays (origin, a / direction), published
// there for the replay, whose lanes test candidates of paths that other lanes own.
#ifndef RT_SCAN_FOLD
#define RT_SCAN_FOLD 1
#endif
constexpr uint32_t L = kListEntries;
constexpr uint32_t kSlots = (uint32_t)(R * kWave);
static_assert(R == 2, "an entry keeps the path slot in 7 bits: lane | r << 6");
static_assert(L >= (uint32_t)(U * R * kWave), "an empty list must take whatever one block can add");
// A list that holds more than this is replayed before the scan goes on.
constexpr uint32_t kListMark = L - (uint32_t)kWave;
constexpr size_t kWaveListBytes = (size_t)L * sizeof(uint32_t) + (size_t)kSlots * sizeof(unsigned long long) +
                                  (size_t)kSlots * 2 * sizeof(float4);
// ... then four words: the list's length while the matrix scan appends to it, "keys live", and
// two that the first two waves lend to the workgroup's work cursor (WgCursor); and, last, the
// wave's table of open pixels (SlotTable)
constexpr size_t kWaveLdsBytes = kWaveListBytes + 16 + kSlotTableBytes;
static_assert(kWaveLdsBytes % 16 == 0, "the next wave's list must stay 16-byte aligned");
constexpr unsigned long long kNoHit = ~0ull;
constexpr float kTMin = 0.001f;  // src/main.rs:196
constexpr float kTMax = FLT_MAX; // f32::MAX, src/main.rs:196

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 divs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// glam: (x*x') + (y*y') + (z*z'), left to right, no contraction
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// sqrt, correctly rounded, for the hot places: what the compiler makes of sqrtf without its first
// and last steps.  v_sqrt_f32 is within an ulp of the root, so the root rounded to nearest is that
// value s or a neighbour, and the signs of x - s- s and x - s+ s (one fused operation each, s-
// and s+ the neighbours) say which.  The compiler's expansion does the same between a scaling by
// 2^32 for x < 2^-96, whose residuals would underflow, and a fix-up for 0 and infinity: nine of
// its fifteen instructions.  Over all 2^32 inputs (tools/ubench/sqrt_exhaustive.hip) the nine
// give the bits of the fifteen except for 0 < |x| < 2^-96:
//  - sqrt_1my is for x = 1 - y or |1 - y| of a float y, which is 0, NaN or at least 2^-24 in
//    magnitude, and for sums of squares of multiples of 2^-23 (0 or at least 2^-46);
//  - sqrt_rn is for any x: a wave that holds an x inside that gap (or a NaN) takes the compiler's
//    expansion for all its lanes, and the others never see it.
#ifndef RT_FAST_SQRT
#define RT_FAST_SQRT 1
#endif
__device__ __forceinline__ float sqrt_1my(float x) {
  if (!RT_FAST_SQRT) return __builtin_sqrtf(x);
  float s = __builtin_amdgcn_sqrtf(x);
  const float below = __uint_as_float(__float_as_uint(s) - 1u);
  const float above = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_below = __builtin_fmaf(-below, s, x);
  const float r_above = __builtin_fmaf(-above, s, x);
  s = 0.0f >= r_below ? below : s;
  s = 0.0f < r_above ? above : s;
  return s;
}
__device__ __forceinline__ float sqrt_rn(float x) {
  float s = sqrt_1my(x);
  if (RT_FAST_SQRT == 1 && __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(x) >= 0x1p-96f)) != 0) {
    // (the empty statement keeps the compiler from computing this ahead of the branch, for every
    // wave: to it, the square root is one cheap operation)
    asm volatile("" : "+v"(x));
    s = __builtin_sqrtf(x);
  }
  return s;
}
// 1.0f / sqrt(x), both correctly rounded, as glam's normalize takes them.  For x from 2^-96 up to
// the largest float the root s lies in [2^-48, 2^64), where the compiler's expansion of 1.0f / s
// scales nothing and fixes nothing up: it is the hardware's reciprocal and the seven fused
// operations below (the eleven instructions without v_div_scale_f32 twice, and with plain fused
// operations for v_div_fmas_f32 and v_div_fixup_f32).  A wave that holds any other x takes the
// compiler's expansions.  The same bits for every float: rt_selftest_sqrt.
__device__ __forceinline__ float inv_sqrt_rn(float x) {
  if (RT_FAST_SQRT != 1) return 1.0f / sqrt_rn(x);
  const float s = sqrt_1my(x);
  float y = __builtin_amdgcn_rcpf(s);
  y = __builtin_fmaf(__builtin_fmaf(-s, y, 1.0f), y, y);
  float q = y;   // 1.0f * y
  q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), y, q);
  q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), y, q);
  if (__builtin_amdgcn_ballot_w64(__float_as_uint(x) - 0x0F800000u >= 0x70000000u) != 0) {
    asm volatile("" : "+v"(x));
    q = 1.0f / __builtin_sqrtf(x);
  }
  return q;
}
// glam: v * (1.0 / sqrt(dot(v, v)))
__device__ __forceinline__ V3 normalize(V3 a) {
  return scale(a, inv_sqrt_rn(dot(a, a)));
}

// The random stream of one path.  `draws` is only read by the per-ray test hook.
struct PathRng {
  rt_rng g;
  uint32_t draws;
};
__device__ __forceinline__ float next_f32(PathRng& r) {
  r.draws++;
  return rt_rng_next_f32(&r.g);
}

// `rand::random::<f32>() * 2.0 - 1.0` (src/main.rs:267-269) in one instruction after the
// conversion: with k the draw's 24 bits, k * 2^-24, that times 2 and that minus 1 are all exact
// (k * 2^-23 - 1 is a multiple of 2^-23 of magnitude at most 1), so one fused multiply-add of
// k, 2^-23 and -1 gives the same bits as the three roundings-that-do-not-round.
__device__ __forceinline__ float next_signed(PathRng& r) {
  r.draws++;
  return __builtin_fmaf((float)(rt_rng_next_u32(&r.g) >> 8), 1.1920928955078125e-07f, -1.0f);
}
// random_vec3, src/main.rs:266-271
__device__ __forceinline__ V3 random_vec3(PathRng& r) {
  float x = next_signed(r);
  float y = next_signed(r);
  float z = next_signed(r);
  return mk(x, y, z);
}
// normalize of a vector whose components are multiples of 2^-23 (random_vec3's): the sum of
// their squares is 0 or at least 2^-46
__device__ __forceinline__ V3 normalize_draw(V3 a) {
  float len = sqrt_1my(dot(a, a));
  float rcp = 1.0f / len;
  return scale(a, rcp);
}
// random_in_unit_sphere_vec3, src/main.rs:273-282
__device__ __forceinline__ V3 random_in_unit_sphere(PathRng& r) {
  for (;;) {
    V3 v = random_vec3(r);
    if (dot(v, v) >= 1.0f) continue;
    return v;
  }
}
// reflect, src/main.rs:312-315
__device__ __forceinline__ V3 reflect(V3 input, V3 normal) {
  float dn = 2.0f * dot(input, normal);
  return sub(input, scale(normal, dn));
}
// refract, src/main.rs:317-323
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float etai_over_etat) {
  float cos_theta = __builtin_fminf(dot(neg(uv), n), 1.0f);
  V3 out_perp = scale(add(uv, scale(n, cos_theta)), etai_over_etat);
  float k = -sqrt_1my(__builtin_fabsf(1.0f - dot(out_perp, out_perp)));
  return add(out_perp, scale(n, k));
}
// reflectance, src/main.rs:162-166; powf(x, 5.0) as three multiplies (within 2 ulp of libm).
// r0 = ((1 - ref_index) / (1 + ref_index))^2 comes with the sphere's record: ref_index is the
// material's ir or 1 / ir, so the host has formed both values, with the same operations.
__device__ __forceinline__ float reflectance(float cosine, float r0) {
  float x = 1.0f - cosine;
  float x2 = x * x;
  float x4 = x2 * x2;
  return r0 + (1.0f - r0) * (x4 * x);
}
// 1 / |d| as glam's normalize forms it, from a = dot(d, d) of the segment (src/main.rs:132 and
// the `v * (1.0 / sqrt(dot(v, v)))` of normalize share that dot product bit for bit).  The sky
// and the dielectric branch both normalise the ray direction; computing the factor once ahead
// of them takes the square root and the division out of both.
__device__ __forceinline__ float inv_length(float a) {
  return inv_sqrt_rn(a);
}
// sky, src/main.rs:256-262; inv_len = inv_length(dot(dir, dir))
__device__ __forceinline__ V3 sky(V3 dir, float inv_len) {
  V3 u = scale(dir, inv_len);
  float t = 0.5f * (u.y + 1.0f);
  float s = 1.0f - t;
  return mk(s * 1.0f + t * 0.5f, s * 1.0f + t * 0.7f, s * 1.0f + t * 1.0f);
}
// f32::clamp of Rust (NaN stays NaN), as used by compose_final_framebuffer, src/cpu/mod.rs:73-75
__device__ __forceinline__ float rust_clamp(float v, float lo, float hi) {
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}
__device__ __forceinline__ float resolve_channel(float sum, float inv_spp) {
  return rust_clamp(__builtin_sqrtf(sum * inv_spp), 0.0f, 0.999f); // src/cpu/mod.rs:67-75
}

// ---- exact Sphere::hit on one candidate, src/main.rs:128-150 -------------------------------
// Updates (closest, best) exactly like the body of World::hit's loop, src/main.rs:182-185.
__device__ __forceinline__ void exact_hit(float4 s, uint32_t index, V3 o, V3 d, float a,
                                          float& closest, uint32_t& best) {
  V3 oc = sub(o, mk(s.x, s.y, s.z));
  float half_b = dot(oc, d);
  float c = dot(oc, oc) - s.w;   // s.w = radius * radius
  float discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0f) return;
  float sqrtd = __builtin_sqrtf(discriminant);
  float root = (-half_b - sqrtd) / a;
  if (root < kTMin || closest < root) {
    root = (-half_b + sqrtd) / a;
    if (root < kTMin || closest < root) return;
  }
  closest = root;
  best = index;
}

// ---- the same decision for a candidate visited in ANY order ----------------------------------
// World::hit's loop keeps the minimum, over spheres, of f = (first root if >= t_min, else second
// root if >= t_min), a later sphere replacing an earlier one at equal f (the rejection is the
// strict `t_max < root`).  That is independent of the visiting order once the tie rule is
// explicit, so candidates can be visited last-to-first (the big spheres sit at the end of the
// reference's scenes, so `closest` tightens early) and most of them can be turned away before
// the correctly rounded sqrt and divisions by three tests that are safe in floating point:
//  (i)   half_b > 0 and c > 0: sqrtd <= half_b, so the second root is <= 0 < t_min;
//  (ii)  a lower bound of the first root's numerator already exceeds closest * a;
//  (iii) an upper bound of the second root's numerator is already below t_min * a.
// (ii) and (iii) use the 1-ulp hardware sqrt widened by 2^-20, and 2^-20 / 2^-19 margins on
// both sides, far more than the few 2^-24 roundings involved.  Requires a finite, non-zero
// ray (prep_segment's `degenerate` is false); the caller falls back to exact_hit otherwise.
struct ReplayBounds {
  float closest_a;   // >= closest * a   (infinite while nothing is hit)
  float tmin_a;      // <= t_min * a
};
__device__ __forceinline__ ReplayBounds replay_bounds(float closest, float a) {
  ReplayBounds b;
  b.closest_a = (closest * a) * 1.00000095367431640625f;   // 1 + 2^-20
  b.tmin_a = (kTMin * a) * 0.99999809265136718750f;        // 1 - 2^-19
  return b;
}
__device__ __forceinline__ void candidate_hit(float4 s, uint32_t index, V3 o, V3 d, float a,
                                              ReplayBounds& rb, float& closest, uint32_t& best) {
  V3 oc = sub(o, mk(s.x, s.y, s.z));
  float half_b = dot(oc, d);
  float c = dot(oc, oc) - s.w;   // s.w = radius * radius
  float discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0f) return;
  if (half_b > 0.0f && c > 0.0f) return;                                            // (i)
  const float s_hi = __builtin_amdgcn_sqrtf(discriminant) * 1.00000095367431640625f;
  if ((-half_b - s_hi) * 0.99999904632568359375f > rb.closest_a) return;            // (ii)
  if ((-half_b + s_hi) * 1.00000190734863281250f < rb.tmin_a) return;               // (iii)
  float sqrtd = sqrt_rn(discriminant);
  float root = (-half_b - sqrtd) / a;
  if (root < kTMin) {
    root = (-half_b + sqrtd) / a;
    if (root < kTMin) return;
  }
  if (root < closest || (root == closest && (best == kNone || index > best))) {
    closest = root;
    best = index;
    rb = replay_bounds(closest, a);
  }
}

// ---- one segment's scatter decision, the body of shoot_ray, src/main.rs:196-251 -------------
struct Bounce {
  uint32_t kind;
  uint32_t material;
  bool front;
  bool scattered;
  V3 point, normal;
  V3 dir;    // scattered direction
  V3 atten;  // factor on the next segment's radiance
};

__device__ __forceinline__ Bounce shade_hit(const SceneDev& sc, V3 o, V3 d, float inv_len, float t,
                                            uint32_t sphere, PathRng& rng) {
  Bounce b;
  // (the sphere's radius and its material's fields sit in one record per sphere: a hit costs one
  // round trip to memory, not one for the material's number and another for the material)
  const float4 g = sc.geom[sphere];
  const float4 m0 = sc.surf[2 * sphere];
  const float4 m1 = sc.surf[2 * sphere + 1];
  const float radius = m1.z;
  b.material = __float_as_uint(m1.w);
  const uint32_t flags = __float_as_uint(m0.w);
  const V3 color = mk(m0.x, m0.y, m0.z);

  b.point = add(o, scale(d, t));                                  // Ray::at
  V3 outward = divs(sub(b.point, mk(g.x, g.y, g.z)), radius);      // src/main.rs:154
  b.front = dot(d, outward) < 0.0f;                               // set_face_normal
  b.normal = b.front ? outward : neg(outward);
  b.scattered = true;

  // Both the metal and the diffuse branch start by drawing random_in_unit_sphere_vec3(); one
  // loop serves the lanes of either kind (same draws, same order, per path).
  V3 ball = mk(0.0f, 0.0f, 0.0f);
  if (!(flags & RT_MAT_DIELECTRIC)) ball = random_in_unit_sphere(rng);

  if (flags & RT_MAT_DIELECTRIC) {                                // src/main.rs:201-227
    b.kind = RT_KIND_DIELECTRIC;
    // (a dielectric scatters white and its colour is never read, src/main.rs:201-227: its place
    // in the record holds 1 / ir and the two r0 of reflectance, see upload in rt_api.cpp)
    float ir = m1.y;
    float ratio = b.front ? m0.x : ir;        // `if front_face { 1.0 / ir } else { ir }`
    float r0 = b.front ? m0.y : m0.z;
    V3 u = scale(d, inv_len);   // normalize(d)
    float cos_theta = __builtin_fminf(dot(neg(u), b.normal), 1.0f);
    float sin_theta = sqrt_1my(1.0f - cos_theta * cos_theta);
    bool cannot_refract = (ratio * sin_theta) > 1.0f;
    bool do_reflect = cannot_refract;
    if (!cannot_refract) do_reflect = reflectance(cos_theta, r0) > next_f32(rng);  // `||`
    b.dir = do_reflect ? reflect(u, b.normal) : refract(u, b.normal, ratio);
    b.atten = mk(1.0f, 1.0f, 1.0f);
  } else if (flags & RT_MAT_METALLIC) {                           // src/main.rs:228-242
    b.kind = RT_KIND_METAL;
    V3 reflected = reflect(d, b.normal);
    float fuzz = m1.x;   // `1.0 - metallic_strength`, formed on the host (upload in rt_api.cpp)
    b.dir = add(reflected, scale(ball, fuzz));
    if (dot(reflected, b.normal) > 0.0f) {
      b.atten = color;   // albedo applied twice, src/main.rs:199,236-237,242: the host's products
    } else {
      b.atten = mk(0.0f, 0.0f, 0.0f);
      b.scattered = false;
    }
  } else {                                                        // src/main.rs:243-251
    b.kind = RT_KIND_DIFFUSE;
    b.dir = add(b.normal, normalize_draw(ball));
    b.atten = color;
  }
  return b;
}

// Sphere records the LDS buffer holds: the whole scene, or one chunk, plus 2*U records of
// padding for the scan's read-ahead.  A multiple of 4, so that the radius^2 array behind the
// records stays 16-byte aligned.
__host__ __device__ constexpr uint32_t lds_capacity(uint32_t n_spheres) {
  const uint32_t cap = n_spheres > kChunkSpheres ? kChunkSpheres : n_spheres;
  return ((cap + 3u) & ~3u) + 2u * kScanUnroll;
}

// Workgroup-cooperative copy of spheres [base, base+cn) into LDS.
__device__ __forceinline__ void stage_chunk(const SceneDev& sc, uint32_t base, uint32_t cn,
                                            float4* sph, float* sr2) {
  for (uint32_t i = threadIdx.x; i < cn; i += blockDim.x) {
    sph[i] = sc.trig[base + i];
    sr2[i] = sc.geom[base + i].w;
  }
}

// ---- per-path state -------------------------------------------------------------------------
struct Path {
  // current segment
  float ox, oy, oz;
  float dx, dy, dz;
  float a;          // dir.length_squared(), src/main.rs:132
  // constants of the conservative scan for this segment (prep_segment)
  float ndx, ndy, ndz;   // -d / (|d| sqrt 2), lengthened by 2^-20
  float od;              // o . (d / |d|)
  float thr;             // |o|^2 (1 - 2^-17) / 2: a sphere is a candidate unless hb'^2 + qh < thr
  bool degenerate;       // zero / non-finite ray: every sphere goes to the literal arithmetic
  float closest;
  uint32_t best;
  bool alive;
  // path
  float tr, tg, tb; // throughput
  uint32_t depth;   // segments the path may still trace
  PathRng rng;
  // item
  bool has_item;
  float ar, ag, ab; // running sum of the item's samples
  float fx, fy;     // pixel coordinates as f32
  uint32_t pixel;   // rt_rng_pixel(seed, y * W + x): the hash that seeds the streams of the pixel's samples
  uint32_t out_index;
  uint32_t sample, sample_end;
};

// Constants of the conservative scan for the segment now in (o, d, a).
// LEAN: only the `degenerate` verdict (the matrix scan builds its own constants from (o, d, a)
// and calls the full form just before a tail step, the one place that wants these).
template <bool LEAN = false>
__device__ __forceinline__ void prep_segment(Path& P) {
  // The scan evaluates HALF the normalised discriminant, hb'^2 / 2 - c' / 2, as
  // (hb' / sqrt 2)^2 + qh compared with |o|^2 (1 - 2^-17) / 2, where qh = c.o - k'/2 comes
  // straight from the sphere record: seven fused instructions per (sphere, path) and no per-ray
  // term left to add.  So the direction is scaled by 1 / (|d| sqrt 2): v_rsq_f32 is good to
  // 1 ulp, and the factor below is (1 + 2^-20) / sqrt 2 rounded up, which keeps hb'^2 / 2 from
  // coming out low after the three roundings involved.
  const float s = __builtin_amdgcn_rsqf(P.a) * 0.7071074843406677f;
  const float dnx = P.dx * s, dny = P.dy * s, dnz = P.dz * s;
  P.ndx = -dnx; P.ndy = -dny; P.ndz = -dnz;
  P.od = __builtin_fmaf(P.oz, dnz, __builtin_fmaf(P.oy, dny, P.ox * dnx));
  const float oo = P.ox * P.ox + P.oy * P.oy + P.oz * P.oz;
  P.thr = (oo * 0.99999237060546875f) * 0.5f;   // 1 - 2^-17; the halving is exact
  // A direction of zero, infinite or NaN length, or an origin that is not finite: the scan's
  // algebra does not apply.  Such a ray is kept out of the candidate lists and takes the
  // literal loop over all spheres (which then does whatever the reference does with it); a
  // threshold of -inf makes every filter that uses these constants pass every sphere on.
  const float inf = __builtin_inff();
  P.degenerate = !(P.a > 0.0f) || !(P.a < inf) || !(oo < 1e37f) || !(__builtin_fabsf(P.od) < inf);
  if (P.degenerate || LEAN) {
    P.ndx = P.ndy = P.ndz = 0.0f;
    P.od = 0.0f;
    P.thr = -inf;
  }
}

// The wave's candidate list, hit keys and published rays (see L above).
struct WaveLists {
  uint32_t* list;               // [L]
  unsigned long long* keys;     // [kSlots]
  float4* rays;                 // [kSlots][2]: (ox, oy, oz, a), (dx, dy, dz, -)
  uint32_t* counter;            // [0] the list's length, valid inside scan_pass_mfma only
                                // [1] "keys live": not zero when the keys hold the paths' best hits
                                //     of this step so far (a replay without collection has run).
                                //     In LDS rather than a register so that the scan pass's
                                //     loop carries no state for its rare replay
  uint32_t count;               // entries in the list, wave-uniform
  uint32_t redo_from;           // first sphere of a block whose candidates did not all fit, or kNone
};

__device__ __forceinline__ unsigned long long hit_key(float t, uint32_t index) {
  return ((unsigned long long)__float_as_uint(t) << 32) | (0xFFFFFFFFu - index);
}

// NB spheres against the R paths of the lane.  `s[u]` = (cx, cy, cz, -k'/2), `r2[u]` = radius^2
// (only read when STRICT).  All NB*R values are evaluated first, as independent instruction
// chains; one compare each (`not below the path's threshold`, so that a NaN counts as a
// candidate, as it does in `if discriminant < 0.0 { return false }`) leaves the lanes with a
// candidate in NB*R scalar masks, which the scalar unit ORs together: ONE rarely-taken branch
// per block leads to the code that appends the candidates to the wave's list.  `ok[r]` is the
// mask of lanes whose path r may be listed (`okl[r]`: this lane's bit of it): alive, with a
// finite non-zero ray (other rays take the literal loop at the end of scan_chunk).
template <bool STRICT, int NB>
__device__ __forceinline__ void test_block(const float4 (&s)[NB], const float (&r2)[NB],
                                           uint32_t first_index, Path (&P)[R],
                                           const unsigned long long (&ok)[R], const bool (&okl)[R],
                                           WaveLists& wl, uint32_t lane, bool& flush,
                                           PhaseClock& clk) {
  float v[NB][R];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (STRICT) {   // src/main.rs:131-135, literally
        const float ocx = P[r].ox - s[u].x, ocy = P[r].oy - s[u].y, ocz = P[r].oz - s[u].z;
        const float half_b = (ocx * P[r].dx + ocy * P[r].dy) + ocz * P[r].dz;
        const float c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - r2[u];
        v[u][r] = half_b * half_b - P[r].a * c;
      } else {        // conservative expanded form, 7 instructions (see prep_segment)
        const float hb = __builtin_fmaf(s[u].z, P[r].ndz,
                         __builtin_fmaf(s[u].y, P[r].ndy, __builtin_fmaf(s[u].x, P[r].ndx, P[r].od)));
        // s.w = -k'/2, so qh = c.o - k'/2
        const float qh = __builtin_fmaf(s[u].z, P[r].oz,
                         __builtin_fmaf(s[u].y, P[r].oy, __builtin_fmaf(s[u].x, P[r].ox, s[u].w)));
        v[u][r] = __builtin_fmaf(hb, hb, qh);
      }
    }
  }
  // the literal form compares the discriminant with zero, as the reference does
  unsigned long long m[NB][R], hot[R];
  constexpr bool kFold = !STRICT && NB > 1 && RT_SCAN_FOLD != 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float thr = STRICT ? 0.0f : P[r].thr;
    hot[r] = 0;
    if (kFold) {
      // One compare per path and block: the largest of the block's values against the path's
      // threshold (the conservative form never yields a NaN for a ray the scan lists: its terms
      // are finite and the only infinity it can reach is +inf, from hb'^2).  The masks per
      // sphere are formed in the rare block that has a candidate.
      float top = v[0][r];
#pragma unroll
      for (int u = 1; u < NB; ++u) top = __builtin_fmaxf(top, v[u][r]);
      hot[r] = __builtin_amdgcn_ballot_w64(!(top < thr));
    } else {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        m[u][r] = __builtin_amdgcn_ballot_w64(!(v[u][r] < thr));
        hot[r] |= m[u][r];
      }
    }
    hot[r] &= ok[r];
  }
  unsigned long long any = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) any |= hot[r];
#ifdef RT_DIAG_PHASES
  clk.blocks++;
#endif
  if (__builtin_expect(any != 0, 0)) {
#ifdef RT_DIAG_PHASES
    clk.trig_blocks++;
#endif
    if (kFold) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < NB; ++u) m[u][r] = __builtin_amdgcn_ballot_w64(!(v[u][r] < P[r].thr));
    }
    // All further decisions are the scalar unit's, and only a pair that has a candidate (two
    // of the eight, typically, with a single lane each) costs vector instructions: a ballot
    // count gives every such lane its place behind the list's end.
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const unsigned long long mm = m[u][r] & ok[r];
        if (mm != 0) {
#ifdef RT_DIAG_PHASES
          clk.trig_pairs++;
#endif
          const uint32_t k = (uint32_t)__builtin_popcountll(mm);
          if (__builtin_expect(wl.count + k > L, 0)) {
            // no room: replay what the list holds, then scan this block again (an entry that
            // gets listed twice that way changes nothing: the minimum is idempotent)
            if (first_index < wl.redo_from) wl.redo_from = first_index;
          } else {
            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            // (first_index + u) << 7 | r << 6 | lane
            const bool listed = !(v[u][r] < (STRICT ? 0.0f : P[r].thr)) && okl[r];
            if (listed) wl.list[wl.count + at] = lane | (((first_index + u) * R + r) << 6);
            wl.count = __builtin_amdgcn_readfirstlane(wl.count + k);
          }
        }
      }
    }
    if (wl.count > kListMark || wl.redo_from != kNone) flush = true;
  }
}

// The exact test of one candidate for a path whose best hit so far is `closest`: the accepted
// root of Sphere::hit (src/main.rs:128-150) or nothing, by candidate_hit's arithmetic and
// filters.  Whether it replaces the path's hit is decided by the key minimum, which also
// implements the tie rule.
__device__ __forceinline__ bool candidate_root(float4 s, V3 o, V3 d, float a, float closest,
                                               float* root_out) {
  const ReplayBounds rb = replay_bounds(closest, a);
  V3 oc = sub(o, mk(s.x, s.y, s.z));
  float half_b = dot(oc, d);
  float c = dot(oc, oc) - s.w;   // s.w = radius * radius
  float discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0f) return false;
  if (half_b > 0.0f && c > 0.0f) return false;                                            // (i)
  const float s_hi = __builtin_amdgcn_sqrtf(discriminant) * 1.00000095367431640625f;
  if ((-half_b - s_hi) * 0.99999904632568359375f > rb.closest_a) return false;            // (ii)
  if ((-half_b + s_hi) * 1.00000190734863281250f < rb.tmin_a) return false;               // (iii)
  float sqrtd = sqrt_rn(discriminant);
  float root = (-half_b - sqrtd) / a;
  if (root < kTMin) {
    root = (-half_b + sqrtd) / a;
    if (root < kTMin) return false;
  }
  if (!(root <= closest)) return false;   // `t_max < root` rejects; NaN is never a hit
  *root_out = root;
  return true;
}

// Replay the wave's candidate list through the exact arithmetic.  The entries are dealt to the
// 64 lanes in turn, so every lane has work in every trip whichever paths the candidates belong
// to (a lane that replayed only its own candidates waited for the longest queue of its wave: a
// third of the lanes were busy).  A lane reads the ray of the entry's path from LDS, prunes
// against that path's best hit so far (read from its key; a stale value only prunes less) and
// folds an accepted root into the key with an LDS atomic minimum, which does not depend on the
// order of arrival.
// PERM: the records in LDS are (cx, cy, cz, r*r) in the matrix scan's own sphere order, and
// `index` maps a position in the chunk to the sphere's number in the World (which is what the
// tie rule and the shading want).
// COLLECT = false leaves the paths' best hits in the keys (wl.keys_live) instead of taking them
// back into registers, for a replay in the middle of a scan pass: the pass's loop then changes no
// path state, and the replay that ends the step starts from those keys and collects.
// COMPACT: the wave's live paths are numbered 0, 1, ... in the order (slot 0 of every lane, then
// slot 1), and the entries, the published rays and the keys go by those numbers (compact_slot);
// paths that are not alive publish and collect nothing.
__device__ __forceinline__ uint32_t compact_slot(const Path (&P)[R], int r) {
  const unsigned long long m0 = __builtin_amdgcn_ballot_w64(P[0].alive);
  if (r == 0) return __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
  const unsigned long long m1 = __builtin_amdgcn_ballot_w64(P[1].alive);
  return (uint32_t)__builtin_popcountll(m0) +
         __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
}

template <bool PERM = false, bool COLLECT = true, bool COMPACT = false>
__device__ __forceinline__ void replay_lists(Path (&P)[R], const float4* sph, const float* sr2,
                                             uint32_t chunk_base, WaveLists& wl, uint32_t lane,
                                             PhaseClock& clk, const uint32_t* index = nullptr,
                                             uint32_t limit = kNone) {
  const uint32_t n = wl.count;
  // wave-uniform: every lane reads the same word
  const bool keys_live = __builtin_amdgcn_readfirstlane((int)wl.counter[1]) != 0;
  if (n == 0 && !(COLLECT && keys_live)) return;
  // (formed here, where the whole wave is: compact_slot counts the live paths of all lanes)
  uint32_t where[R];
#pragma unroll
  for (int r = 0; r < R; ++r) where[r] = COMPACT ? compact_slot(P, r) : (uint32_t)r * kWave + lane;
  if (n != 0) {
#ifdef RT_DIAG_PHASES
    if (lane == 0) clk.cand += n;   // summed over lanes at the end
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t slot = where[r];
      if (!COMPACT || P[r].alive) {
        wl.rays[2 * slot] = make_float4(P[r].ox, P[r].oy, P[r].oz, P[r].a);
        wl.rays[2 * slot + 1] = make_float4(P[r].dx, P[r].dy, P[r].dz, 0.0f);
        if (!keys_live) wl.keys[slot] = P[r].best == kNone ? kNoHit : hit_key(P[r].closest, P[r].best);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t j = 0; j < n; j += kWave) {
#ifdef RT_DIAG_PHASES
      clk.replay_iters++;
#endif
      // (PERM: an entry can only name a position below `limit` - the rows beyond it pass no ray,
      // see mfma_row_never in rt_api.cpp - and is turned away here should it ever do otherwise,
      // before its position is used to read a record and a sphere number that do not exist)
      const uint32_t e = j + lane < n ? wl.list[j + lane] : 0u;
      if (j + lane < n && (!PERM || (e >> 7) < limit)) {
        const uint32_t slot = e & (kSlots - 1), li = e >> 7;
        const float4 ro = wl.rays[2 * slot], rd = wl.rays[2 * slot + 1];
        const unsigned long long now = wl.keys[slot];
        const float closest = now == kNoHit ? kTMax : __uint_as_float((uint32_t)(now >> 32));
        float4 rec = sph[li];
        if (!PERM) rec.w = sr2[li];
        float root;
        if (candidate_root(rec, mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), ro.w, closest, &root))
          atomicMin(&wl.keys[slot], hit_key(root, PERM ? index[li] : chunk_base + li));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (COLLECT) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (COMPACT && !P[r].alive) continue;
      const unsigned long long mine = wl.keys[where[r]];
      if (mine != kNoHit) {
        P[r].closest = __uint_as_float((uint32_t)(mine >> 32));
        P[r].best = 0xFFFFFFFFu - (uint32_t)mine;
      }
    }
    if (keys_live) wl.counter[1] = 0u;
  } else {
    wl.counter[1] = 1u;
  }
  wl.count = 0;
}

// Scan one LDS-resident chunk of `cn` spheres for all paths of the wave.
template <bool STRICT>
__device__ __forceinline__ void scan_chunk(Path (&P)[R], const float4* sph, const float* sr2,
                                           uint32_t cn, uint32_t chunk_base, WaveLists& wl,
                                           uint32_t lane, PhaseClock& clk) {
  // the lanes whose path r can be listed: alive, and with a finite non-zero ray
  unsigned long long ok[R];
  bool okl[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    okl[r] = P[r].alive && !P[r].degenerate;
    ok[r] = __builtin_amdgcn_ballot_w64(okl[r]);
  }
  uint32_t i = 0;
  while (i < cn) {
    bool flush = false;
    wl.redo_from = kNone;
    RT_MARK("scan");
    RT_PRIO_SCAN();
    // The records of the next block are read from LDS while the current one is evaluated,
    // so the read latency is off the critical path: two register sets, A and B, alternate
    // (the loop body is two blocks long so that no set is ever copied).  The buffer is padded
    // by 2*U records, so the read-ahead stays inside it; what it reads there is never used.
    float4 A[U], B[U];
    float rA[U], rB[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      A[u] = sph[i + u];
      rA[u] = STRICT ? sr2[i + u] : 0.0f;
    }
    for (; i + 2 * U <= cn && !flush; i += 2 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        B[u] = sph[i + U + u];
        rB[u] = STRICT ? sr2[i + U + u] : 0.0f;
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the read-ahead ahead of the arithmetic
      test_block<STRICT, U>(A, rA, i, P, ok, okl, wl, lane, flush, clk);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        A[u] = sph[i + 2 * U + u];
        rA[u] = STRICT ? sr2[i + 2 * U + u] : 0.0f;
      }
      __builtin_amdgcn_sched_barrier(0);
      test_block<STRICT, U>(B, rB, i + U, P, ok, okl, wl, lane, flush, clk);
    }
    if (!flush && i + U <= cn) {
      test_block<STRICT, U>(A, rA, i, P, ok, okl, wl, lane, flush, clk);
      i += U;
    }
    if (!flush) {
      for (; i < cn && !flush; ++i) {
        const float4 s1[1] = {sph[i]};
        const float r1[1] = {STRICT ? sr2[i] : 0.0f};
        test_block<STRICT, 1>(s1, r1, i, P, ok, okl, wl, lane, flush, clk);
      }
    }
    if (wl.redo_from != kNone) i = wl.redo_from;   // a block that did not fit is scanned again
    RT_STAMP(clk.scan, clk.t0);
    RT_PRIO_REST();
    RT_MARK("replay");
    replay_lists(P, sph, sr2, chunk_base, wl, lane, clk);
    RT_MARK("replay_end");
    RT_STAMP(clk.replay, clk.t0);
  }
  // Zero or non-finite rays were kept out of the lists: World::hit's loop as written
  // (src/main.rs:175-189), first sphere of the chunk to last, which is the reference for any ray.
  bool odd = false;
#pragma unroll
  for (int r = 0; r < R; ++r) odd = odd || (P[r].alive && P[r].degenerate);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (P[r].alive && P[r].degenerate) {
        const V3 o = mk(P[r].ox, P[r].oy, P[r].oz), d = mk(P[r].dx, P[r].dy, P[r].dz);
        for (uint32_t k = 0; k < cn; ++k) {
          float4 rec = sph[k];
          rec.w = sr2[k];
          exact_hit(rec, chunk_base + k, o, d, P[r].a, P[r].closest, P[r].best);
        }
      }
    }
  }
}

// ---- the scan's filter on the matrix pipe ------------------------------------------------------
// The sign of the reference's discriminant (src/main.rs:131-135), divided by 2 |d|^2, is a bilinear
// form between ten numbers of the sphere and ten of the ray.  With n = d / |d|, p = o - (o.n) n
// (the part of the origin across the ray) and k = |c|^2 - r^2,
//     disc / (2 |d|^2) = 1/2 (c.n)^2 + c.p - k/2 - |p|^2 / 2
//   sphere S: cx^2 cy^2 cz^2 cxcy cxcz cycz cx cy cz k
//   ray    F: nx^2/2 ny^2/2 nz^2/2 nxny nxnz nynz px py pz -1/2         threshold |p|^2 / 2
// so 16 spheres x 16 rays are one v_mfma_f32_16x16x32_f16: every number is split into two f16
// halves (hi + lo, 21 bits or more together), a term S F becomes the three products
// Sh Fh + Sh Fl + Sl Fh, and the 32 slots of the instruction's K hold
//     0-8 Sh x Fh | 9-17 Sh x Fl | 18-26 Sl x Fh | 27, 28 k in two parts x -1/2 | 29-31: 2^15, 2^5,
//     2^-5 x the three parts of -threshold (scaled to stay inside f16 for origins out to 46340)
// A sphere is a candidate for a ray iff the product's entry is not negative.  Centres, radii and
// ray origins are scaled by a power of two chosen on the host (SceneDev::mfma_scale) so that
// the bulk of the scene sits below 128, where every feature is a normal f16 number; the few
// spheres beyond (the ground of the reference's scenes) are listed for every ray instead.
//
// Like the vector form, the filter must never drop a sphere that the reference's own arithmetic
// accepts.  In scaled units, with C = |c|, O = |o|, u = 2^-24: the reference's computed
// discriminant is >= 0 only if the exact form above is >= -22u (O^2 + C^2) - 11u r^2.  What is
// computed here differs from the exact form by at most
//     8.5u C^2 + 20u C O + 23u O^2      n from v_rsq_f32 and two roundings (8u), p and |p|^2 from it
//   + u C^2                             rounding of the ray's quadratic features
//   + 28u (C^2 + C O + r^2 / 2)         the halves: host side rounded to nearest (4u), ray side both
//                                       truncated (16u), lo.lo dropped (8u)
//   + 62u (C^2 + C O + (r^2 + O^2) / 2) accumulation inside the instruction, for ANY order of
//                                       summation.  The 32 products (f16 x f16: exact in f32) sum
//                                       to at most M = C^2 + C O + (r^2 + O^2) / 2 in magnitude
//                                       (1/2 (|cx nx| + |cy ny| + |cz nz|)^2 <= C^2 / 2, |c.p| <=
//                                       C O, |k| / 2, |p|^2 / 2).  ASSUMED of the pipe, whose
//                                       internal order and rounding are not documented: every one
//                                       of its 31 additions returns the exact sum of its operands
//                                       to within 2u relative, which holds for round-to-nearest
//                                       (u), for truncation (2u), for products aligned to the
//                                       largest one and cut to 24 bits or more, and for any wider
//                                       accumulator; then the total is within 31 * 2u * M whatever
//                                       the order.  Measured (tools/ubench/mfma_scan.hip, 8e5
//                                       products with heavy cancellation): 3.9u M at worst, and
//                                       f16 subnormals are honoured by the instruction
//   + 4u (C^2 + r^2)                    k's two parts
//   + u O^2                             the threshold's three parts
//   + 2^-21 max(C^2, C) + 2^-22 (1 + O) subnormal quanta (2^-25) of lo parts times the other side
// With C O <= (C^2 + O^2) / 2 that is below 158.5u C^2 + 110u O^2 + 49u r^2 plus the last line, on
// top of the reference's 22u / 22u / 11u: 180.5u C^2 + 132u O^2 + 60u r^2 in all.  The slack taken
// off is
//     sphere side (k, host): 2^-16 C^2 + 2^-17 r^2 + 2^-21 (C^2 + C)          (256u + 8u, 128u)
//     ray side (threshold):  (2^-16 + 2^-23) O^2 + 1.5 * 2^-22               (256u + 2u)
// which covers the sum with more than 80u of each to spare.  A false positive costs one exact
// replay, and the slack is a band 2^-16 of the discriminant's scale wide around the silhouette:
// doubling it from 2^-17 took the candidates per ray of the benchmark frame from 1.02 to 1.03.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr float kMfmaOriginLimit2 = 2147483648.0f;   // scaled |o|^2 below 2^31: the origin's
                                                     // features (|p| <= |o| < 46341) and the first
                                                     // part of the threshold (|o|^2 / 2^16) stay
                                                     // inside f16

// A ray the matrix filter cannot take (scaled origin beyond the limit) joins the zero and
// non-finite ones: kept out of the lists, literal loop over all spheres.
__device__ __forceinline__ void mfma_guard(Path& P, float scale) {
  const float ox = P.ox * scale, oy = P.oy * scale, oz = P.oz * scale;
  const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
  if (!(oo < kMfmaOriginLimit2) && !P.degenerate) {
    P.degenerate = true;
    P.ndx = P.ndy = P.ndz = 0.0f;
    P.od = 0.0f;
    P.thr = -__builtin_inff();
  }
}

// The ray's column: 32 f16 in four quarters of 8 slots, built as 16 dwords.  The hi halves are
// converted ONCE (two at a time, truncating: any rounding serves, the lo half takes what is left)
// and both copies of a hi half, as well as the value its lo half is measured against, come from
// those same bits.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bits(float lo_half, float hi_half) {
  const fp16x2 v = __builtin_amdgcn_cvt_pkrtz(lo_half, hi_half);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float half_lo(uint32_t w) {
  return (float)__builtin_bit_cast(h2, w)[0];
}
__device__ __forceinline__ float half_hi(uint32_t w) {
  return (float)__builtin_bit_cast(h2, w)[1];
}
__device__ __forceinline__ void mfma_ray_column(const Path& P, bool ok, float scale, h8 (&q)[4]) {
  uint32_t d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = 0u;
  d[14] = 0xBC000000u;   // a path that is not scanned: slot 29 = -1, i.e. -2^15 against every row
  if (ok) {
    const float s = __builtin_amdgcn_rsqf(P.a);
    const float nx = P.dx * s, ny = P.dy * s, nz = P.dz * s;
    const float ox = P.ox * scale, oy = P.oy * scale, oz = P.oz * scale;
    const float od = __builtin_fmaf(oz, nz, __builtin_fmaf(oy, ny, ox * nx));
    const float px = __builtin_fmaf(-od, nx, ox), py = __builtin_fmaf(-od, ny, oy),
                pz = __builtin_fmaf(-od, nz, oz);
    const float F[9] = {0.5f * (nx * nx), 0.5f * (ny * ny), 0.5f * (nz * nz),
                        nx * ny, nx * nz, ny * nz, px, py, pz};
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float pp = __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px));
    // (2^-16 + 2^-23) O^2 + 1.5 * 2^-22 off the threshold
    const float thr = 0.5f * pp - __builtin_fmaf(oo, 1.5377998352050781e-05f, 3.5762786865234375e-07f);
    const uint32_t h01 = pk_bits(F[0], F[1]), h23 = pk_bits(F[2], F[3]), h45 = pk_bits(F[4], F[5]),
                   h67 = pk_bits(F[6], F[7]), h8x = pk_bits(F[8], 0.0f);
    // what the hi halves leave (each subtraction is exact)
    const float l0 = F[0] - half_lo(h01), l1 = F[1] - half_hi(h01), l2 = F[2] - half_lo(h23),
                l3 = F[3] - half_hi(h23), l4 = F[4] - half_lo(h45), l5 = F[5] - half_hi(h45),
                l6 = F[6] - half_lo(h67), l7 = F[7] - half_hi(h67), l8 = F[8] - half_lo(h8x);
    const uint32_t minus_half = 0xB800u;   // -0.5
    // slots 0-8 Fh | 9-17 Fl | 18-26 Fh | 27, 28 -1/2 | 29-31 the threshold
    d[0] = h01; d[1] = h23; d[2] = h45; d[3] = h67;
    d[4] = (h8x & 0xFFFFu) | (pk_bits(l0, 0.0f) << 16);
    d[5] = pk_bits(l1, l2); d[6] = pk_bits(l3, l4); d[7] = pk_bits(l5, l6); d[8] = pk_bits(l7, l8);
    d[9] = h01; d[10] = h23; d[11] = h45; d[12] = h67;
    d[13] = (h8x & 0xFFFFu) | (minus_half << 16);
    // -threshold in three parts, against 2^15, 2^5 and 2^-5 on the sphere's side: each part takes
    // ten more bits of what the one before left (the subtractions are exact)
    const uint32_t t0 = pk_bits(thr * -3.0517578125e-05f, 0.0f) & 0xFFFFu;
    const float r1 = -thr - half_lo(t0) * 32768.0f;
    const uint32_t t1 = pk_bits(r1 * 0.03125f, 0.0f) & 0xFFFFu;
    const float r2 = r1 - half_lo(t1) * 32.0f;
    const uint32_t t2 = pk_bits(r2 * 32.0f, 0.0f) & 0xFFFFu;
    d[14] = minus_half | (t0 << 16);
    d[15] = t1 | (t2 << 16);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 w = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
    q[k] = __builtin_bit_cast(h8, w);
  }
}

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v);

// The columns of one half of the wave's paths (path slot r of every lane) as the instruction
// wants them: ray block j (slots 64 r + 16 j .. + 15) in b[j], lane l holding slots
// 8 (l >> 4) .. + 7 of ray l & 15.  Every lane builds the column of its own path and the wave
// passes them round through `stage`: 4 KB of its own LDS that nothing else uses at that moment
// (the published rays of the replay; the candidate list may hold entries already).
__device__ __forceinline__ void mfma_prepare_half(const Path& P, float scale, h8* stage, uint32_t lane,
                                                  h8 (&b)[4]) {
  h8 q[4];
  mfma_ray_column(P, P.alive && !P.degenerate, scale, q);
  lane = here(lane);   // (the eight swizzled addresses below were spilled and read back one by one)
  const uint32_t c = lane & 15, g = lane >> 4;
  // A column is 64 bytes, so lanes that follow each other would all start in the same two
  // banks: quarter k of ray l sits at place k ^ sw(l) of its column instead, with sw chosen so
  // that any 8 (or 16) consecutive rays spread one quarter over all 32 (or 64) banks, on the
  // way in (one ray per lane) and on the way out (16 consecutive rays per quarter).
  const uint32_t sw = ((c >> 1) + (c >> 3)) & 3u;
#pragma unroll
  for (int k = 0; k < 4; ++k) stage[lane * 4 + ((uint32_t)k ^ sw)] = q[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) b[jj] = stage[(16 * jj + c) * 4 + (g ^ sw)];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same for a wave that has no more work to wait for and at most 16 NB live paths: their
// columns go to the places compact_slot numbers them by, whichever lane and path slot they sit
// in, so that NB ray blocks (NB products per 16 spheres instead of eight) cover them all.  The
// places of the last block that no path takes get the column that passes no row.
template <int NB>
__device__ __forceinline__ void mfma_prepare_compact(const Path (&P)[R], float scale, h8* stage,
                                                     uint32_t lane, h8 (&b)[NB]) {
  const uint32_t c = lane & 15, g = lane >> 4;
  uint32_t where[R];
#pragma unroll
  for (int r = 0; r < R; ++r) where[r] = compact_slot(P, r);
  uint32_t live = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) live += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(P[r].alive));
  if (lane >= live && lane < 16u * NB) {
    h8 q[4];
    mfma_ray_column(P[0], false, scale, q);
    const uint32_t sw = ((c >> 1) + (c >> 3)) & 3u;
#pragma unroll
    for (int k = 0; k < 4; ++k) stage[lane * 4 + ((uint32_t)k ^ sw)] = q[k];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (P[r].alive) {
      h8 q[4];
      mfma_ray_column(P[r], true, scale, q);
      const uint32_t cc = where[r] & 15u;
      const uint32_t sw = ((cc >> 1) + (cc >> 3)) & 3u;
#pragma unroll
      for (int k = 0; k < 4; ++k) stage[where[r] * 4 + ((uint32_t)k ^ sw)] = q[k];
    }
  }
  const uint32_t sw = ((c >> 1) + (c >> 3)) & 3u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) b[jj] = stage[(16 * jj + c) * 4 + (g ^ sw)];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef RT_LEND
#define RT_LEND 1   // out of work, an item's last sample is lent to an idle path slot (lend_samples)
#endif
#ifndef RT_DRAIN_BLOCKS
#define RT_DRAIN_BLOCKS 4   // blocks of spheres per trip of the one-ray-block pass (scan_pass_mfma, SB)
#endif
#ifndef RT_COMPACT_TIERS
#define RT_COMPACT_TIERS 1   // one ray block for up to 16 live paths, RT_COMPACT_BLOCKS beyond
#endif
#ifndef RT_COMPACT_BLOCKS
#define RT_COMPACT_BLOCKS 4
#endif
constexpr int kCompactBlocks = RT_COMPACT_BLOCKS > 0 ? RT_COMPACT_BLOCKS : 1;
constexpr uint32_t kCompactPaths = 16u * RT_COMPACT_BLOCKS;   // 0: never

// What a chunk looks like in LDS for the matrix scan.
struct MfmaChunk {
  const h8* rows;          // [4][kMfmaRows]: quarter q of the sphere at position i at rows[q * kMfmaRows + i]
  const float4* geom;      // [kMfmaChunk] (cx, cy, cz, r*r)
  const uint32_t* index;   // [kMfmaChunk] sphere number in the World
  uint32_t count;          // spheres the chunk holds: positions beyond have rows that no ray passes
                           // and no record (the replay turns an entry that names one away)
};

// The spheres outside the filter's range (positions [n_rows, cn) of the chunk): one entry per
// scanned path.
__device__ __forceinline__ void mfma_list_always(Path (&P)[R], const MfmaChunk& m, uint32_t n_rows,
                                                 uint32_t cn, WaveLists& wl, uint32_t lane,
                                                 PhaseClock& clk) {
  for (uint32_t li = n_rows; li < cn; ++li) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool okl = P[r].alive && !P[r].degenerate;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(okl);
      const uint32_t k = (uint32_t)__builtin_popcountll(mm);
      if (k == 0) continue;
      if (wl.count + k > L) replay_lists<true>(P, m.geom, nullptr, 0, wl, lane, clk, m.index, m.count);
      const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
      if (okl) wl.list[wl.count + at] = lane | ((li * R + r) << 6);
      wl.count += k;
    }
  }
}

// The same spheres tested on the spot, every lane its own paths: what the list and the replay
// would do with one entry per scanned path, without the entry (the reference's scenes have one
// such sphere, the ground, which would otherwise be half of all entries).  The replay that
// follows starts from the hits found here.
__device__ __forceinline__ void mfma_test_always(Path (&P)[R], const MfmaChunk& m, uint32_t n_rows,
                                                 uint32_t cn) {
  for (uint32_t li = n_rows; li < cn; ++li) {
    const float4 rec = m.geom[li];
    const uint32_t world = m.index[li];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (P[r].alive && !P[r].degenerate) {
        ReplayBounds rb = replay_bounds(P[r].closest, P[r].a);
        candidate_hit(rec, world, mk(P[r].ox, P[r].oy, P[r].oz), mk(P[r].dx, P[r].dy, P[r].dz),
                      P[r].a, rb, P[r].closest, P[r].best);
      }
    }
  }
}

// One pass over the chunk's rows for one half of the wave's paths (path slot `half` of every
// lane; b = mfma_prepare_half's columns).  Sixteen results per lane at a time, for 16 spheres x
// 64 paths; thirty-two at a time (both halves) does not fit the register budget.
// NB < 4 with COMPACT: the wave's live paths, numbered by compact_slot, fill the first NB ray
// blocks (mfma_prepare_compact), `half` is 0 and an entry names a path by that number.
// SB > 1 (with NB = 1 and COMPACT): SB blocks of spheres per trip against the one ray block, their
// rows read a trip ahead.  A wave that is out of work takes its steps one after the other with
// little else on its SIMD to fill the gaps, and a trip of one product - wait for the rows, the
// product, the fold, the branch - is a chain of some 230 cycles, 31 times over; four products in
// a row and one fold and branch for the four make it eight chains of that length (a step of a
// lone wave with 8 paths: 5.07 -> 4.15 us).
template <int NB = 4, bool COMPACT = false, int SB = 1>
__device__ __forceinline__ void scan_pass_mfma(Path (&P)[R], const h8 (&b)[NB], uint32_t half,
                                               const MfmaChunk& m, uint32_t n_rows, WaveLists& wl,
                                               uint32_t lane, PhaseClock& clk) {
  static_assert(SB == 1 || (NB == 1 && COMPACT && SB <= 4), "several sphere blocks per trip: one ray block only");
  const uint32_t c = lane & 15, g = lane >> 4;
  RT_MARK("scan");
  RT_PRIO_MFMA();
  const f4v zero = {0.0f, 0.0f, 0.0f, 0.0f};
  __hip_atomic_store(wl.counter, wl.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  // (the rows behind the chunk's spheres, up to and including the block of padding, pass no ray:
  // a trip may run over the end of the spheres, and the read-ahead stops at the padding)
  h8 a[SB];
#pragma unroll
  for (int s = 0; s < SB; ++s) a[s] = m.rows[g * kMfmaRows + (uint32_t)s * kMfmaBlock + c];
  for (uint32_t blk = 0; blk < n_rows; blk += kMfmaBlock * SB) {
    f4v acc[NB * SB];
    if (SB == 1) {
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j], zero, 0, 0, 0);
      a[0] = m.rows[g * kMfmaRows + blk + kMfmaBlock + c];   // read ahead (the rows are padded by a block)
    } else {
#pragma unroll
      for (int s = 0; s < SB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[0], zero, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < SB; ++s) {
        const uint32_t ahead = blk + (uint32_t)(SB + s) * kMfmaBlock;
        a[s] = m.rows[g * kMfmaRows + (ahead < kMfmaChunk ? ahead : kMfmaChunk) + c];
      }
    }
    // Result i of ray block j is v[4 i + j]: sphere blk + 4 g + i against the path in slot
    // 64 half + 16 j + c (SB > 1: of sphere block j, sphere blk + 16 j + 4 g + i against path c).
    // A float that is not negative is an integer that is not negative: the maximum of the 16 as
    // integers says whether this lane has any candidate at all.
    constexpr int NJ = NB * SB;
    int v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 * i + j] = j < NJ ? __float_as_int(acc[j < NJ ? j : 0][i]) : -1;
    // Is any of the 16 not negative?  The bitwise AND of all of them has a clear sign bit exactly
    // then (the places of ray blocks that do not exist hold -1, all ones).  A three-input AND is
    // one v_bitop3_b32, which the vector ALU issues at the rate of an fma, twice that of the
    // v_max3_i32 this fold used to be made of (tools/ubench/op_cost.hip: 2.2 against 4.2 cycles).
    int top = v[0];
#pragma unroll
    for (int k = 1; k + 1 < 16; k += 2) top = top & v[k] & v[k + 1];
    top = top & v[15];
#ifdef RT_DIAG_PHASES
    clk.blocks++;
#endif
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(top >= 0) != 0, 0)) {
#ifdef RT_DIAG_PHASES
      clk.trig_blocks++;
      const unsigned long long trig_t0 = __builtin_amdgcn_s_memtime();
#endif
      // The sign of result (i, j) goes to bit 8 i + j of a mask: four short chains of
      // v_alignbit_b32 that do not wait for each other, and an entry of the list is then the
      // lane's base plus the bit's number times 16 (sphere position << 7 | path slot).
      uint32_t neg = 0;
#pragma unroll
      for (int i = 3; i >= 0; --i) {
        uint32_t ni = 0;
#pragma unroll
        for (int j = 3; j >= 0; --j) ni = __builtin_amdgcn_alignbit(ni, (uint32_t)v[4 * i + j], 31);
        neg = (neg << 8) | ni;
      }
      uint32_t cand = ~neg & 0x0F0F0F0Fu;
      // (the lane's own part of an entry: the compiler keeps it in a register across the pass)
      const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const uint32_t base = (blk << 7) | (half << 6) | ((ln >> 4) << 9) | (ln & 15u);
      // bit 8 i + j of the mask: sphere + i, and ray block j (16 path slots) or sphere block j
      const auto entry_offset = [](uint32_t q) {
        return SB == 1 ? q << 4 : ((q >> 3) << 7) | ((q & 7u) << 11);
      };
      // A lane with candidates reserves its run of the list with one LDS atomic on the list's
      // length (the order of the entries does not matter to the replay).
      uint32_t mine = (uint32_t)__builtin_popcount(cand);
      uint32_t at = 0;
      if (mine != 0)
        at = __hip_atomic_fetch_add(wl.counter, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifdef RT_DIAG_PHASES
      clk.trig_pairs += mine;   // per lane; summed over the wave at the end
#endif
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(at + mine > L) == 0, 1)) {
        uint32_t* dst = wl.list + at;
        while (cand != 0) {
          const uint32_t q = (uint32_t)__builtin_ctz(cand);
          cand &= cand - 1;
          *dst++ = base + entry_offset(q);
        }
      } else {
        // The list is full (kept apart from the usual case above, which then is straight code
        // with one loop: no flags and branches of an enclosing loop around it).  Runs that end
        // below its end are written whole, the one that straddles the end as far as it fits (so
        // every place up to the end holds an entry), the list is replayed, and what did not fit
        // (still set in `cand`) is reserved again.
        for (;;) {
          while (cand != 0 && at < L) {
            const uint32_t q = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1;
            wl.list[at] = base + entry_offset(q);
            ++at;
          }
          wl.count = L;
          RT_STAMP(clk.scan, clk.t0);
          RT_PRIO_REST();
          replay_lists<true, false, COMPACT>(P, m.geom, nullptr, 0, wl, lane, clk, m.index, m.count);
          RT_STAMP(clk.replay, clk.t0);
          RT_PRIO_MFMA();
          __hip_atomic_store(wl.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (__builtin_amdgcn_ballot_w64(cand != 0) == 0) break;
          mine = (uint32_t)__builtin_popcount(cand);
          at = 0;
          if (mine != 0)
            at = __hip_atomic_fetch_add(wl.counter, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifdef RT_DIAG_PHASES
          clk.trig_pairs += mine;
#endif
          if (__builtin_amdgcn_ballot_w64(at + mine > L) == 0) {
            uint32_t* dst = wl.list + at;
            while (cand != 0) {
              const uint32_t q = (uint32_t)__builtin_ctz(cand);
              cand &= cand - 1;
              *dst++ = base + entry_offset(q);
            }
            break;
          }
        }
      }
#ifdef RT_DIAG_PHASES
      clk.trig_cycles += __builtin_amdgcn_s_memtime() - trig_t0;
#endif
    }
  }
  // LDS operations of one wave complete in order: this read sees every reservation above
  wl.count = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)__hip_atomic_load(wl.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
  RT_STAMP(clk.scan, clk.t0);
  RT_PRIO_REST();
}

// Does any lane scan its path in slot r?
__device__ __forceinline__ bool mfma_half_scanned(const Path& P) {
  return __builtin_amdgcn_ballot_w64(P.alive && !P.degenerate) != 0;
}

// Zero, non-finite or out-of-range rays of the matrix scan: World::hit's loop as written
// (src/main.rs:175-189) over the World in its own order, from global memory.
__device__ __forceinline__ void literal_world_hit(Path (&P)[R], const SceneDev& sc) {
  bool odd = false;
#pragma unroll
  for (int r = 0; r < R; ++r) odd = odd || (P[r].alive && P[r].degenerate);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (P[r].alive && P[r].degenerate) {
        const V3 o = mk(P[r].ox, P[r].oy, P[r].oz), d = mk(P[r].dx, P[r].dy, P[r].dz);
        for (uint32_t k = 0; k < sc.n_spheres; ++k)
          exact_hit(sc.geom[k], k, o, d, P[r].a, P[r].closest, P[r].best);
      }
    }
  }
}

// Workgroup-cooperative copy of one chunk of the matrix scan's scene into LDS.
// `khalf` (the vector filter's -k'/2 per sphere, for tail steps) is filled when the whole scene is
// one chunk; a scene of several chunks takes its tail steps from global memory.
__device__ __forceinline__ void stage_chunk_mfma(const SceneDev& sc, uint32_t chunk, uint32_t cn,
                                                 uint4* rows, float4* geom, uint32_t* index,
                                                 float* khalf = nullptr) {
  const uint4* src = sc.mfma_rows + (size_t)chunk * 4 * kMfmaRows;
  for (uint32_t i = threadIdx.x; i < 4 * kMfmaRows; i += blockDim.x) rows[i] = src[i];
  for (uint32_t i = threadIdx.x; i < cn; i += blockDim.x) {
    geom[i] = sc.mfma_geom[chunk * kMfmaChunk + i];
    const uint32_t world = sc.mfma_index[chunk * kMfmaChunk + i];
    index[i] = world;
    if (khalf) khalf[i] = sc.trig[world].w;
  }
}

// World::hit for the last paths of a wave.  When a wave has no work left and at most kTailPaths
// live paths, scanning all 128 path slots against every sphere wastes most of its lanes, and
// what remains is a chain of dependent steps (a path at max_depth 50 inside glass) that holds
// the launch open.  Here the live paths are packed and the SPHERES are spread over lanes instead:
// with n live paths, the wave is cut into groups of g = 64 / pow2(n) lanes, one group per path;
// lane j of a group takes spheres j, j+g, ... first through the scan's conservative
// discriminant (7 fused instructions; the literal arithmetic when STRICT) and, for the few that
// pass, through the exact test; the group then takes the minimum t, the larger index winning a tie
// like a later sphere does in World::hit's loop.  A step costs n_spheres / g cheap tests per lane.
// The paths' rays travel to their groups, and the results back, through `xch`, the wave's own
// candidate-list memory (idle while nothing is scanned).
// IN_LDS: the whole scene is resident in the workgroup's LDS (records in `sph`, radius^2 in
// `sr2`); a chunked scene is read from global memory.
constexpr int kTailFloats = 12;   // ox oy oz dx dy dz a ndx ndy ndz od thr
static_assert((kTailFloats + 2) * kWave * sizeof(float) <= kWaveLdsBytes,
              "the tail exchange must fit the wave's candidate lists");
static_assert(kTailPaths <= (uint32_t)kWave, "one group of lanes per live path");

// PERM (with IN_LDS): the LDS holds the matrix scan's copy instead, spheres in its own order:
// `sph` = (cx, cy, cz, r*r), `sr2` = -k'/2, `index` = the sphere's number in the World (what the
// tie rule compares).
template <bool STRICT, bool IN_LDS, bool PERM = false>
__device__ __forceinline__ void tail_hit(Path (&P)[R], const SceneDev& sc, const float4* sph,
                                         const float* sr2, float* xch, uint32_t lane,
                                         const uint32_t* index = nullptr) {
  // pack the live paths: slot numbers 0..n-1, path slot 0 of every lane first
  uint32_t slot[R];
  uint32_t n = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(P[r].alive);
    slot[r] = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    n += __builtin_popcountll(m);
    if (P[r].alive) {
      float* f = xch + slot[r];
      f[0 * kWave] = P[r].ox; f[1 * kWave] = P[r].oy; f[2 * kWave] = P[r].oz;
      f[3 * kWave] = P[r].dx; f[4 * kWave] = P[r].dy; f[5 * kWave] = P[r].dz;
      f[6 * kWave] = P[r].a;
      f[7 * kWave] = P[r].ndx; f[8 * kWave] = P[r].ndy; f[9 * kWave] = P[r].ndz;
      f[10 * kWave] = P[r].od; f[11 * kWave] = P[r].thr;
    }
  }
  if (n == 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // g lanes per path, a power of two: 64 for one path ... 1 for 33 to 64 paths
  uint32_t shift = 0;                       // log2(g)
  while (shift < 6 && (n << (shift + 1)) <= (uint32_t)kWave) ++shift;
  const uint32_t g = 1u << shift;
  const uint32_t path = lane >> shift, sub = lane & (g - 1);
  float c = kTMax;
  uint32_t b = kNone;
  if (path < n) {
    const float* f = xch + path;
    const float ox = f[0 * kWave], oy = f[1 * kWave], oz = f[2 * kWave];
    const float dx = f[3 * kWave], dy = f[4 * kWave], dz = f[5 * kWave];
    const float a = f[6 * kWave];
    const float ndx = f[7 * kWave], ndy = f[8 * kWave], ndz = f[9 * kWave];
    const float od = f[10 * kWave], thr = f[11 * kWave];
    ReplayBounds rb = replay_bounds(c, a);
    // Four spheres per trip, their records read before any is looked at: a step of the last
    // paths is a chain of dependent instructions with nothing else on the SIMD to hide a read's
    // latency behind, and one read per trip made the loop wait for LDS 31 times over.
    constexpr int TU = 4;
    const uint32_t last = sc.n_spheres - 1;   // n_spheres > 0 here: a path is alive
    for (uint32_t i = sub; i < sc.n_spheres; i += TU * g) {
      float4 s[TU];
      float kh[TU];
      uint32_t at[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const uint32_t want = i + (uint32_t)u * g;
        at[u] = want < last ? want : last;   // past the end: the last sphere again (harmless)
        s[u] = IN_LDS ? sph[at[u]] : sc.trig[at[u]];
        kh[u] = PERM ? sr2[at[u]] : s[u].w;
      }
      bool pass[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        pass[u] = true;
        if (!STRICT) {
          // the same filter, with the same guarantee, as test_block's conservative form
          const float hb = __builtin_fmaf(s[u].z, ndz, __builtin_fmaf(s[u].y, ndy, __builtin_fmaf(s[u].x, ndx, od)));
          const float qh = __builtin_fmaf(s[u].z, oz, __builtin_fmaf(s[u].y, oy, __builtin_fmaf(s[u].x, ox, kh[u])));
          pass[u] = !(__builtin_fmaf(hb, hb, qh) < thr);
        }
      }
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        if (pass[u]) {
          const float4 geo = PERM ? s[u] : IN_LDS ? make_float4(s[u].x, s[u].y, s[u].z, sr2[at[u]]) : sc.geom[at[u]];
          candidate_hit(geo, PERM ? index[at[u]] : at[u], mk(ox, oy, oz), mk(dx, dy, dz), a, rb, c, b);
        }
      }
    }
  }
  // minimum over the lanes of each group (all 64 lanes take part; idle ones carry "no hit")
  for (uint32_t off = g >> 1; off > 0; off >>= 1) {
    const float oc = __shfl_xor(c, (int)off, kWave);
    const uint32_t ob = (uint32_t)__shfl_xor((int)b, (int)off, kWave);
    const bool take = ob != kNone && (b == kNone || oc < c || (oc == c && ob > b));
    if (take) { c = oc; b = ob; }
  }
  float* res = xch + kTailFloats * kWave;
  if (path < n && sub == 0) {
    res[path] = c;
    res[kWave + path] = __uint_as_float(b);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (P[r].alive) {
      P[r].closest = res[slot[r]];
      P[r].best = __float_as_uint(res[kWave + slot[r]]);
    }
  }
}

// ---- World::hit through the optional bounding-volume hierarchy (RT_FLAG_ACCEL) ----------------
// Every sphere the traversal reaches goes through candidate_hit, i.e. the reference's exact
// Sphere::hit arithmetic and World::hit's minimum-t / later-sphere-wins rule, which does not
// depend on the visiting order.  So the result is the linear scan's, bit for bit, as long as no
// sphere is pruned that the reference's arithmetic would accept.  A sphere (c, r) is accepted
// only if its COMPUTED discriminant is >= 0 and its computed root t lies in [t_min, closest].
// With u = 2^-24, S = |o-c|^2 + r^2, a = |d|^2, hb = (o-c).d, hb' its computed value and s the
// computed square root, the point at t = (-hb' +- s) / a satisfies, identically,
//   |o + t d - c|^2 = r^2 + (s^2 - disc_exact) / a + ((hb - hb')^2 +- 2 s (hb - hb')) / a.
// s^2 is the computed discriminant to within 2u of itself, which is within 22u a S of the exact
// one, and |hb - hb'| <= 4u |o-c| |d| with s <= |o-c| |d|: the right-hand side is at most
// r^2 + 32u S, so that point lies within r + 16u S / r of c (plus the rounding of t itself,
// below 8u (|o-c| + r)).  A box that holds the sphere, widened by that
// much, therefore holds the point at the computed root, and a slab test against it (with a
// relative 1e-5 on its own rounding) brackets that root.  The margin below is twice the bound,
// with |o-c| taken up to the distance to the ball that holds all centres and r between the
// tree's smallest and largest radius.
constexpr int kBvhStack = 24;   // balanced median splits: depth <= log2(n) <= 24

struct BvhRay {
  V3 o, d, inv;
  float a, margin;
};

__device__ __forceinline__ float bvh_margin(const BvhInfo& bi, V3 o) {
  const float u = 5.9604644775390625e-8f;   // 2^-24
  const V3 oc = sub(o, mk(bi.centre[0], bi.centre[1], bi.centre[2]));
  const float dist = __builtin_sqrtf(dot(oc, oc)) * 1.000001f + bi.extent;
  const float S = dist * dist + bi.r_max * bi.r_max;
  return 2.0f * (16.0f * u * S / bi.r_min + 8.0f * u * (dist + bi.r_max));
}

// Does the ray reach the widened box within [t_min, closest]?  *t_near is its entry parameter.
__device__ __forceinline__ bool bvh_box(const BvhRay& r, const float4 lo, const float4 hi,
                                        float closest, float* t_near) {
  const float ax = (lo.x - r.margin - r.o.x) * r.inv.x, bx = (hi.x + r.margin - r.o.x) * r.inv.x;
  const float ay = (lo.y - r.margin - r.o.y) * r.inv.y, by = (hi.y + r.margin - r.o.y) * r.inv.y;
  const float az = (lo.z - r.margin - r.o.z) * r.inv.z, bz = (hi.z + r.margin - r.o.z) * r.inv.z;
  // fminf/fmaxf drop a NaN operand (0 * inf when the origin sits on a slab of an axis-parallel
  // ray), which only removes a constraint
  float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
                             __builtin_fminf(az, bz));
  float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
                             __builtin_fmaxf(az, bz));
  tn = tn - 1e-5f * __builtin_fabsf(tn);
  tf = tf + 1e-5f * __builtin_fabsf(tf);
  *t_near = tn;
  // written so that a NaN (nothing constrains the ray) visits the box
  return !(tf < tn) && !(tf < kTMin * 0.999f) && !(tn > closest);
}

// Where the traversal reads nodes and leaf records: global memory, or copies in LDS when the
// whole hierarchy fits beside the stacks (IN_LDS; explicit LDS loads, because a load through
// the generic pointer would take the slow path).
struct BvhSource {
  const float4* nodes;        // two float4 per node
  const float4* leaf_geom;
  const uint32_t* leaf_index;
};
template <bool IN_LDS>
__device__ __forceinline__ float4 bvh_load4(const float4* p) {
  if (IN_LDS) {
    typedef float lds_f4 __attribute__((ext_vector_type(4)));
    const lds_f4 v = *(const __attribute__((address_space(3))) lds_f4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
template <bool IN_LDS>
__device__ __forceinline__ uint32_t bvh_load1(const uint32_t* p) {
  if (IN_LDS) return *(const __attribute__((address_space(3))) uint32_t*)p;
  return *p;
}

// `stack` is this thread's column of a [kBvhStack][kThreadsAccel] LDS array.
template <bool IN_LDS>
__device__ __forceinline__ void bvh_hit(Path& P, const SceneDev& sc, const BvhSource& src,
                                        uint32_t* stack, uint32_t& n_interior, uint32_t& n_tests) {
  if (!P.alive) return;
  const V3 o = mk(P.ox, P.oy, P.oz), d = mk(P.dx, P.dy, P.dz);
  float closest = kTMax;
  uint32_t best = kNone;
  if (__builtin_expect(P.degenerate, 0)) {
    // zero or non-finite ray: World::hit's loop as written, first sphere to last
    for (uint32_t i = 0; i < sc.n_spheres; ++i)
      exact_hit(sc.geom[i], i, o, d, P.a, closest, best);
    P.closest = closest;
    P.best = best;
    return;
  }
  ReplayBounds rb = replay_bounds(closest, P.a);
  const BvhInfo& bi = sc.bvh_info;
  for (uint32_t k = 0; k < bi.n_big; ++k)
    candidate_hit(bvh_load4<IN_LDS>(src.leaf_geom + k), bvh_load1<IN_LDS>(src.leaf_index + k), o, d,
                  P.a, rb, closest, best);
  if (bi.n_nodes != 0) {
    BvhRay r;
    r.o = o;
    r.d = d;
    r.a = P.a;
    r.inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.margin = bvh_margin(bi, o);
    const float4* nodes = src.nodes;
    int sp = 0;
    // (a, b) of the node in hand; its box was tested when it was chosen (the root's is not
    // tested at all: its children's are)
    uint32_t a = __float_as_uint(bvh_load4<IN_LDS>(nodes).w);
    uint32_t b = __float_as_uint(bvh_load4<IN_LDS>(nodes + 1).w);
    // Two nested loops rather than one that takes either kind of node: the lanes of a wave first
    // all walk down to a leaf (those that arrive early wait), then test their leaves together,
    // so a wave pays for the sphere tests once per leaf round and not once per node.
    const auto pop = [&]() {
      sp--;
      const uint32_t* words = reinterpret_cast<const uint32_t*>(nodes + 2 * stack[sp * kThreadsAccel]);
      a = bvh_load1<IN_LDS>(words + 3);
      b = bvh_load1<IN_LDS>(words + 7);
    };
    for (bool live = true; live;) {
      while (!(b & kBvhLeaf)) {
#ifdef RT_DIAG_PHASES
        n_interior++;
#endif
        const float4 l0 = bvh_load4<IN_LDS>(nodes + 2 * a), l1 = bvh_load4<IN_LDS>(nodes + 2 * a + 1);
        const float4 r0 = bvh_load4<IN_LDS>(nodes + 2 * b), r1 = bvh_load4<IN_LDS>(nodes + 2 * b + 1);
        float tl, tr;
        const bool hl = bvh_box(r, l0, l1, closest, &tl);
        const bool hr = bvh_box(r, r0, r1, closest, &tr);
        if (hl || hr) {
          const bool left = hl && (!hr || !(tr < tl));
          if (hl && hr) {
            stack[sp * kThreadsAccel] = left ? b : a;   // the farther child waits
            sp++;
          }
          a = __float_as_uint(left ? l0.w : r0.w);
          b = __float_as_uint(left ? l1.w : r1.w);
        } else if (sp == 0) {
          live = false;
          break;
        } else {
          pop();
        }
      }
      if (!live) break;
      const uint32_t count = b & ~kBvhLeaf;
#ifdef RT_DIAG_PHASES
      n_tests += count;
#endif
      for (uint32_t k = 0; k < count; ++k)
        candidate_hit(bvh_load4<IN_LDS>(src.leaf_geom + a + k),
                      bvh_load1<IN_LDS>(src.leaf_index + a + k), o, d, P.a, rb, closest, best);
      if (sp == 0) break;
      pop();
    }
  }
  P.closest = closest;
  P.best = best;
}

// ---- sliced samples: a wave's open pixels ------------------------------------------------------
// With more than one slice per pixel a wave reserves WHOLE pixels, so every slice of a pixel is
// run by one of its lanes.  A lane adds its slice's samples in registers, in sample order, parks
// the sum in the wave's ring (device memory, [slot][slice][3]) and counts it in; when a pixel's
// last slice is in, the pixel is marked ready, and the wave adds ready pixels' parked sums in
// slice order - one lane per (pixel, channel) - and writes them out (merge_ready).  The sum of a
// pixel is therefore a function of the frame alone: it does not depend on which lane ran which
// slice, on the order in which they finished, on tiling or on the number of GPUs.
struct SlotTable {
  uint32_t* out_at;   // [kRingSlots] where the open pixel goes in the call's output
  uint32_t* cnt;      // [kRingSlots / 4] slices that are in, one byte per slot
};
// The slots whose pixel has every slice in, as a wave-uniform mask: lane s looks at slot s's
// byte.  Called by all 64 lanes.
__device__ __forceinline__ uint32_t ready_slots(const SlotTable& st, uint32_t slices, uint32_t lane) {
  const unsigned char* bytes = reinterpret_cast<const unsigned char*>(st.cnt);
  const uint32_t mine = lane < kRingSlots ? bytes[lane] : 0u;
  return (uint32_t)__builtin_amdgcn_ballot_w64(mine == slices);
}
struct RingState {                 // wave-uniform
  uint32_t free_mask;              // slots that hold no pixel
  uint32_t map;                    // the current reservation's slots: byte j = the slot of its
                                   // j-th pixel counted from the LAST one
  bool starved;                    // a reservation was wanted and too few slots were free
  bool waiting;                    // a reservation was wanted while the workgroup's next chunk
                                   // was being fetched
};
// Where a wave's ring starts in p.ring, in floats.  Formed where it is used, from a thread index
// the compiler cannot see through, so that it is not kept in registers across the main loop.
__device__ __forceinline__ size_t ring_offset(const TraceParams& p) {
  uint32_t t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return (size_t)(blockIdx.x * (blockDim.x / kWave) + t / kWave) * ((size_t)p.ring_slots * 3u * p.slices);
}

// ---- the frame's work cursor, in two levels ---------------------------------------------------
// A round trip to the frame's cursor in device memory takes microseconds when every wave of the
// GPU goes there, and a wave's idle lanes wait for it.  So a WORKGROUP reserves a chunk of pixels
// there, keeps it as a range in LDS, and its waves reserve their few pixels at a time from that
// range with an LDS compare-and-swap.  Chunks shrink towards the end of the frame (a chunk's size
// follows from what was left when the one before it was made), so that the workgroups run out
// of work together.
struct WgCursor {
  unsigned long long* range;   // end << 32 | next, in pixels; all ones: the frame is exhausted
  uint32_t* lock;              // 1 while a wave is fetching the next chunk
  uint32_t* chunk;             // pixels the next chunk asks for
};
// Reserve up to `want` pixels for this wave.  Returns false when there is none to be had right
// now: the frame is exhausted (`exhausted` is set), or another wave of the workgroup is fetching
// the next chunk (ask again next step).  Called by all 64 lanes.
__device__ __forceinline__ bool wg_reserve(uint32_t want, uint32_t& first, uint32_t& np, bool& exhausted,
                                           const WgCursor& wg, const TraceParams& p, uint32_t lane) {
  uint32_t got_first = 0, got = 0;   // got: 0 = nothing now, kNone = exhausted
  if (lane == 0) {
    for (;;) {
      unsigned long long w = __hip_atomic_load(wg.range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t nx = (uint32_t)w, en = (uint32_t)(w >> 32);
      if (nx == kNone) {
        got = kNone;
        break;
      }
      if (nx < en) {
        const uint32_t take = en - nx < want ? en - nx : want;
        if (__hip_atomic_compare_exchange_strong(wg.range, &w, ((unsigned long long)en << 32) | (nx + take),
                                                 __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP)) {
          got_first = nx;
          got = take;
          break;
        }
        continue;   // another wave moved the range: look again
      }
      uint32_t unlocked = 0;
      if (!__hip_atomic_compare_exchange_strong(wg.lock, &unlocked, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP))
        break;      // another wave is fetching the next chunk
      // (the range may have been refilled between the look and the lock)
      w = __hip_atomic_load(wg.range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((uint32_t)w == kNone || (uint32_t)w < (uint32_t)(w >> 32)) {
        __hip_atomic_store(wg.lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        continue;
      }
      const uint32_t chunk = __hip_atomic_load(wg.chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t f = atomicAdd(&p.counters->next_pixel, chunk);
      if (f >= p.total_pixels) {
        __hip_atomic_store(wg.range, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(wg.lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        got = kNone;
        break;
      }
      const uint32_t left = p.total_pixels - f;
      const uint32_t c = left < chunk ? left : chunk;
      const uint32_t guided = (left - c) >> p.guide_shift;
      __hip_atomic_store(wg.chunk, guided > p.chunk_max ? p.chunk_max : guided < p.chunk_min ? p.chunk_min : guided,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(wg.range, ((unsigned long long)(f + c) << 32) | f, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(wg.lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  got_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)got_first);
  got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
  if (got == kNone) {
    exhausted = true;
    return false;
  }
  if (got == 0) return false;
  first = got_first;
  np = got;
  return true;
}

// Reserve work for the lanes of this wave whose `need` is set.  Returns the lane's item or
// kNone: the frame's items are exhausted, the wave holds too many pixels open to reserve more
// (rs.starved), or the workgroup's next chunk is on its way (rs.waiting).  `slot` is the ring
// slot of the item's pixel.  Must be called by all 64 lanes.
__device__ __forceinline__ uint32_t wave_fetch(bool need, uint32_t& next, uint32_t& end,
                                               bool& exhausted, RingState& rs, uint32_t& slot,
                                               const WgCursor& wg, const TraceParams& p, uint32_t lane) {
  uint32_t item = kNone;
  slot = 0;
  const bool sliced = p.slices != 1;
  unsigned long long mask = __builtin_amdgcn_ballot_w64(need);
  while (mask != 0 && !exhausted) {
    if (next == end) {
      if (sliced && (uint32_t)__builtin_popcount(rs.free_mask) < p.grab_pixels) {
        rs.starved = true;
        break;
      }
      uint32_t first = 0, np = 0;
      if (!wg_reserve(p.grab_pixels, first, np, exhausted, wg, p, lane)) {
        rs.waiting = !exhausted;
        break;
      }
      next = first * p.slices;   // below 2^32: checked on the host
      end = next + np * p.slices;
      if (sliced) {
        rs.map = 0;
        for (uint32_t i = 0; i < np; ++i) {
          const uint32_t s = (uint32_t)__builtin_ctz(rs.free_mask);
          rs.free_mask &= rs.free_mask - 1;
          rs.map = (rs.map << 8) | s;
        }
      }
    }
    const uint32_t avail = end - next;
    const bool asking = (mask >> lane) & 1ull;
    const uint32_t rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    const bool mine = asking && rank < avail;
    if (mine) {
      item = next + rank;
      if (sliced) {
        uint32_t j = 0;   // pixels of the reservation behind this item's
        if (p.grab_pixels != 1) j = fast_div(end - 1u - item, p.slices_div);
        slot = (rs.map >> (8 * j)) & 0xFFu;
      }
    }
    const uint32_t want = __builtin_popcountll(mask);
    next += want < avail ? want : avail;
    mask = __builtin_amdgcn_ballot_w64(asking && !mine);
  }
  return item;
}

// Pixel g of the call's tile list -> its tile (tiles are ordered by first_pixel).
__device__ __forceinline__ TileDev find_tile(uint32_t g, uint32_t n_tiles, const TileDev& tile0,
                                             const TileDev* tiles) {
  if (n_tiles == 1) return tile0;
  uint32_t lo = 0, hi = n_tiles;   // largest k with tiles[k].first_pixel <= g
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (tiles[mid].first_pixel <= g) lo = mid; else hi = mid;
  }
  return tiles[lo];
}

// Item numbering: pixel-major, item = pixel * slices + slice (the short slices that end a
// pixel's plan come last within the pixel).
__device__ __forceinline__ void setup_item(Path& P, uint32_t item, uint32_t slot, const TraceParams& p,
                                           const SlotTable& st) {
  uint32_t g = item, sl = 0;
  if (p.slices != 1) {
    g = fast_div(item, p.slices_div);
    sl = item - g * p.slices;
  }
  const TileDev t = find_tile(g, p.n_tiles, p.tile0, p.tiles);
  const uint32_t local = g - t.first_pixel;
  const uint32_t ty = fast_div(local, t.wdiv);
  const uint32_t tx = local - ty * t.w;
  uint32_t px = t.x + tx, py = t.y + ty;
  uint32_t at = (uint32_t)t.out_offset + ty * t.out_stride + tx;
  if (p.row_block != 0) {   // equal row blocks at a constant pitch, passed as one tile
    const uint32_t blk = fast_div(ty, p.row_block_div);
    const uint32_t row = ty - blk * p.row_block;
    py = t.y + blk * p.row_period + row;
    at = (uint32_t)t.out_offset + blk * (uint32_t)p.out_period + row * t.out_stride + tx;
  }
  P.fx = (float)px;
  P.fy = (float)py;
  P.pixel = rt_rng_pixel(p.seed_a, py * p.width + px);   // what the pixel's number is wanted for
  // Where the item's sum goes: the pixel's place in the output (below 2^32: checked on the
  // host), or, with sliced samples, entry `sl` of the pixel's slot in the wave's ring; the lane
  // that runs a pixel's first slice notes where the pixel goes.
  if (p.slices != 1) {
    P.out_index = (slot << 8) | sl;
    if (sl == 0) st.out_at[slot] = at;
  } else {
    P.out_index = at;
  }
  P.sample = p.slice_begin[sl];
  P.sample_end = p.slice_begin[sl + 1];
  P.ar = P.ag = P.ab = 0.0f;
  if (p.accumulate && p.slices == 1) {
    // progressive rendering with the whole pass in one item: carry on from the sum so far,
    // `pixel_color += color` (src/cpu/job.rs:58) continuing where the last pass stopped
    const float* o = p.out + 3ull * P.out_index;
    P.ar = o[0]; P.ag = o[1]; P.ab = o[2];
  }
  P.has_item = true;
}

// The sample loop body up to the first segment: jitter (src/cpu/job.rs:50-54) and
// Camera::get_ray (src/main.rs:70-83, Mat4*Vec4 = ((c0*x + c1*y) + c2*z) + c3*w).
template <bool LEAN = false>
__device__ __forceinline__ void start_sample(Path& P, const TraceParams& p) {
  P.rng.g = rt_rng_from_pixel(P.pixel, P.sample);
  P.rng.draws = 0;
  float u = div_by(P.fx + next_f32(P.rng), p.fwidth, p.inv_width);     // `/ image_width as f32`
  float v = div_by(P.fy + next_f32(P.rng), p.fheight, p.inv_height);   // `/ image_height as f32`
  u = u * 2.0f - 1.0f;
  v = v * 2.0f - 1.0f;
  const float* PI = p.cam.proj_inv;
  const float* VI = p.cam.view_inv;
  float d4[3], w[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    d4[i] = ((PI[i] * u + PI[4 + i] * v) + p.proj_z0[i]) + p.proj_w1[i];   // + c2 * 0.0 + c3 * 1.0
#pragma unroll
  for (int i = 0; i < 3; ++i)
    w[i] = ((VI[i] * d4[0] + VI[4 + i] * d4[1]) + VI[8 + i] * d4[2]) + p.view_w0[i];   // + c3 * 0.0
  const V3 dir = normalize(mk(w[0], w[1], w[2]));
  P.ox = p.origin[0]; P.oy = p.origin[1]; P.oz = p.origin[2];
  P.dx = dir.x; P.dy = dir.y; P.dz = dir.z;
  P.a = dot(dir, dir);
  prep_segment<LEAN>(P);
  P.tr = P.tg = P.tb = 1.0f;
  P.depth = p.max_depth;
  P.alive = true;
}

__device__ __forceinline__ void write_item(const Path& P, const TraceParams& p, const SlotTable& st) {
  float r = P.ar, g = P.ag, b = P.ab;
  if (p.slices != 1) {
    // park the slice's sum and count it in
    const uint32_t slot = P.out_index >> 8, sl = P.out_index & 0xFFu;
    float* o = p.ring + ring_offset(p) + slot * (3u * p.slices) + 3u * sl;
    o[0] = r; o[1] = g; o[2] = b;
    // (nothing waits for the count: the wave looks at its slots' counts now and then, ready_slots)
    (void)__hip_atomic_fetch_add(st.cnt + (slot >> 2), 1u << ((slot & 3u) * 8u), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WAVEFRONT);
    return;
  }
  if (p.resolve) {
    r = resolve_channel(r, p.inv_spp);
    g = resolve_channel(g, p.inv_spp);
    b = resolve_channel(b, p.inv_spp);
  }
  float* o = p.out + 3ull * P.out_index;
  o[0] = r; o[1] = g; o[2] = b;
}

// The wave's ready pixels: sum of the parked slice sums in slice order (`pixel_color += color`,
// src/cpu/job.rs:58, over slices instead of samples), then compose_final_framebuffer's arithmetic
// when asked for (src/cpu/mod.rs:67-75).  One lane per (pixel, channel), up to 21 pixels at a
// time.  Called by all 64 lanes; `ready` is wave-uniform and not zero.  `mode`: 1 = add to what
// the output holds, 2 = resolve.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void merge_pixels(uint32_t ready, const float* ring_wave, uint32_t slices,
                                             float* out, lds_u32* out_at, uint32_t mode, float inv_spp,
                                             uint32_t lane) {
  // the parked sums were stored by lanes of this wave: wait until the stores have left it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t group = lane / 3u, c = lane - 3u * group;
  lds_u32* cnt = out_at + kRingSlots;
  while (ready != 0) {
    uint32_t slot = kNone;
    for (uint32_t k = 0; k < 21u && ready != 0; ++k) {
      const uint32_t s = (uint32_t)__builtin_ctz(ready);
      ready &= ready - 1;
      if (group == k) slot = s;
    }
    if (slot != kNone) {
      // read through to the L2 (the lines of a slot were read before, for the pixel it held
      // last, and written since)
      const float* src = ring_wave + slot * (3u * slices) + c;
      const auto parked = [&](uint32_t sl) {
        return __hip_atomic_load(src + 3u * sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      float v = parked(0);
      uint32_t sl = 1;
      for (; sl + 8 <= slices; sl += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = parked(sl + (uint32_t)u);
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
      }
      for (; sl < slices; ++sl) v += parked(sl);
      float* dst = out + 3ull * out_at[slot] + c;
      if (mode & 1u) v = *dst + v;
      if (mode & 2u) v = resolve_channel(v, inv_spp);
      *dst = v;
      if (c == 0)
        __hip_atomic_fetch_and(cnt + (slot >> 2), ~(0xFFu << ((slot & 3u) * 8u)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
}
// The same as a function of its own: the registers the merge needs are then found around the
// call, in the rare step that merges, and not taken from the main loop (the vector-filter and
// traversal kernels: -3 % and -1.5 % kernel time; the matrix-filter kernel is 0.7 % faster with
// the merge inline).
__device__ __attribute__((noinline)) void merge_pixels_call(uint32_t ready, const float* ring_wave,
                                                            uint32_t slices, float* out, lds_u32* out_at,
                                                            uint32_t mode, float inv_spp) {
  merge_pixels(ready, ring_wave, slices, out, out_at, mode, inv_spp, threadIdx.x & (kWave - 1));
}
template <bool INLINE>
__device__ __forceinline__ void merge_ready(uint32_t ready, const TraceParams& p, const SlotTable& st,
                                            RingState& rs, uint32_t lane) {
  rs.free_mask |= ready;   // the slots go back to the free set
  const float* ring_wave = p.ring + ring_offset(p);
  const uint32_t mode = (p.accumulate ? 1u : 0u) | (p.resolve ? 2u : 0u);
  if (INLINE) merge_pixels(ready, ring_wave, p.slices, p.out, (lds_u32*)st.out_at, mode, p.inv_spp, lane);
  else merge_pixels_call(ready, ring_wave, p.slices, p.out, (lds_u32*)st.out_at, mode, p.inv_spp);
}

// ---- out of work: an item's last sample is run by an idle path slot of the wave ---------------
// When the frame's work has run out, what holds a wave (and with the last wave, the launch) is
// its longest chain of dependent steps, and an item of several samples is such a chain twice
// over: a path that runs max_depth segments, and then the item's next sample, which may do the
// same.  So once the wave is out of work, an item that still has a sample to come after the one
// it is running LENDS its last sample to a path slot of the wave that has nothing left to do.
// The borrower runs the sample from a sum of zero and holds the result; when the lender has
// finished the samples it kept, it adds the borrowed sample's sum - after its own, as the last
// term, which is where `pixel_color += color` (src/cpu/job.rs:58) adds that sample - and writes
// the item.  Same bits as without: 0 + x is x for every x but -0, which the sum turns into +0
// either way.  One loan per item at a time; the two sides know each other by path slot number
// (lane | slot << 6), kept in spare bits of out_index (sliced samples only: slot << 8 | slice).
constexpr uint32_t kLent = 0x80000000u;       // this item waits for a borrowed sample: bits 16-22 = the borrower
constexpr uint32_t kBorrowed = 0x40000000u;   // this slot runs a borrowed sample: bits 0-6 = the lender
__device__ __forceinline__ uint32_t rank_in(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Borrowed samples that are finished go back to lenders that are finished.  `xch`: 4 KB of the
// wave's LDS that nothing uses during the refill (the published rays of the replay).
__device__ __forceinline__ void return_loans(Path (&P)[R], uint32_t* xch, uint32_t lane) {
  unsigned long long ready[R], done[R];
  bool fin[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    fin[r] = P[r].has_item && !P[r].alive && P[r].sample >= P[r].sample_end;
    ready[r] = __builtin_amdgcn_ballot_w64(fin[r] && (P[r].out_index & kLent) != 0);
    done[r] = __builtin_amdgcn_ballot_w64(fin[r] && (P[r].out_index & kBorrowed) != 0);
  }
  if ((ready[0] | ready[1]) == 0 || (done[0] | done[1]) == 0) return;
  float* fx = reinterpret_cast<float*>(xch);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (fin[r] && (P[r].out_index & kBorrowed) != 0) {
      const uint32_t o = P[r].out_index & 127u;
      const unsigned long long m = (o & 64u) ? ready[1] : ready[0];
      if ((m >> (o & 63u)) & 1ull) {
        fx[4 * o] = P[r].ar; fx[4 * o + 1] = P[r].ag; fx[4 * o + 2] = P[r].ab;
        P[r].has_item = false;
        P[r].out_index = 0;
      }
    }
  }
  wave_sync_lds();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (fin[r] && (P[r].out_index & kLent) != 0) {
      const uint32_t b = (P[r].out_index >> 16) & 127u;
      const unsigned long long m = (b & 64u) ? done[1] : done[0];
      if ((m >> (b & 63u)) & 1ull) {
        const uint32_t self = lane | ((uint32_t)r << 6);
        P[r].ar += fx[4 * self]; P[r].ag += fx[4 * self + 1]; P[r].ab += fx[4 * self + 2];
        P[r].out_index &= 0xFFFFu;
      }
    }
  }
  wave_sync_lds();
}
// Items with a sample still to come after the one they are running lend it to idle path slots.
__device__ __forceinline__ void lend_samples(Path (&P)[R], uint32_t* xch, uint32_t lane) {
  bool don[R], tak[R];
  unsigned long long md[R], mt[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    don[r] = P[r].alive && P[r].has_item && (P[r].out_index >> 30) == 0 && P[r].sample + 1 < P[r].sample_end;
    tak[r] = !P[r].alive && !P[r].has_item;
    md[r] = __builtin_amdgcn_ballot_w64(don[r]);
    mt[r] = __builtin_amdgcn_ballot_w64(tak[r]);
  }
  const uint32_t nd = (uint32_t)(__builtin_popcountll(md[0]) + __builtin_popcountll(md[1]));
  const uint32_t nt = (uint32_t)(__builtin_popcountll(mt[0]) + __builtin_popcountll(mt[1]));
  const uint32_t n = nd < nt ? nd : nt;
  if (n == 0) return;
  uint32_t rd[R], rk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    rd[r] = (r ? (uint32_t)__builtin_popcountll(md[0]) : 0u) + rank_in(md[r]);
    rk[r] = (r ? (uint32_t)__builtin_popcountll(mt[0]) : 0u) + rank_in(mt[r]);
    const uint32_t self = lane | ((uint32_t)r << 6);
    if (don[r] && rd[r] < n) {
      uint32_t* rec = xch + 8 * rd[r];
      rec[0] = P[r].pixel;
      rec[1] = __float_as_uint(P[r].fx);
      rec[2] = __float_as_uint(P[r].fy);
      rec[3] = P[r].sample_end - 1u;
      rec[4] = self;
    }
    if (tak[r] && rk[r] < n) xch[8 * rk[r] + 5] = self;
  }
  wave_sync_lds();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (don[r] && rd[r] < n) {
      P[r].sample_end -= 1u;
      P[r].out_index |= kLent | (xch[8 * rd[r] + 5] << 16);
    }
    if (tak[r] && rk[r] < n) {
      const uint32_t* rec = xch + 8 * rk[r];
      P[r].pixel = rec[0];
      P[r].fx = __uint_as_float(rec[1]);
      P[r].fy = __uint_as_float(rec[2]);
      P[r].sample = rec[3];
      P[r].sample_end = rec[3] + 1u;
      P[r].out_index = kBorrowed | rec[4];
      P[r].ar = P[r].ag = P[r].ab = 0.0f;
      P[r].has_item = true;
    }
  }
  wave_sync_lds();
}

[[maybe_unused]] __device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
  unsigned long long s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
  return s;
}

// ACCEL (the hierarchy instead of the scan) runs ONE path per lane: two paths per lane exist to
// share the scan's sphere reads, and a traversal shares nothing, so the registers are better
// spent on twice the waves per SIMD to hide its dependent loads.
#ifndef RT_ACCEL_MIN_WAVES
#define RT_ACCEL_MIN_WAVES 8
#endif
#ifndef RT_ACCEL_WAVES
#define RT_ACCEL_WAVES RT_ACCEL_MIN_WAVES   // waves per SIMD the traversal kernel runs at
#endif
// MFMA (the scan's filter on the matrix pipe, see scan_chunk_mfma): MULTI then means more than
// kMfmaChunk spheres.
template <bool STRICT, bool MULTI, bool ACCEL = false, bool MFMA = false>
__global__ __launch_bounds__(trace_threads(MULTI, ACCEL, MFMA), ACCEL ? RT_ACCEL_MIN_WAVES : RT_MIN_WAVES)
void trace_kernel(const TraceParams p) {
  static_assert(!MFMA || (!STRICT && !ACCEL), "the matrix filter replaces the conservative vector scan only");
  constexpr int RP = ACCEL ? 1 : R;   // paths per lane
  extern __shared__ float4 lds[];
  const uint32_t n = p.scene.n_spheres;
  const uint32_t cap = lds_capacity(n);
  float4* sph = lds;                                       // (cx, cy, cz, -k'/2) per sphere
  float* sr2 = reinterpret_cast<float*>(lds + cap);        // radius^2 per sphere
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = threadIdx.x / kWave;
  // MFMA: the chunk's rows, exact records and sphere numbers instead
  uint4* mrows = reinterpret_cast<uint4*>(lds);
  float4* mgeom = lds + 4 * kMfmaRows;
  uint32_t* mindex = reinterpret_cast<uint32_t*>(mgeom + kMfmaChunk);
  float* mkhalf = reinterpret_cast<float*>(mindex + kMfmaChunk);
  MfmaChunk mc;
  mc.rows = reinterpret_cast<const h8*>(mrows);
  mc.geom = mgeom;
  mc.index = mindex;
  mc.count = 0;
  // this wave's candidate lists and hit keys, behind the sphere records
  unsigned char* wave_lds = (MFMA ? reinterpret_cast<unsigned char*>(mkhalf + kMfmaChunk)
                                  : reinterpret_cast<unsigned char*>(sr2 + cap)) + wave * kWaveLdsBytes;
  WaveLists wl;
  wl.list = reinterpret_cast<uint32_t*>(wave_lds);
  wl.keys = reinterpret_cast<unsigned long long*>(wave_lds + (size_t)L * sizeof(uint32_t));
  wl.rays = reinterpret_cast<float4*>(wave_lds + (size_t)L * sizeof(uint32_t) +
                                      (size_t)kSlots * sizeof(unsigned long long));
  wl.counter = reinterpret_cast<uint32_t*>(wave_lds + kWaveListBytes);
  wl.count = 0;
  if (!ACCEL) wl.counter[1] = 0u;   // (the traversal keeps no lists: its LDS holds stacks and the hierarchy)
  wl.redo_from = kNone;
  // the wave's open pixels (sliced samples): behind its lists, or, for the traversal, where the
  // host put the tables
  SlotTable st;
  {
    unsigned char* tab = ACCEL ? reinterpret_cast<unsigned char*>(lds) + p.slot_table_offset +
                                     wave * (kSlotTableBytes + 16)
                               : wave_lds + kWaveListBytes;
    st.out_at = reinterpret_cast<uint32_t*>(tab + 16);
    st.cnt = st.out_at + kRingSlots;
    if (lane < kRingSlots / 4) st.cnt[lane] = 0u;
  }
  const uint32_t all_slots = p.ring_slots >= 32u ? ~0u : (1u << p.ring_slots) - 1u;
  RingState rs;
  rs.free_mask = all_slots;
  rs.map = 0;
  rs.starved = false;
  rs.waiting = false;
  const bool sliced = p.slices != 1;
  // the workgroup's share of the frame's cursor: the spare words of its first two waves' headers
  // (a wave's four header words: list length, "keys live", two spare)
  WgCursor wg;
  {
    unsigned char* base = ACCEL ? reinterpret_cast<unsigned char*>(lds) + p.slot_table_offset
                                : wave_lds - wave * kWaveLdsBytes + kWaveListBytes;
    constexpr size_t stride = ACCEL ? kSlotTableBytes + 16 : kWaveLdsBytes;
    wg.range = reinterpret_cast<unsigned long long*>(base + 8);
    wg.lock = reinterpret_cast<uint32_t*>(base + stride + 8);
    wg.chunk = wg.lock + 1;
    if (threadIdx.x == 0) {
      *wg.range = 0ull;   // empty
      *wg.lock = 0u;
      *wg.chunk = p.chunk_max;
    }
  }
  __syncthreads();

  if (!MULTI && !ACCEL) {
    if (MFMA) stage_chunk_mfma(p.scene, 0, n, mrows, mgeom, mindex, mkhalf);
    else stage_chunk(p.scene, 0, n, sph, sr2);
    mc.count = n;
    __syncthreads();
  }
  // ACCEL: the LDS holds the traversal stacks instead, one column per thread, and behind them
  // a copy of the hierarchy when the host found room for it
  uint32_t* bvh_stack = reinterpret_cast<uint32_t*>(lds) + threadIdx.x;   // p.bvh_stack rows
  BvhSource bvh_src;
  bvh_src.nodes = reinterpret_cast<const float4*>(p.scene.bvh);
  bvh_src.leaf_geom = p.scene.leaf_geom;
  bvh_src.leaf_index = p.scene.leaf_index;
  if (ACCEL && p.bvh_in_lds) {
    float4* nodes_lds = lds + p.bvh_stack * kThreadsAccel / 4;
    float4* geom_lds = nodes_lds + 2 * p.scene.bvh_info.n_nodes;
    uint32_t* index_lds = reinterpret_cast<uint32_t*>(geom_lds + n);
    for (uint32_t i = threadIdx.x; i < 2 * p.scene.bvh_info.n_nodes; i += kThreadsAccel)
      nodes_lds[i] = bvh_src.nodes[i];
    for (uint32_t i = threadIdx.x; i < n; i += kThreadsAccel) {
      geom_lds[i] = p.scene.leaf_geom[i];
      index_lds[i] = p.scene.leaf_index[i];
    }
    bvh_src.nodes = nodes_lds;
    bvh_src.leaf_geom = geom_lds;
    bvh_src.leaf_index = index_lds;
    __syncthreads();
  }

  Path P[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    P[r].alive = false;
    P[r].has_item = false;
    P[r].sample = P[r].sample_end = 0;
    P[r].ox = P[r].oy = P[r].oz = 0.0f;
    P[r].dx = P[r].dy = P[r].dz = 0.0f;
    P[r].a = 1.0f;
    prep_segment<MFMA>(P[r]);
    P[r].closest = kTMax;
    P[r].best = kNone;
  }
  uint32_t grab_next = 0, grab_end = 0;
  bool exhausted = false;
  bool idle = false;   // no path of the wave was alive at the last look (wave-uniform)
  unsigned long long n_rays = 0, n_samples = 0;   // of the wave (wave-uniform)
  uint32_t bvh_interior = 0, bvh_tests = 0;   // diagnostic build: nodes opened, spheres tested
  PhaseClock clk = {};
#ifdef RT_DIAG_PHASES
  clk.t0 = __builtin_amdgcn_s_memtime();
  clk.t_start = clk.t0;
#endif
#ifdef RT_DIAG_DRAIN
  clk.real_start = __builtin_amdgcn_s_memrealtime();
#endif

  for (;;) {
    RT_MARK("refill");
    // ---- refill: next sample of the item, or the next item ------------------------------
    // Ready pixels are merged a few at a time (the wave looks every kMergeEvery steps), or at once
    // when the wave is about to reserve pixels and has too few free slots, or has no live path
    // left.  (A merge is seven round trips to the ring in a row: a wave that merged every pixel as
    // it became ready while its last paths ran paid that latency in most of those steps.)
    if (sliced) {
      const bool short_of_slots = grab_next == grab_end &&
                                  (uint32_t)__builtin_popcount(rs.free_mask) < p.grab_pixels;
      if (short_of_slots || idle || (clk.steps & (kMergeEvery - 1u)) == 0) {
        const uint32_t ready = ready_slots(st, p.slices, lane);
        if (ready != 0 && (short_of_slots || idle || (uint32_t)__builtin_popcount(ready) >= kMergeBatch)) {
#ifdef RT_DIAG_PHASES
          clk.merges++;
          clk.merged_pixels += (uint32_t)__builtin_popcount(ready);
#endif
          merge_ready<MFMA>(ready, p, st, rs, lane);
        }
      }
    }
    rs.starved = false;
    rs.waiting = false;
    if constexpr (!ACCEL && RT_LEND != 0) {
      if (exhausted && sliced) return_loans(P, reinterpret_cast<uint32_t*>(wl.rays), lane);
    }
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const bool need = !P[r].alive && P[r].sample >= P[r].sample_end;
      // (an item that waits for a borrowed sample, or is one, is not written: see lend_samples)
      const bool on_loan = !ACCEL && RT_LEND != 0 && sliced && (P[r].out_index >> 30) != 0;
      if (need && P[r].has_item && !on_loan) {
        write_item(P[r], p, st);
        P[r].has_item = false;
      }
      uint32_t slot;
      const uint32_t item = wave_fetch(need, grab_next, grab_end, exhausted, rs, slot, wg, p, lane);
      if (need && item != kNone) setup_item(P[r], item, slot, p, st);
      const bool begin = !P[r].alive && P[r].has_item && P[r].sample < P[r].sample_end;
      // (counted for the wave on the scalar unit: no register per lane, no vector instruction)
      n_samples += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(begin));
      if (begin) {
        start_sample<MFMA>(P[r], p);
        if (MFMA) mfma_guard(P[r], p.scene.mfma_scale);
      }
    }
    if constexpr (!ACCEL) {
      // Out of work: a lane whose first path slot is finished takes its second slot's path into
      // the first.  The wave's last paths then sit in slot 0 (unless a lane still has two), and
      // everything a step does per slot - a pass of the matrix scan, the shading - is done once.
      if (RT_LEND != 0 && exhausted && sliced) lend_samples(P, reinterpret_cast<uint32_t*>(wl.rays), lane);
      if (exhausted) {
        // (a path that lent or borrowed a sample is known to its partner by its slot and stays)
        const bool move = !P[0].alive && !P[0].has_item && P[1].alive &&
                          !(RT_LEND != 0 && sliced && (P[1].out_index >> 30) != 0);
        if (move) {
          P[0] = P[1];
          P[1].alive = false;
          P[1].has_item = false;
          P[1].sample = P[1].sample_end = 0;
        }
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int r = 0; r < RP; ++r) any_alive = any_alive || P[r].alive;
    // (a wave whose paths have all ended may still hold pixels open, all of them ready: it goes
    // round once more - an empty step - and merges them above, being short of slots or out of
    // work, or it would have found work; then it either finds work or ends)
    // A wave that found the workgroup's range empty while another wave was fetching the next
    // chunk goes round too, and asks again.
    const bool pending = (sliced && rs.free_mask != all_slots) || rs.waiting;
    idle = __builtin_amdgcn_ballot_w64(any_alive) == 0;
    const bool wave_alive = !idle || pending;
    if (MULTI) {
      if (!__syncthreads_or(wave_alive ? 1 : 0)) break;   // workgroup-uniform
    } else {
      if (!wave_alive) break;
    }

    RT_STAMP(clk.refill, clk.t0);
    RT_MARK("hit_setup");
    clk.steps++;
#ifdef RT_DIAG_PHASES
    if (sliced && rs.starved) {
      clk.starved_steps++;
#pragma unroll
      for (int r = 0; r < RP; ++r)
        clk.starved_idle += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(!P[r].alive));
    }
#endif

    // ---- World::hit --------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      P[r].closest = kTMax;
      P[r].best = kNone;
    }
    // Out of work and nearly drained: trace what is left spheres-across-lanes (tail_hit).
    // Zero or non-finite rays keep to the scan, whose replay has the literal fallback.
    bool tail = false, compact = false;
    [[maybe_unused]] bool few = false;
#ifdef RT_DIAG_DRAIN
    if (exhausted && clk.t_exhausted == 0) {
      clk.at_exhaustion[0] = clk.refill; clk.at_exhaustion[1] = clk.scan;
      clk.at_exhaustion[2] = clk.replay; clk.at_exhaustion[3] = clk.shade;
      clk.real_exhausted = __builtin_amdgcn_s_memrealtime();
    }
#endif
#ifdef RT_DIAG_PHASES
    if (exhausted && clk.t_exhausted == 0) clk.t_exhausted = __builtin_amdgcn_s_memtime();
#endif
    if (!ACCEL && exhausted) {
      uint32_t live = 0;
      bool odd = false;
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        live += __builtin_popcountll(__builtin_amdgcn_ballot_w64(P[r].alive));
        odd = odd || (P[r].alive && P[r].degenerate);
      }
      tail = live <= (MFMA ? kTailPathsMfma : kTailPaths) && __builtin_amdgcn_ballot_w64(odd) == 0;
      // (matrix filter, one chunk) few enough paths for one ray block: see mfma_prepare_compact
      compact = MFMA && !MULTI && !tail && live <= kCompactPaths && __builtin_amdgcn_ballot_w64(odd) == 0;
      few = live <= 16u;
    }
    if constexpr (ACCEL) {
      if (p.bvh_in_lds) bvh_hit<true>(P[0], p.scene, bvh_src, bvh_stack, bvh_interior, bvh_tests);
      else bvh_hit<false>(P[0], p.scene, bvh_src, bvh_stack, bvh_interior, bvh_tests);
      RT_STAMP(clk.scan, clk.t0);   // the traversal is this mode's "scan"
    } else if constexpr (MFMA) {
      const uint32_t n_in = n - p.scene.mfma_always;   // positions with rows; the rest are listed
      h8* stage = reinterpret_cast<h8*>(wl.rays);
      if constexpr (MULTI) {
        // several chunks: the columns of both halves are built once and kept across the chunks
        h8 cols[R][4];
        if (!tail) {
#pragma unroll
          for (int r = 0; r < R; ++r) mfma_prepare_half(P[r], p.scene.mfma_scale, stage, lane, cols[r]);
        }
        for (uint32_t base = 0, chunk = 0; base < n; base += kMfmaChunk, ++chunk) {
          const uint32_t cn = n - base < kMfmaChunk ? n - base : kMfmaChunk;
          __syncthreads();   // every wave is done with the previous chunk
          stage_chunk_mfma(p.scene, chunk, cn, mrows, mgeom, mindex);
          mc.count = cn;
          __syncthreads();
          const uint32_t n_rows = base >= n_in ? 0u : (n_in - base < cn ? n_in - base : cn);
          if (!tail) {
            mfma_list_always(P, mc, n_rows, cn, wl, lane, clk);
#pragma unroll
            for (int r = 0; r < R; ++r)
              if (mfma_half_scanned(P[r])) scan_pass_mfma(P, cols[r], (uint32_t)r, mc, n_rows, wl, lane, clk);
            RT_MARK("replay");
            replay_lists<true>(P, mc.geom, nullptr, 0, wl, lane, clk, mc.index, mc.count);
            RT_MARK("replay_end");
            RT_STAMP(clk.replay, clk.t0);
          }
        }
      } else if (compact) {
        mfma_test_always(P, mc, n_in, n);
#if RT_COMPACT_TIERS
        if (few) {   // one ray block
          h8 cols[1];
          mfma_prepare_compact<1>(P, p.scene.mfma_scale, stage, lane, cols);
          scan_pass_mfma<1, true, RT_DRAIN_BLOCKS>(P, cols, 0u, mc, n_in, wl, lane, clk);
        } else
#endif
        {
          h8 cols[kCompactBlocks];
          mfma_prepare_compact<kCompactBlocks>(P, p.scene.mfma_scale, stage, lane, cols);
          scan_pass_mfma<kCompactBlocks, true>(P, cols, 0u, mc, n_in, wl, lane, clk);
        }
        replay_lists<true, true, true>(P, mc.geom, nullptr, 0, wl, lane, clk, mc.index, mc.count);
        RT_STAMP(clk.replay, clk.t0);
      } else if (!tail) {
        // one chunk: a half's columns are built right before its pass, so that only four
        // operands (16 VGPRs) are alive at a time
        mfma_test_always(P, mc, n_in, n);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!mfma_half_scanned(P[r])) continue;
          h8 cols[4];
#ifdef RT_DIAG_PHASES
          const unsigned long long prep_t0 = __builtin_amdgcn_s_memtime();
#endif
          mfma_prepare_half(P[r], p.scene.mfma_scale, stage, lane, cols);
#ifdef RT_DIAG_PHASES
          clk.prep_cycles += __builtin_amdgcn_s_memtime() - prep_t0;
#endif
          scan_pass_mfma(P, cols, (uint32_t)r, mc, n_in, wl, lane, clk);
        }
        RT_MARK("replay");
        replay_lists<true>(P, mc.geom, nullptr, 0, wl, lane, clk, mc.index, mc.count);
        RT_MARK("replay_end");
        RT_STAMP(clk.replay, clk.t0);
      }
      if (tail) {
        // the tail step's filter is the vector form: its constants are formed here, on demand
        // (no tail step is taken while a zero, non-finite or out-of-range ray is alive)
#pragma unroll
        for (int r = 0; r < RP; ++r)
          if (P[r].alive) prep_segment<false>(P[r]);
        if (MULTI) tail_hit<false, false>(P, p.scene, sph, sr2, reinterpret_cast<float*>(wl.list), lane);
        else tail_hit<false, true, true>(P, p.scene, mgeom, mkhalf, reinterpret_cast<float*>(wl.list), lane, mindex);
#pragma unroll
        for (int r = 0; r < RP; ++r) prep_segment<true>(P[r]);
      } else {
        literal_world_hit(P, p.scene);
      }
    } else if constexpr (MULTI) {
      for (uint32_t base = 0; base < n; base += kChunkSpheres) {
        const uint32_t cn = n - base < kChunkSpheres ? n - base : kChunkSpheres;
        __syncthreads();   // every wave is done with the previous chunk
        stage_chunk(p.scene, base, cn, sph, sr2);
        __syncthreads();
        if (!tail) scan_chunk<STRICT>(P, sph, sr2, cn, base, wl, lane, clk);
      }
      if (tail) tail_hit<STRICT, false>(P, p.scene, sph, sr2, reinterpret_cast<float*>(wl.list), lane);
    } else {
      if (tail) tail_hit<STRICT, true>(P, p.scene, sph, sr2, reinterpret_cast<float*>(wl.list), lane);
      else scan_chunk<STRICT>(P, sph, sr2, n, 0, wl, lane, clk);
    }
#ifdef RT_DIAG_PHASES
    if (tail) clk.tail_steps++;
    else if (exhausted) clk.drain_scan_steps++;
    if (compact) clk.compact_steps++;
    if (tail) clk.tail_cycles += __builtin_amdgcn_s_memtime() - clk.t0;
#endif

    // ---- the rest of shoot_ray's body ----------------------------------------------------
    RT_MARK("shade");
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      n_rays += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(P[r].alive));
      if (P[r].alive) {
        const V3 o = mk(P[r].ox, P[r].oy, P[r].oz), d = mk(P[r].dx, P[r].dy, P[r].dz);
        bool ended;
        V3 radiance = mk(0.0f, 0.0f, 0.0f);
        const float inv_len = inv_length(P[r].a);
        if (P[r].best == kNone) {
          radiance = mul(mk(P[r].tr, P[r].tg, P[r].tb), sky(d, inv_len));
          ended = true;
        } else {
          const Bounce b = shade_hit(p.scene, o, d, inv_len, P[r].closest, P[r].best, P[r].rng);
          if (!b.scattered) {
            ended = true;
          } else {
            if (b.kind != RT_KIND_DIELECTRIC) {
              P[r].tr *= b.atten.x; P[r].tg *= b.atten.y; P[r].tb *= b.atten.z;
            }
            P[r].ox = b.point.x; P[r].oy = b.point.y; P[r].oz = b.point.z;
            P[r].dx = b.dir.x; P[r].dy = b.dir.y; P[r].dz = b.dir.z;
            P[r].a = dot(b.dir, b.dir);
            prep_segment<MFMA>(P[r]);
            if (MFMA) mfma_guard(P[r], p.scene.mfma_scale);
            P[r].depth--;
            ended = P[r].depth == 0;   // `if depth <= 0 { return black }`, src/main.rs:192-194
          }
        }
        if (ended) {
          P[r].ar += radiance.x; P[r].ag += radiance.y; P[r].ab += radiance.z;
          P[r].alive = false;
          P[r].sample++;
        }
      }
    }
    RT_STAMP(clk.shade, clk.t0);
    RT_MARK("shade_end");
  }
  RT_MARK("epilogue");

#ifdef RT_DIAG_PHASES
  [[maybe_unused]] const unsigned long long t_end_of_wave = __builtin_amdgcn_s_memtime();
  if (clk.t_exhausted == 0) clk.t_exhausted = t_end_of_wave;   // never out of work while it had paths
  if (lane == 0) {
    atomicAdd(&p.counters->phase[0], clk.refill);
    atomicAdd(&p.counters->phase[1], clk.scan);
    atomicAdd(&p.counters->phase[2], clk.replay);
    atomicAdd(&p.counters->phase[3], clk.shade);
    atomicAdd(&p.counters->phase[4], clk.steps);
    const unsigned long long t_end = t_end_of_wave;
    atomicAdd(&p.counters->phase[5], t_end - clk.t_exhausted);   // drain time, summed over waves
    atomicMax(&p.counters->phase[6], t_end - clk.t_exhausted);   // longest drain
    atomicMax(&p.counters->phase[7], t_end - clk.t_start);       // longest-lived wave
    atomicAdd(&p.counters->phase[8], clk.drain_scan_steps);      // drain steps that scanned
    atomicAdd(&p.counters->phase[9], clk.tail_steps);            // drain steps in tail mode
    atomicMax(&p.counters->phase[10], clk.drain_scan_steps);
    atomicMax(&p.counters->phase[11], clk.tail_steps);
#ifndef RT_DIAG_TRIGGERS
    atomicAdd(&p.counters->phase[12], clk.tail_cycles);          // cycles inside tail_hit
#endif
    atomicAdd(&p.counters->phase[13], 1ull);                     // waves
  }
#ifdef RT_DIAG_DRAIN
  // a build for reading where a wave's time goes after it ran out of work: wave-cycles per phase
  // from that moment on, summed over waves
  if (lane == 0 && clk.t_exhausted != t_end_of_wave) {
    atomicAdd(&p.counters->phase[16], clk.refill - clk.at_exhaustion[0]);
    atomicAdd(&p.counters->phase[17], clk.scan - clk.at_exhaustion[1]);
    atomicAdd(&p.counters->phase[18], clk.replay - clk.at_exhaustion[2]);
    atomicAdd(&p.counters->phase[19], clk.shade - clk.at_exhaustion[3]);
  }
  // and when the waves started, ran out of work and ended, by the constant 100 MHz clock all
  // compute units share (complements, where the earliest is wanted: the counters start at zero)
  if (clk.real_exhausted == 0) clk.real_exhausted = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    atomicMax(&p.counters->phase[20], ~clk.real_start);
    atomicMax(&p.counters->phase[21], ~clk.real_exhausted);
    atomicMax(&p.counters->phase[22], clk.real_exhausted);
    atomicMax(&p.counters->phase[23], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    atomicMax(&p.counters->phase[15], clk.drain_scan_steps + clk.tail_steps);   // most steps after the work ran out
    // and a record per wave at the start of its own ring (free by now: every open pixel is
    // merged), for rt_debug_wave_log: started, out of work, ended (100 MHz clock), cycles and
    // steps after the work ran out, and the phases' cycles in that time
    if (p.slices != 1) {
      unsigned long long* rec = reinterpret_cast<unsigned long long*>(p.ring + ring_offset(p));
      rec[0] = clk.real_start;
      rec[1] = clk.real_exhausted;
      rec[2] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
      rec[3] = t_end_of_wave - clk.t_exhausted;
      rec[4] = clk.drain_scan_steps;
      rec[5] = clk.tail_steps;
      rec[6] = clk.steps;
      rec[7] = clk.compact_steps;
      rec[8] = clk.refill - clk.at_exhaustion[0];
      rec[9] = clk.scan - clk.at_exhaustion[1];
      rec[10] = clk.replay - clk.at_exhaustion[2];
      rec[11] = clk.shade - clk.at_exhaustion[3];
      rec[12] = clk.tail_cycles;
    }
  }
#endif
#ifdef RT_DIAG_RING
  if (lane == 0) {
    atomicAdd(&p.counters->phase[16], clk.starved_steps);
    atomicAdd(&p.counters->phase[17], clk.starved_idle);
    atomicAdd(&p.counters->phase[18], clk.merges);
    atomicAdd(&p.counters->phase[19], clk.merged_pixels);
  }
#endif
#ifdef RT_DIAG_TRIGGERS
  // a build for reading the scan's branch rate: {scan blocks, blocks that branched to the
  // queueing code, (sphere, path) pairs queued there}, summed over waves
  const unsigned long long pairs = MFMA ? wave_sum((uint32_t)clk.trig_pairs) : clk.trig_pairs;
  if (lane == 0) {
    atomicAdd(&p.counters->phase[16], clk.blocks);
    atomicAdd(&p.counters->phase[17], clk.trig_blocks);
    atomicAdd(&p.counters->phase[18], pairs);
    atomicAdd(&p.counters->phase[19], clk.trig_cycles);
    atomicAdd(&p.counters->phase[12], clk.prep_cycles);   // (this build reports no tail cycles)
  }
#endif
  {
    // the hierarchy's counts, or (scan) queued candidates and 64 x the replay loop's trips
    const unsigned long long bi = ACCEL ? wave_sum(bvh_interior) : wave_sum((uint32_t)clk.cand);
    const unsigned long long bt = ACCEL ? wave_sum(bvh_tests) : clk.replay_iters * 64ull;
#ifndef RT_DIAG_DRAIN
    if (lane == 0) {
      atomicAdd(&p.counters->phase[14], bi);                     // hierarchy: interior nodes opened
      atomicAdd(&p.counters->phase[15], bt);                     // hierarchy: leaf spheres tested
    }
#else
    (void)bi; (void)bt;
#endif
  }
#endif
  if (lane == 0) {
    atomicAdd(&p.counters->rays, n_rays);
    atomicAdd(&p.counters->samples, n_samples);
  }
}

// A frame at max_depth 0: shoot_ray(ray, 0) is black (src/main.rs:192-194), so every pixel the
// call's tiles cover is (0, 0, 0) - resolved or not - and nothing is traced.
__global__ __launch_bounds__(256) void zero_kernel(const ZeroParams p) {
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < p.total_pixels; g += gridDim.x * 256u) {
    const TileDev t = find_tile(g, p.n_tiles, p.tile0, p.tiles);
    const uint32_t local = g - t.first_pixel;
    const uint32_t ty = fast_div(local, t.wdiv);
    const uint32_t tx = local - ty * t.w;
    float* dst = p.out + 3ull * (t.out_offset + (unsigned long long)ty * t.out_stride + tx);
    dst[0] = 0.0f; dst[1] = 0.0f; dst[2] = 0.0f;
  }
}

// compose_final_framebuffer's per-channel arithmetic (src/cpu/mod.rs:67-75) over a buffer of
// sums that several passes accumulated.
__global__ __launch_bounds__(256) void resolve_kernel(const ResolveParams p) {
  const uint64_t n = p.n_pixels * 3;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    p.out[i] = resolve_channel(p.sums[i], p.inv_spp);
}

// Per-ray test hook: one World::hit and one scatter decision per input ray, through the same
// scan / replay / shade code as the main kernel.  Path slot 0 of each lane carries the ray.
// bvh_hit strides its LDS stack by kThreadsAccel; this kernel runs it in kThreads-wide workgroups.
static_assert(kThreads == kThreadsAccel, "hook_kernel's traversal stack assumes equal workgroup sizes");
template <bool STRICT, bool MFMA = false>
__global__ __launch_bounds__(kThreads) void hook_kernel(const HookParams p) {
  extern __shared__ float4 lds[];
  float4* sph = lds;
  constexpr uint32_t cap = lds_capacity(kChunkSpheres);
  float* sr2 = reinterpret_cast<float*>(lds + cap);
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = threadIdx.x / kWave;
  uint4* mrows = reinterpret_cast<uint4*>(lds);
  float4* mgeom = lds + 4 * kMfmaRows;
  uint32_t* mindex = reinterpret_cast<uint32_t*>(mgeom + kMfmaChunk);
  // this wave's candidate lists and hit keys, behind the sphere records
  unsigned char* wave_lds = (MFMA ? reinterpret_cast<unsigned char*>(mindex + 2 * kMfmaChunk)
                                  : reinterpret_cast<unsigned char*>(sr2 + cap)) + wave * kWaveLdsBytes;
  WaveLists wl;
  wl.list = reinterpret_cast<uint32_t*>(wave_lds);
  wl.keys = reinterpret_cast<unsigned long long*>(wave_lds + (size_t)L * sizeof(uint32_t));
  wl.rays = reinterpret_cast<float4*>(wave_lds + (size_t)L * sizeof(uint32_t) +
                                      (size_t)kSlots * sizeof(unsigned long long));
  wl.counter = reinterpret_cast<uint32_t*>(wave_lds + kWaveListBytes);
  wl.count = 0;
  wl.counter[1] = 0u;
  wl.redo_from = kNone;
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  const bool live = i < p.n_rays;

  Path P[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    P[r].alive = false;
    P[r].ox = P[r].oy = P[r].oz = 0.0f;
    P[r].dx = P[r].dy = P[r].dz = 0.0f;
    P[r].a = 1.0f;
    prep_segment(P[r]);
    P[r].closest = kTMax;
    P[r].best = kNone;
  }
  if (live) {
    const rt_ray_in in = p.rays[i];
    P[0].ox = in.ox; P[0].oy = in.oy; P[0].oz = in.oz;
    P[0].dx = in.dx; P[0].dy = in.dy; P[0].dz = in.dz;
    P[0].a = dot(mk(in.dx, in.dy, in.dz), mk(in.dx, in.dy, in.dz));
    prep_segment(P[0]);
    if (MFMA) mfma_guard(P[0], p.scene.mfma_scale);
    P[0].alive = true;
  }
  const uint32_t n = p.scene.n_spheres;
  if constexpr (MFMA) {
    MfmaChunk mc;
    mc.rows = reinterpret_cast<const h8*>(mrows);
    mc.geom = mgeom;
    mc.index = mindex;
    mc.count = 0;
    h8 cols[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
      mfma_prepare_half(P[r], p.scene.mfma_scale, reinterpret_cast<h8*>(wl.rays), lane, cols[r]);
    const uint32_t n_in = n - p.scene.mfma_always;
    PhaseClock clk = {};
    for (uint32_t base = 0, chunk = 0; base < n; base += kMfmaChunk, ++chunk) {
      const uint32_t cn = n - base < kMfmaChunk ? n - base : kMfmaChunk;
      __syncthreads();
      stage_chunk_mfma(p.scene, chunk, cn, mrows, mgeom, mindex);
      mc.count = cn;
      __syncthreads();
      const uint32_t n_rows = base >= n_in ? 0u : (n_in - base < cn ? n_in - base : cn);
      mfma_list_always(P, mc, n_rows, cn, wl, lane, clk);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (mfma_half_scanned(P[r])) scan_pass_mfma(P, cols[r], (uint32_t)r, mc, n_rows, wl, lane, clk);
      replay_lists<true>(P, mc.geom, nullptr, 0, wl, lane, clk, mc.index, mc.count);
    }
    literal_world_hit(P, p.scene);
  } else if (p.accel) {
    BvhSource bvh_src;
    bvh_src.nodes = reinterpret_cast<const float4*>(p.scene.bvh);
    bvh_src.leaf_geom = p.scene.leaf_geom;
    bvh_src.leaf_index = p.scene.leaf_index;
    uint32_t unused_a = 0, unused_b = 0;
    bvh_hit<false>(P[0], p.scene, bvh_src, reinterpret_cast<uint32_t*>(lds) + threadIdx.x, unused_a,
                   unused_b);
  } else {
    for (uint32_t base = 0; base < n; base += kChunkSpheres) {
      const uint32_t cn = n - base < kChunkSpheres ? n - base : kChunkSpheres;
      __syncthreads();
      stage_chunk(p.scene, base, cn, sph, sr2);
      __syncthreads();
      PhaseClock clk = {};
      scan_chunk<STRICT>(P, sph, sr2, cn, base, wl, lane, clk);
    }
  }
  if (!live) return;

  rt_ray_out out;
  const V3 o = mk(P[0].ox, P[0].oy, P[0].oz), d = mk(P[0].dx, P[0].dy, P[0].dz);
  PathRng rng;
  rng.g = rt_rng_init(p.seed, i, 0);
  rng.draws = 0;
  out.sphere = 0; out.material = 0; out.front_face = 0; out.scattered = 0; out.draws = 0;
  out.t = 0.0f;
  out.px = out.py = out.pz = 0.0f;
  out.nx = out.ny = out.nz = 0.0f;
  out.sx = out.sy = out.sz = 0.0f;
  const float inv_len = inv_length(P[0].a);
  if (P[0].best == kNone) {
    const V3 s = sky(d, inv_len);
    out.hit = 0;
    out.kind = RT_KIND_MISS;
    out.ar = s.x; out.ag = s.y; out.ab = s.z;
  } else {
    const Bounce b = shade_hit(p.scene, o, d, inv_len, P[0].closest, P[0].best, rng);
    out.hit = 1;
    out.sphere = P[0].best;
    out.material = b.material;
    out.kind = b.kind;
    out.front_face = b.front ? 1u : 0u;
    out.scattered = b.scattered ? 1u : 0u;
    out.draws = rng.draws;
    out.t = P[0].closest;
    out.px = b.point.x; out.py = b.point.y; out.pz = b.point.z;
    out.nx = b.normal.x; out.ny = b.normal.y; out.nz = b.normal.z;
    out.sx = b.dir.x; out.sy = b.dir.y; out.sz = b.dir.z;
    out.ar = b.atten.x; out.ag = b.atten.y; out.ab = b.atten.z;
  }
  p.out[i] = out;
}

constexpr size_t queue_bytes(int threads) {
  return (size_t)(threads / kWave) * kWaveLdsBytes;
}
static_assert(kThreadsSingle >= 2 * kWave && kThreadsAccel >= 2 * kWave,
              "the workgroup's cursor sits in the headers of its first two waves");
// the chunked kernel's workgroup: records of one chunk, the waves' lists, 256 bytes of statics
static_assert((size_t)lds_capacity(kChunkSpheres + 1) * (sizeof(float4) + sizeof(float)) +
                      queue_bytes(kThreads) + 256 <= kLdsPerWorkgroup,
              "two workgroups of the chunked kernel must fit into a CU's LDS");

// Self-test of the short square roots: every one of the 2^32 floats through sqrt_rn and
// sqrt_1my against the compiler's sqrtf.  out[0] = inputs where sqrt_rn gives other bits,
// out[1] = inputs outside 0 < |x| < 2^-96 where sqrt_1my does, out[2] = inputs inside that gap
// where sqrt_1my does (what sqrt_rn's branch is there for; not zero), out[3] = inputs where
// inv_sqrt_rn gives other bits than 1.0f / sqrtf.
__global__ __launch_bounds__(256) void sqrt_check_kernel(unsigned long long* out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t bad_rn = 0, bad_1my = 0, gap = 0, bad_inv = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += stride) {
    float x = __uint_as_float((uint32_t)v);
    asm volatile("" : "+v"(x));
    const uint32_t want = __float_as_uint(__builtin_sqrtf(x));
    const uint32_t mag = (uint32_t)v & 0x7FFFFFFFu;
    const bool in_gap = mag != 0u && mag < 0x0F800000u;
    if (__float_as_uint(sqrt_rn(x)) != want) ++bad_rn;
    if (__float_as_uint(sqrt_1my(x)) != want) {
      if (in_gap) ++gap; else ++bad_1my;
    }
    const uint32_t want_inv = __float_as_uint(1.0f / __builtin_sqrtf(x));
    if (__float_as_uint(inv_sqrt_rn(x)) != want_inv) ++bad_inv;
  }
  if (bad_inv) atomicAdd(&out[3], (unsigned long long)bad_inv);
  if (bad_rn) atomicAdd(&out[0], (unsigned long long)bad_rn);
  if (bad_1my) atomicAdd(&out[1], (unsigned long long)bad_1my);
  if (gap) atomicAdd(&out[2], (unsigned long long)gap);
}

}  // namespace

size_t trace_lds_bytes(uint32_t n_spheres, bool accel, uint32_t bvh_nodes, uint32_t bvh_depth,
                       bool* bvh_in_lds, uint32_t* slot_table_offset) {
  if (accel) {
    // one stack row per level of the tree (at least one); the copy of the hierarchy goes in
    // when the workgroups that fill a CU's wave slots still all fit into its LDS
    const size_t stacks = (size_t)(bvh_depth ? bvh_depth : 1) * kThreadsAccel * sizeof(uint32_t);
    size_t copy = (size_t)bvh_nodes * sizeof(BvhNode) +
                  (size_t)n_spheres * (sizeof(float4) + sizeof(uint32_t));
    copy = (copy + 15) & ~(size_t)15;
    // the waves' tables of open pixels come last
    const size_t tables = (size_t)(kThreadsAccel / kWave) * (kSlotTableBytes + 16);
    constexpr size_t wgs_per_cu = (size_t)RT_ACCEL_WAVES * 4 * kWave / kThreadsAccel;
    const bool fits = stacks + copy + tables <= 160 * 1024 / wgs_per_cu;
    if (bvh_in_lds) *bvh_in_lds = fits;
    if (slot_table_offset) *slot_table_offset = (uint32_t)(fits ? stacks + copy : stacks);
    return (fits ? stacks + copy : stacks) + tables;
  }
  return (size_t)lds_capacity(n_spheres) * (sizeof(float4) + sizeof(float)) +
         queue_bytes(trace_threads(n_spheres > kChunkSpheres, false));
}

size_t trace_lds_bytes_mfma(int threads) {
  return (size_t)4 * kMfmaRows * sizeof(uint4) +
         (size_t)kMfmaChunk * (sizeof(float4) + sizeof(uint32_t) + sizeof(float)) + queue_bytes(threads);
}
static_assert((size_t)4 * kMfmaRows * sizeof(uint4) +
                      (size_t)kMfmaChunk * (sizeof(float4) + sizeof(uint32_t) + sizeof(float)) +
                      queue_bytes(kThreadsMfma) + 256 <= 160 * 1024,
              "the matrix scan's workgroup must fit into a CU's LDS");
static_assert((size_t)kSlots * 2 * sizeof(float4) >= 64 * 4 * sizeof(uint4), "mfma_prepare_half stages 64 columns in the wave's published-ray memory");

uint32_t bvh_stack_depth() { return kBvhStack; }

template <bool STRICT, bool MULTI>
static hipError_t occupancy(size_t lds, int* out) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, trace_kernel<STRICT, MULTI>,
                                                      trace_threads(MULTI, false), lds);
}

hipError_t trace_max_blocks_per_cu(bool strict, bool multi, bool accel, size_t lds, int* out) {
  if (accel)
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, trace_kernel<false, false, true>,
                                                        kThreadsAccel, lds);
  if (strict) return multi ? occupancy<true, true>(lds, out) : occupancy<true, false>(lds, out);
  return multi ? occupancy<false, true>(lds, out) : occupancy<false, false>(lds, out);
}

hipError_t trace_max_blocks_per_cu_mfma(bool multi, size_t lds, int* out) {
  // 155 KB of dynamic LDS: above the default limit of a kernel
  hipError_t e;
  if (multi) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_kernel<false, true, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, trace_kernel<false, true, false, true>,
                                                        kThreadsMfma, lds);
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_kernel<false, false, false, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, trace_kernel<false, false, false, true>,
                                                      kThreadsMfma, lds);
}

hipError_t launch_trace(const TraceParams& p, bool strict, bool accel, bool mfma, int blocks,
                        hipStream_t stream) {
  if (mfma) {
    const bool many = p.scene.n_spheres > kMfmaChunk;
    const size_t lds = trace_lds_bytes_mfma(kThreadsMfma);
    const dim3 grid(blocks), block(kThreadsMfma);
    if (many) hipLaunchKernelGGL((trace_kernel<false, true, false, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((trace_kernel<false, false, false, true>), grid, block, lds, stream, p);
    return hipGetLastError();
  }
  const bool multi = p.scene.n_spheres > kChunkSpheres;
  const size_t lds = trace_lds_bytes(p.scene.n_spheres, accel, p.scene.bvh_info.n_nodes,
                                     p.bvh_stack, nullptr);
  const dim3 grid(blocks), block(trace_threads(multi, accel));
  if (accel) {
    hipLaunchKernelGGL((trace_kernel<false, false, true>), grid, block, lds, stream, p);
  } else if (strict) {
    if (multi) hipLaunchKernelGGL((trace_kernel<true, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((trace_kernel<true, false>), grid, block, lds, stream, p);
  } else {
    if (multi) hipLaunchKernelGGL((trace_kernel<false, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((trace_kernel<false, false>), grid, block, lds, stream, p);
  }
  return hipGetLastError();
}

// The work cursor and the ray / sample counts of a launch, cleared by a kernel of its own (a
// node like any other when the stream is being captured into a graph).
__global__ __launch_bounds__(64) void clear_counters_kernel(Counters* c) {
  uint32_t* words = reinterpret_cast<uint32_t*>(c);
  for (uint32_t i = threadIdx.x; i < sizeof(Counters) / sizeof(uint32_t); i += 64) words[i] = 0u;
}
hipError_t launch_clear_counters(Counters* c, hipStream_t stream) {
  hipLaunchKernelGGL(clear_counters_kernel, dim3(1), dim3(64), 0, stream, c);
  return hipGetLastError();
}

hipError_t launch_zero(const ZeroParams& p, hipStream_t stream) {
  uint64_t blocks = ((uint64_t)p.total_pixels + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_resolve(const ResolveParams& p, hipStream_t stream) {
  uint64_t blocks = (p.n_pixels * 3 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_sqrt_check(unsigned long long* out4, hipStream_t stream) {
  hipLaunchKernelGGL(sqrt_check_kernel, dim3(4096), dim3(256), 0, stream, out4);
  return hipGetLastError();
}

hipError_t launch_hook(const HookParams& p, bool strict, hipStream_t stream) {
  if (p.n_rays == 0) return hipSuccess;
  const size_t lds = (size_t)lds_capacity(kChunkSpheres) * (sizeof(float4) + sizeof(float)) +
                     queue_bytes(kThreads);
  const dim3 grid((p.n_rays + kThreads - 1) / kThreads), block(kThreads);
  if (p.mfma) {
    const size_t mlds = trace_lds_bytes_mfma(kThreads);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hook_kernel<false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((hook_kernel<false, true>), grid, block, mlds, stream, p);
  } else if (strict) hipLaunchKernelGGL((hook_kernel<true>), grid, block, lds, stream, p);
  else hipLaunchKernelGGL((hook_kernel<false>), grid, block, lds, stream, p);
  return hipGetLastError();
}

}  // namespace rt
