/* rt_abi.h - C ABI of the MI355X (gfx950) path-tracing library, librt_hip.so.
 *
 * This is the drop-in boundary for the reference's CPU tile renderer (Nanoteck137/ray_tracing,
 * src/cpu).  A host that owns the reference's scene / camera / tile types fills the POD mirrors
 * below and calls rt_render() where it used to call dispatch_work_cpu_multithreaded(), or
 * rt_render_tiles() where it used to loop execute_job() over its tile queue.  No type from any
 * framework appears in a signature: plain pointers, sizes and integers only.
 *
 * Conventions
 *  - every function returns 0 on success and a negative rt_status on failure; the message is
 *    available from rt_last_error() (thread-local).  Nothing aborts and nothing throws.
 *  - the caller owns every buffer it passes; the library keeps no pointer after a call returns
 *    (rt_scene objects own their own device copies).
 *  - calls are blocking unless the name ends in _async or _noalloc.  The library keeps one set
 *    of contexts per process behind one lock: calls from several host threads are serialised,
 *    never concurrent.  Every call leaves the calling thread's current HIP device as it found it.
 *  - results are a pure function of (seed, scene, camera, W, H, spp, max_depth, flags): they do
 *    not depend on tiling, on launch geometry or on how many GPUs share the frame.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rt_render renders exactly the listed tiles (a partial list used to be ignored and the whole
 *    frame rendered); rt_render_multi, rt_accumulate_*, rt_resolve_device_async,
 *    rt_render_rows_device_noalloc, rt_scene_scan_info and RT_FLAG_SCAN_VALU were added; calls
 *    from several host threads are serialised behind one lock.  Nothing was removed.
 * 3: no entry point changed, but a seed gives other frames than before: the draws inside a sample
 *    come from another recurrence (include/rt_rng.h); the distribution and the order of draws are
 *    the same.
 * 4: the same again (a linear congruential step inside a sample). */
#define RT_ABI_VERSION 4

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARGUMENT = -1,
  RT_ERR_NOT_INITIALIZED = -2,
  RT_ERR_DEVICE = -3,        /* a HIP call failed; rt_last_error() names it */
  RT_ERR_OUT_OF_MEMORY = -4,
  RT_ERR_LIMIT = -5          /* a documented size limit was exceeded */
} rt_status;

/* Mirror of `struct Sphere { position: Vec3, radius: f32, material_id: usize }`
 * (reference src/main.rs:119-125).  20 bytes, no padding. */
typedef struct rt_sphere {
  float x, y, z;
  float radius;
  uint32_t material_id;
} rt_sphere;

/* Mirror of `struct Material { color, metallic, metallic_strength, dielectric, ir }`
 * (reference src/main.rs:108-117).  The two bools travel as bits of `flags`.
 * Shading precedence is dielectric > metallic > diffuse (src/main.rs:201,228,243). */
#define RT_MAT_METALLIC 1u
#define RT_MAT_DIELECTRIC 2u
typedef struct rt_material {
  float r, g, b;
  uint32_t flags;
  float metallic_strength;
  float ir;
} rt_material;

/* Mirror of `struct Camera { view_matrix_inv: Mat4, projection_matrix_inv: Mat4 }`
 * (reference src/main.rs:51-55): two column-major 4x4 f32, the order of glam's to_cols_array(). */
typedef struct rt_camera {
  float view_inv[16];
  float proj_inv[16];
} rt_camera;

/* Mirror of `struct TileJob { x, y, width, height }` (reference src/cpu/job.rs:6-13). */
typedef struct rt_tile {
  uint32_t x, y, width, height;
} rt_tile;

/* Request flags */
#define RT_FLAG_STRICT_FP 1u      /* evaluate the ray/sphere discriminant in the reference's exact
                                     operation order with no fused multiply-add (slower; every
                                     hit decision is then bit-identical to the CPU oracle) */
#define RT_FLAG_SEQUENTIAL_SUM 2u /* never split one pixel's samples over several lanes: the
                                     per-pixel sum is then formed in sample order, exactly like
                                     `pixel_color += color` in src/cpu/job.rs:58 */

#define RT_FLAG_ACCEL 4u          /* World::hit (src/main.rs:175-189) through a bounding-volume
                                     hierarchy instead of the linear scan over every sphere.
                                     Opt-in: the work no longer grows with the sphere count, so
                                     a roofline counted in ray/sphere tests does not apply to it.
                                     Every sphere the traversal reaches goes through the same
                                     exact arithmetic and tie rule, and its boxes are widened by
                                     a bound on the reference's own rounding, so the result is
                                     bit-identical to the scan's */

#define RT_FLAG_SCAN_VALU 8u      /* keep the linear scan's sphere filter on the vector ALU (seven
                                     fused f32 instructions and a compare per ray/sphere pair).
                                     Without this flag the filter runs on the matrix pipe where
                                     the scene allows it: the sign of the discriminant is a
                                     bilinear form of ten numbers of the sphere and ten of the
                                     ray, evaluated in split f16 by v_mfma_f32_16x16x32_f16 with a
                                     slack that covers its rounding.  Either filter only chooses
                                     which spheres get the exact arithmetic, and neither drops a
                                     sphere that arithmetic would accept: same bits both ways */

/* Mirror of `struct Dispatch` (reference src/cpu/mod.rs:19-30) plus the render seed.
 * rt_render() renders exactly the listed tiles and leaves every other pixel (0,0,0), like the
 * reference's compose_final_framebuffer (src/cpu/mod.rs:57-80); `tiles` == NULL (or a list that
 * covers the image) means the whole frame.  rt_render_tiles() requires the list. */
typedef struct rt_request {
  uint32_t image_width;
  uint32_t image_height;
  uint32_t max_depth;
  uint32_t samples_per_pixel;
  uint32_t seed;
  uint32_t flags;
  rt_camera camera;
  const rt_sphere* spheres;
  uint32_t n_spheres;
  const rt_material* materials;
  uint32_t n_materials;
  const rt_tile* tiles;
  uint32_t n_tiles;
} rt_request;

/* Counters of the most recent render on this process's device. */
typedef struct rt_stats {
  uint64_t rays;        /* World::hit calls (one per path segment), src/main.rs:196 */
  uint64_t samples;     /* shoot_ray calls from the pixel loop, src/cpu/job.rs:57 */
  uint64_t sphere_tests; /* rays * n_spheres */
  float kernel_ms;      /* device time of the path-tracing kernel (HIP events on its stream) */
  uint32_t rays_per_lane;
  uint32_t slices;      /* lanes that shared one pixel's samples (1 = sequential sum) */
  uint32_t workgroups;
  uint32_t threads_per_workgroup;
} rt_stats;

/* Opaque device-resident scene (sphere geometry, per-sphere material ids, materials). */
typedef struct rt_scene rt_scene;

/* Parameters of a device-level render: everything in rt_request that is not a pointer. */
typedef struct rt_frame {
  uint32_t image_width;
  uint32_t image_height;
  uint32_t max_depth;
  uint32_t samples_per_pixel;
  uint32_t seed;
  uint32_t flags;
  rt_camera camera;
} rt_frame;

/* One record of the per-ray test hook. */
typedef struct rt_ray_in {
  float ox, oy, oz;
  float dx, dy, dz;
} rt_ray_in;

#define RT_KIND_MISS 0u
#define RT_KIND_DIFFUSE 1u
#define RT_KIND_METAL 2u
#define RT_KIND_DIELECTRIC 3u
typedef struct rt_ray_out {
  uint32_t hit;          /* 0 = miss (then ar,ag,ab hold the background colour) */
  uint32_t sphere;       /* index of the closest sphere */
  uint32_t material;
  uint32_t kind;         /* RT_KIND_* */
  uint32_t front_face;
  uint32_t scattered;    /* 0 when the path ends here (miss, or absorbed metal reflection) */
  uint32_t draws;        /* random numbers consumed by this bounce */
  float t;
  float px, py, pz;      /* hit point */
  float nx, ny, nz;      /* face normal */
  float sx, sy, sz;      /* scattered direction (not normalised, as in the reference) */
  float ar, ag, ab;      /* factor applied to the radiance returned by the next segment */
} rt_ray_out;

/* ---- lifecycle --------------------------------------------------------------------------- */

/* Select `device` (HIP ordinal) for this process, create the stream and the work counters.
 * Replaces nothing in the reference (its renderer needs no set-up); it is the FFI's one-time
 * initialisation.  Calling it again with the same device is a no-op. */
int rt_init(int device);
int rt_shutdown(void);
const char* rt_last_error(void);
int rt_abi_version(void);

/* ---- drop-in entry points (host pointers in, host pointers out) -------------------------- */

/* Replaces cpu::dispatch_work_cpu_multithreaded(dispatch, n) and
 * cpu::dispatch_work_cpu_singlethreaded(dispatch) (reference src/cpu/multi.rs:131-172,
 * src/cpu/single.rs:10-50) including their call of compose_final_framebuffer
 * (src/cpu/mod.rs:51-83).  out_rgb receives image_width*image_height RGB triples, row-major,
 * fb[x + y*W], y = 0 at the bottom, each channel clamp(sqrt(sum/spp), 0, 0.999). */
int rt_render(const rt_request* req, float* out_rgb);

/* rt_render over several GPUs driven by ONE process, which is what the reference's host is.
 * Rows are dealt to devices[0..n_devices) in blocks of 8, round-robin (the reference deals
 * tiles to `num_threads` workers, src/cpu/multi.rs:30-62, 140-141); every device renders its
 * blocks at the same time and the blocks are copied straight into out_rgb, so there is no
 * device-to-device exchange.  Pixels are keyed by their position in the whole frame: the result
 * is bit-identical to rt_render's for any device list (a device may be listed more than once).
 * rt_init() must have been called; the listed devices get contexts of their own, kept until
 * rt_shutdown(). */
int rt_render_multi(const rt_request* req, const int* devices, int n_devices, float* out_rgb);

/* Replaces the loop `for job in queue { execute_job(&job, &data) }` (reference
 * src/cpu/job.rs:36-74 as driven by src/cpu/single.rs:17-39 / src/cpu/multi.rs:30-62).
 * out_sums receives, tile after tile in the order of req->tiles, width*height RGB triples per
 * tile, row-major inside the tile: the UN-NORMALISED sums of JobResult.pixel_colors. */
int rt_render_tiles(const rt_request* req, float* out_sums);

/* The same loop cut into passes: adds samples [sample_begin, sample_begin +
 * req->samples_per_pixel) of every pixel to the sums inout_sums already holds (layout of
 * rt_render_tiles).  `pixel_color += color` (src/cpu/job.rs:58) is additive and the division by
 * the sample count happens later (src/cpu/mod.rs:67), so a host can render progressively, stop
 * on a time budget, or checkpoint the sums; the reference itself renders in one pass.  See
 * rt_accumulate_device_async for what the passes add up to. */
int rt_accumulate_tiles(const rt_request* req, uint32_t sample_begin, float* inout_sums);

/* Replaces compose_final_framebuffer (reference src/cpu/mod.rs:51-83) for callers that kept the
 * per-tile sums: scatters the tiles into out_rgb (W*H*3) applying x*(1/spp), sqrt, clamp. */
int rt_compose(const rt_tile* tiles, uint32_t n_tiles, const float* tile_sums,
               uint32_t image_width, uint32_t image_height, uint32_t samples_per_pixel,
               float* out_rgb);

/* ---- device-resident entry points (for callers that keep the frame in HBM) --------------- */

/* Upload a World (reference src/main.rs:168-172).  The reference keeps its world in host
 * vectors shared by the workers (src/cpu/multi.rs:143); this is the device copy. */
int rt_scene_create(const rt_sphere* spheres, uint32_t n_spheres,
                    const rt_material* materials, uint32_t n_materials, rt_scene** out_scene);
int rt_scene_destroy(rt_scene* scene);

/* execute_job over a set of tiles with the output left in device memory.
 *   d_out      device pointer; pixel (px,py) of tile i goes to
 *              d_out[3 * (out_offset[i] + (py - tile.y) * out_stride[i] + (px - tile.x))]
 *   resolve    0: un-normalised sums (JobResult contract); 1: compose_final_framebuffer's
 *              x*(1/spp), sqrt, clamp applied on the way out
 *   stream     a hipStream_t passed as void*, or NULL for the library's own stream
 * Enqueues the work and returns; rt_synchronize() or the caller's own stream sync completes it.
 * Same reference counterpart as rt_render_tiles. */
int rt_render_device_async(const rt_scene* scene, const rt_frame* frame,
                           const rt_tile* tiles, const uint64_t* out_offset,
                           const uint32_t* out_stride, uint32_t n_tiles,
                           int resolve, float* d_out, void* stream);

/* Convenience form of the above for one horizontal band of full-width rows
 * [row_begin, row_end): d_out holds (row_end-row_begin)*W RGB triples, row-major.  This is the
 * unit a multi-GPU host gives each rank (rows shard like the reference's tiles do over
 * threads, src/cpu/multi.rs:30-62). */
int rt_render_rows_device_async(const rt_scene* scene, const rt_frame* frame,
                                uint32_t row_begin, uint32_t row_end,
                                int resolve, float* d_out, void* stream);

/* Progressive rendering: adds samples [sample_begin, sample_begin + frame->samples_per_pixel)
 * of every covered pixel to the UN-NORMALISED sums d_sums already holds (same addressing as
 * rt_render_device_async with resolve = 0).  A sample's random stream is keyed by its absolute
 * index, so passes that cover [0, S) between them draw exactly the paths of one S-sample render.
 * With RT_FLAG_SEQUENTIAL_SUM every pass carries the per-pixel running sum on in sample order
 * and the result is bit-identical to that one render; otherwise a pass adds its own sum
 * (sliced like a render of its length) to the buffer in one addition.  The contract it builds on
 * is the additive JobResult.pixel_colors of src/cpu/job.rs:48-61 with the division by the
 * sample count left to compose_final_framebuffer (src/cpu/mod.rs:67); the reference itself
 * renders in one pass. */
int rt_accumulate_device_async(const rt_scene* scene, const rt_frame* frame, uint32_t sample_begin,
                               const rt_tile* tiles, const uint64_t* out_offset,
                               const uint32_t* out_stride, uint32_t n_tiles, float* d_sums,
                               void* stream);

/* compose_final_framebuffer's arithmetic (src/cpu/mod.rs:67-75), x*(1/spp), sqrt, clamp, over
 * n_pixels RGB sums in device memory; d_rgb may be d_sums. */
int rt_resolve_device_async(const float* d_sums, uint64_t n_pixels, uint32_t samples_per_pixel,
                            float* d_rgb, void* stream);

/* rt_render_rows_device_async for callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference.
 * Same arguments, same pixels (bit for bit), but the call never allocates, frees, waits or
 * records an event: everything it puts on the stream is two kernel launches (one that clears its
 * work counters, one that renders), so a call made while the stream is being captured becomes
 * two kernel nodes of the graph.  Its work counters and the rings in which the waves park the
 * sums of sliced samples live in one fixed device buffer that the library holds on every GPU it
 * runs on from load to exit (RT_NOALLOC_PLANE_BYTES of rings).  Frames of any size are accepted:
 * a wave's ring holds as many open pixels as fit (27 at 50 slices a pixel, against 32 when the
 * memory is the library's to size), which does not change a pixel.
 * Limits and conditions:
 *  - the launch geometry of a kernel variant is queried the first time the variant is used on a
 *    device (at rt_init for the default, matrix-filter kernels; at the first frame for
 *    RT_FLAG_SCAN_VALU / RT_FLAG_STRICT_FP / RT_FLAG_ACCEL and each new scene size): a call that
 *    would need the query while its stream is capturing fails with RT_ERR_INVALID_ARGUMENT;
 *    render once outside capture first;
 *  - RT_FLAG_ACCEL is accepted only for a scene whose hierarchy an allocating call has built;
 *  - rt_get_stats after such a call reports rays and samples, and kernel_ms as NaN;
 *  - every call on a GPU shares that GPU's buffer: two calls that overlap on one GPU (different
 *    streams, or different host threads' streams) RACE; calls on one stream, or otherwise
 *    ordered - the nodes of one captured stream are - are fine.  The counters are cleared by
 *    each call. */
#define RT_NOALLOC_PLANE_BYTES (64u << 20)
int rt_render_rows_device_noalloc(const rt_scene* scene, const rt_frame* frame,
                                  uint32_t row_begin, uint32_t row_end,
                                  int resolve, float* d_out, void* stream);

int rt_synchronize(void* stream);
int rt_get_stats(rt_stats* out);

/* Device time, in milliseconds, of each of the last `n_last` path-tracing kernel launches
 * (oldest first), measured with HIP events recorded on the stream each launch ran on.  At most
 * the 64 most recent launches are kept.  Waits for those launches to finish.  Returns the
 * number of values written, or a negative rt_status.  The counterpart in the reference is the
 * Instant::now()/elapsed() pair around the tile loop (src/cpu/multi.rs:146,163). */
int rt_get_launch_ms(uint32_t n_last, float* out_ms);

/* Diagnostic: wave-cycles the last launch spent per phase of its main loop, as
 * out16[0..7] = {refill, scan, replay, shade, steps, drain cycles summed over waves, longest
 * drain, longest-lived wave} and out16[8..13] = {drain steps that scanned (sum over waves),
 * drain steps in tail mode (sum), most scanning drain steps of one wave, most tail-mode steps of
 * one wave, cycles inside tail mode (sum), waves} and out16[14..15] = {interior nodes opened,
 * leaf spheres tested} by RT_FLAG_ACCEL traversals, or without that flag {candidates the scan
 * queued, 64 x trips of the waves' replay loops}; "drain" is the time after the work queue ran
 * dry.  All zero unless the library was built with -DRT_DIAG_PHASES (a build for reading
 * shares; its stamps slow the kernel down).  No counterpart in the reference. */
int rt_get_phase_cycles(uint64_t* out16);

/* How World::hit's scan will filter this scene's spheres.  No counterpart in the reference
 * (its World::hit, src/main.rs:175-189, tests every sphere); exists so that a host, a test or a
 * benchmark can tell which of the two filters its frames run on. */
typedef struct rt_scan_info {
  uint32_t matrix_filter;      /* 1: the filter runs on the matrix pipe unless a request sets
                                  RT_FLAG_SCAN_VALU, RT_FLAG_STRICT_FP or RT_FLAG_ACCEL (or its camera
                                  sits beyond the filter's range); 0: this scene keeps to the
                                  vector ALU (no sphere, or non-finite sizes) */
  uint32_t listed_spheres;     /* spheres outside the filter's f16 range (a factor 4 above the rest
                                  in max(|centre|, radius), at most 8): listed for every ray */
  float scale;                 /* power of two applied to centres, radii and ray origins */
  float origin_limit;          /* rays whose origin is farther than this from the World's origin
                                  take the literal loop over all spheres */
} rt_scan_info;
int rt_scene_scan_info(const rt_scene* scene, rt_scan_info* out);

/* ---- test hook --------------------------------------------------------------------------- */

/* One World::hit + one scatter decision per input ray (reference src/main.rs:175-189 and the
 * body of shoot_ray, src/main.rs:196-251), with ray i drawing its random numbers from the
 * stream rt_rng_init(seed, i, 0) of rt_rng.h.  Exists so that hit distance, normal and
 * scatter direction can be compared ray by ray with the CPU oracle. */
int rt_trace_rays(const rt_scene* scene, const rt_ray_in* rays, uint32_t n_rays,
                  uint32_t seed, uint32_t flags, rt_ray_out* out);

/* Self-test of the kernels' square roots.  Where the reference takes a square root (`.sqrt()`,
 * src/main.rs:141, 211, 320, and inside `.normalize()`, src/main.rs:80, 209, 256, 285; correctly
 * rounded, and in normalize followed by the correctly rounded `1.0 / length`), the kernels' hot
 * places use shorter instruction sequences than the compiler's sqrtf and division; this runs
 * every one of the 2^32 floats through both on the GPU and counts the inputs whose results differ
 * in any bit: out4[0] for the guarded square root used on arbitrary inputs (must be 0), out4[1]
 * for the unguarded form on the inputs it is used for (everything but 0 < |x| < 2^-96; must be
 * 0), out4[2] for the unguarded form inside that gap (what the guard is for; not 0), out4[3] for
 * 1 / sqrt(x) as normalize takes it (must be 0).  Takes about a second.  No counterpart in the
 * reference. */
int rt_selftest_sqrt(uint64_t* out4);

#ifdef __cplusplus
}
#endif

#endif /* RT_ABI_H */
