/* rt_oracle.h - CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker or as the timed CPU baseline.  Nothing under ray_tracing_amd/ links,
 * imports or calls it.
 *
 * PARITY UNPINNED: the reference (Nanoteck137/ray_tracing, Rust) has no tests, no golden
 * vectors and no recorded output, and neither rustc nor its crates (glam 0.21.2, rand 0.8.5,
 * bmp 0.5.0) exist in the build environment, so this restatement could not be checked against
 * a run of the reference.  It is pinned by the reference's source text, statement by statement
 * (each function cites the lines it follows), and by known-answer vectors derived by hand from
 * that text (tests/test_oracle_known_answers.py, tests/golden/).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/rt_abi.h" /* POD scene/camera/tile types only */
#include "../include/rt_rng.h" /* the counter-based stream shared with the GPU */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_vec3 {
  float x, y, z;
} ro_vec3;

typedef struct ro_ray {
  ro_vec3 origin, dir;
} ro_ray;

/* struct HitRecord, src/main.rs:86-95 */
typedef struct ro_hit {
  ro_vec3 point;
  ro_vec3 normal;
  float t;
  int front_face;
  uint32_t material_id;
  uint32_t sphere; /* index of the sphere that produced the record (not in the reference) */
} ro_hit;

/* struct World, src/main.rs:168-172 */
typedef struct ro_world {
  const rt_material* materials;
  uint32_t n_materials;
  const rt_sphere* spheres;
  uint32_t n_spheres;
} ro_world;

/* ChaCha12 as rand 0.8.5's thread_rng runs it (ro_chacha.c): an independent stream. */
typedef struct ro_chacha {
  uint32_t key[8];
  uint64_t counter; /* next block */
  uint64_t stream;
  uint32_t buffer[16];
  uint32_t index;   /* next unread word of buffer; 16 = empty */
} ro_chacha;
void ro_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                     uint32_t out[16]);
void ro_chacha12_seed(ro_chacha* c, const uint32_t key[8], uint64_t stream);
void ro_chacha12_seed_from(ro_chacha* c, uint32_t seed, uint32_t a, uint32_t b);
uint32_t ro_chacha12_next_u32(ro_chacha* c);
float ro_chacha12_next_f32(ro_chacha* c);

/* Source of "rand::random::<f32>()" values.  Three kinds:
 *   counter  : the rt_rng.h stream (shared with the GPU): a sample's draws, or, for the stream
 *              RT_RNG_SCENE_STREAM, the draws of scene construction
 *   scripted : values read from an array, for known-answer tests; counts what it hands out
 *   chacha   : a ChaCha12 generator that outlives the ro_rng (one per worker, like the
 *              reference's thread-local generator), independent of rt_rng.h */
typedef struct ro_rng {
  rt_rng counter;
  const float* script;
  uint32_t script_len;
  uint32_t draws; /* number of values handed out so far */
  ro_chacha* chacha;
  uint32_t scene; /* counter kind: the scene-construction form of the stream (rt_rng_counter_next_f32) */
} ro_rng;

void ro_rng_counter(ro_rng* r, uint32_t seed, uint32_t pixel, uint32_t sample);
void ro_rng_scripted(ro_rng* r, const float* values, uint32_t n);
void ro_rng_chacha(ro_rng* r, ro_chacha* generator);
float ro_random_f32(ro_rng* r);

/* Per-call counters (thread-local accumulation is the caller's business). */
typedef struct ro_counters {
  uint64_t rays;    /* World::hit calls */
  uint64_t samples; /* shoot_ray calls from the pixel loop */
} ro_counters;

#define RO_POW5_LIBM 0 /* reflectance uses powf(x, 5.0f) like the reference */
#define RO_POW5_MUL 1  /* reflectance uses x2=x*x; x4=x2*x2; x4*x (what the GPU does) */

/* ---- glam restatements (crate absent; published algorithm of glam 0.21.2) ---- */
void ro_mat4_look_at_rh(ro_vec3 eye, ro_vec3 center, ro_vec3 up, float out[16]);
void ro_mat4_perspective_rh(float fov_y_radians, float aspect, float z_near, float z_far, float out[16]);
int ro_mat4_inverse(const float m[16], float out[16]);
void ro_mat4_mul_vec4(const float m[16], const float v[4], float out[4]);

/* ---- scene / camera ---- */
void ro_camera_new(ro_vec3 position, ro_vec3 look_at, float fov_deg, float aspect, rt_camera* out);
ro_ray ro_camera_get_ray(const rt_camera* cam, float u, float v);

/* Scene builders.  The *_ex form is create_random_world with its literals as parameters:
 * grid range [lo,hi) on both axes, material thresholds, and whether the three big spheres are
 * all dielectric (ir 1.5).  Buffers must hold (hi-lo)^2+4 spheres and as many materials.
 * Returns the number of spheres written (== number of materials). */
uint32_t ro_create_random_world_ex(uint32_t seed, int lo, int hi, float diffuse_below,
                                   float metal_below, int big_all_dielectric,
                                   rt_sphere* spheres, rt_material* materials);
uint32_t ro_create_random_world(uint32_t seed, rt_sphere* spheres, rt_material* materials);
uint32_t ro_create_simple_world(rt_sphere* spheres, rt_material* materials);

/* ---- intersection and shading ---- */
int ro_sphere_hit(const rt_sphere* s, const ro_ray* ray, float t_min, float t_max, ro_hit* rec);
int ro_world_hit(const ro_world* w, const ro_ray* ray, float t_min, float t_max, ro_hit* rec);
float ro_reflectance(float cosine, float ref_index, int pow5_mode);
ro_vec3 ro_reflect(ro_vec3 input, ro_vec3 normal);
ro_vec3 ro_refract(ro_vec3 uv, ro_vec3 n, float etai_over_etat);
ro_vec3 ro_random_vec3(ro_rng* rng);
ro_vec3 ro_random_in_unit_sphere_vec3(ro_rng* rng);
ro_vec3 ro_random_unit_vec3(ro_rng* rng);
ro_vec3 ro_sky(const ro_ray* ray);

/* World::shoot_ray, recursive exactly like the reference. */
ro_vec3 ro_shoot_ray(const ro_world* w, const ro_ray* ray, uint32_t depth, ro_rng* rng,
                     ro_counters* ctr);
/* The same estimator as a loop with a running throughput (the order the GPU multiplies in).
 * Consumes random numbers in the identical order; differs from ro_shoot_ray only in the
 * association of the colour products. */
ro_vec3 ro_shoot_ray_iter(const ro_world* w, const ro_ray* ray, uint32_t depth, ro_rng* rng,
                          int pow5_mode, ro_counters* ctr);
/* One segment: World::hit plus the scatter decision of shoot_ray's body. */
void ro_bounce(const ro_world* w, const ro_ray* ray, ro_rng* rng, int pow5_mode, rt_ray_out* out);

/* ---- tile renderer ---- */
#define RO_MODE_RECURSIVE 0 /* ro_shoot_ray, powf */
#define RO_MODE_ITER 1      /* ro_shoot_ray_iter, powf */
#define RO_MODE_ITER_MUL 2  /* ro_shoot_ray_iter, pow5 by multiplication: bit-comparable to the GPU */

typedef struct ro_job_data {
  uint32_t image_width, image_height;
  uint32_t max_depth, samples_per_pixel;
  uint32_t seed;
  int mode; /* RO_MODE_* */
  const rt_camera* camera;
  const ro_world* world;
} ro_job_data;

/* execute_job: out receives tile->width*tile->height RGB sums, row-major inside the tile. */
void ro_execute_job(const rt_tile* job, const ro_job_data* data, float* out, ro_counters* ctr);
/* execute_job's sample loop continued from the sums in `inout`: samples
 * [sample_begin, sample_begin + n_samples) are added in order. */
void ro_accumulate_job(const rt_tile* job, const ro_job_data* data, uint32_t sample_begin,
                       uint32_t n_samples, float* inout, ro_counters* ctr);
/* create_job_queue: returns the number of jobs; writes at most cap of them. */
uint32_t ro_create_job_queue(uint32_t image_width, uint32_t image_height, uint32_t tile_width,
                             uint32_t tile_height, rt_tile* out, uint32_t cap);
void ro_debug_write_job_queue_to_framebuffer(uint32_t image_width, uint32_t image_height,
                                             const rt_tile* jobs, uint32_t n_jobs, float* out_rgb);
/* compose_final_framebuffer over concatenated tile sums. */
void ro_compose_final_framebuffer(uint32_t image_width, uint32_t image_height,
                                  uint32_t samples_per_pixel, const rt_tile* jobs,
                                  uint32_t n_jobs, const float* tile_sums, float* out_rgb);
/* dispatch_work_cpu_singlethreaded / _multithreaded.  tile_sums may be NULL.  Return the
 * seconds spent in the tile loop (resolve excluded, like the reference's timer). */
double ro_dispatch_singlethreaded(const rt_tile* jobs, uint32_t n_jobs, const ro_job_data* data,
                                  float* out_rgb, float* tile_sums, ro_counters* ctr);
double ro_dispatch_multithreaded(const rt_tile* jobs, uint32_t n_jobs, const ro_job_data* data,
                                 uint32_t num_threads, float* out_rgb, float* tile_sums,
                                 ro_counters* ctr);

/* Per-pixel first and second moments of the estimator, for statistical comparisons: the
 * reference's recursive shoot_ray (powf) over `samples_per_pixel` samples per pixel of the
 * whole image, drawing either from the counter stream (rng_kind 0; data->seed, pixel, sample)
 * or from ChaCha12 generators (rng_kind 1; one generator per image row keyed by data->seed and
 * the row, consumed sample after sample and pixel after pixel like a worker thread's
 * generator in the reference).  sum and sumsq receive W*H*3 doubles: the sum of the samples'
 * radiance and of its square.  Rows are spread over num_threads threads; the result does not
 * depend on the thread count. */
#define RO_RNG_COUNTER 0
#define RO_RNG_CHACHA12 1
void ro_render_moments(const ro_job_data* data, int rng_kind, uint32_t num_threads, double* sum,
                       double* sumsq, ro_counters* ctr);

/* write_framebuffer_to_image: 24-bit BMP, y flipped, (c*255.0) as u8.  Returns 0 on success. */
int ro_write_framebuffer_to_image(const char* path, uint32_t image_width, uint32_t image_height,
                                  const float* rgb);
/* The `as u8` cast of Rust: saturating, NaN -> 0. */
uint8_t ro_f32_as_u8(float v);

#ifdef __cplusplus
}
#endif

#endif /* RT_ORACLE_H */
