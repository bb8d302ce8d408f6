/* rt_oracle.c - CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY.
 *
 * See rt_oracle.h for the rules on who may load this and for the "parity unpinned" statement.
 * Every function names the lines of the reference (Nanoteck137/ray_tracing) it follows.
 * Arithmetic is IEEE f32 in the reference's operation order; build with -ffp-contract=off and
 * without -ffast-math (rustc neither contracts to FMA nor reassociates).
 */
#define _POSIX_C_SOURCE 200809L
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------
 * glam::f32::Vec3 (scalar implementation) operations used by the reference
 * ---------------------------------------------------------------------------------------- */
static inline ro_vec3 v3(float x, float y, float z) { ro_vec3 r = {x, y, z}; return r; }
static inline ro_vec3 v3_add(ro_vec3 a, ro_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ro_vec3 v3_sub(ro_vec3 a, ro_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ro_vec3 v3_mul(ro_vec3 a, ro_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ro_vec3 v3_scale(ro_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline ro_vec3 v3_div(ro_vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline ro_vec3 v3_neg(ro_vec3 a) { return v3(-a.x, -a.y, -a.z); }
/* glam: dot = (x*x') + (y*y') + (z*z'), left to right */
static inline float v3_dot(ro_vec3 a, ro_vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline float v3_length_squared(ro_vec3 a) { return v3_dot(a, a); }
/* glam: normalize(v) = v * (1.0 / sqrt(dot(v, v))) */
static inline ro_vec3 v3_normalize(ro_vec3 a) {
  float len = sqrtf(v3_dot(a, a));
  float rcp = 1.0f / len;
  return v3_scale(a, rcp);
}
static inline ro_vec3 v3_cross(ro_vec3 a, ro_vec3 b) {
  return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

/* ------------------------------------------------------------------------------------------
 * random numbers: rand::random::<f32>() call sites (src/cpu/job.rs:50-51, src/main.rs:215,
 * 267-269, 342-361)
 * ---------------------------------------------------------------------------------------- */
void ro_rng_counter(ro_rng* r, uint32_t seed, uint32_t pixel, uint32_t sample) {
  r->scene = pixel == RT_RNG_SCENE_STREAM;
  r->counter = r->scene ? rt_rng_counter_init(seed, pixel, sample) : rt_rng_init(seed, pixel, sample);
  r->script = NULL;
  r->script_len = 0;
  r->draws = 0;
  r->chacha = NULL;
}

void ro_rng_scripted(ro_rng* r, const float* values, uint32_t n) {
  r->counter = rt_rng_init(0, 0, 0);
  r->script = values;
  r->script_len = n;
  r->draws = 0;
  r->chacha = NULL;
  r->scene = 0;
}

void ro_rng_chacha(ro_rng* r, ro_chacha* generator) {
  r->counter = rt_rng_init(0, 0, 0);
  r->script = NULL;
  r->script_len = 0;
  r->draws = 0;
  r->chacha = generator;
  r->scene = 0;
}

float ro_random_f32(ro_rng* r) {
  float v;
  if (r->chacha != NULL) {
    v = ro_chacha12_next_f32(r->chacha);
  } else if (r->script != NULL && r->draws < r->script_len) {
    v = r->script[r->draws];
  } else if (r->scene) {
    v = rt_rng_counter_next_f32(&r->counter);
  } else {
    v = rt_rng_next_f32(&r->counter);
  }
  r->draws++;
  return v;
}

/* ------------------------------------------------------------------------------------------
 * glam::f32::Mat4, column-major: m[4*col + row]
 * ---------------------------------------------------------------------------------------- */

/* glam 0.21.2 Mat4::look_at_rh(eye, center, up) = look_to_rh(eye, center - eye, up);
 * call site src/main.rs:59-60 */
void ro_mat4_look_at_rh(ro_vec3 eye, ro_vec3 center, ro_vec3 up, float out[16]) {
  ro_vec3 f = v3_normalize(v3_sub(center, eye));
  ro_vec3 s = v3_normalize(v3_cross(f, up));
  ro_vec3 u = v3_cross(s, f);
  out[0] = s.x; out[1] = u.x; out[2] = -f.x; out[3] = 0.0f;
  out[4] = s.y; out[5] = u.y; out[6] = -f.y; out[7] = 0.0f;
  out[8] = s.z; out[9] = u.z; out[10] = -f.z; out[11] = 0.0f;
  out[12] = -v3_dot(s, eye); out[13] = -v3_dot(u, eye); out[14] = v3_dot(f, eye); out[15] = 1.0f;
}

/* glam 0.21.2 Mat4::perspective_rh (depth range 0..1); call site src/main.rs:61-62 */
void ro_mat4_perspective_rh(float fov_y_radians, float aspect, float z_near, float z_far,
                            float out[16]) {
  float sin_fov = sinf(0.5f * fov_y_radians);
  float cos_fov = cosf(0.5f * fov_y_radians);
  float h = cos_fov / sin_fov;
  float w = h / aspect;
  float r = z_far / (z_near - z_far);
  memset(out, 0, 16 * sizeof(float));
  out[0] = w;
  out[5] = h;
  out[10] = r;
  out[11] = -1.0f;
  out[14] = r * z_near;
}

/* glam 0.21.2 Mat4::inverse, scalar implementation, operation for operation; call sites
 * src/main.rs:65-66.  glam is not under /root/reference (Cargo.toml:10, Cargo.lock:277-279); this
 * restates the crate's published scalar algorithm (the cofactor scheme it shares with GLM's
 * compute_inverse): eighteen 2x2 sub-determinants `coef`, four cofactor columns
 * inv_k = v_a * fac_x - v_b * fac_y + v_c * fac_z evaluated left to right, signs applied by a
 * multiplication with (+1, -1, +1, -1) / (-1, +1, -1, +1), the determinant as the dot product of
 * the first column with the first ROW of the cofactor matrix summed x + y + z + w left to right,
 * and a final multiplication of every entry by 1 / det.  mCR below is column C, row R, as in glam.
 * The x86-64 build of the crate takes its SSE2 path, which forms the same products and sums but
 * adds the determinant's four terms as (x + z) + (y + w): not pinned here (an ulp-level effect).
 * Returns 0 if the matrix is singular (glam asserts instead). */
int ro_mat4_inverse(const float m[16], float out[16]) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
  const float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
  const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
  const float m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];

  const float coef00 = m22 * m33 - m32 * m23;
  const float coef02 = m12 * m33 - m32 * m13;
  const float coef03 = m12 * m23 - m22 * m13;
  const float coef04 = m21 * m33 - m31 * m23;
  const float coef06 = m11 * m33 - m31 * m13;
  const float coef07 = m11 * m23 - m21 * m13;
  const float coef08 = m21 * m32 - m31 * m22;
  const float coef10 = m11 * m32 - m31 * m12;
  const float coef11 = m11 * m22 - m21 * m12;
  const float coef12 = m20 * m33 - m30 * m23;
  const float coef14 = m10 * m33 - m30 * m13;
  const float coef15 = m10 * m23 - m20 * m13;
  const float coef16 = m20 * m32 - m30 * m22;
  const float coef18 = m10 * m32 - m30 * m12;
  const float coef19 = m10 * m22 - m20 * m12;
  const float coef20 = m20 * m31 - m30 * m21;
  const float coef22 = m10 * m31 - m30 * m11;
  const float coef23 = m10 * m21 - m20 * m11;

  const float fac0[4] = {coef00, coef00, coef02, coef03};
  const float fac1[4] = {coef04, coef04, coef06, coef07};
  const float fac2[4] = {coef08, coef08, coef10, coef11};
  const float fac3[4] = {coef12, coef12, coef14, coef15};
  const float fac4[4] = {coef16, coef16, coef18, coef19};
  const float fac5[4] = {coef20, coef20, coef22, coef23};
  const float vec0[4] = {m10, m00, m00, m00};
  const float vec1[4] = {m11, m01, m01, m01};
  const float vec2[4] = {m12, m02, m02, m02};
  const float vec3[4] = {m13, m03, m03, m03};
  const float sign_a[4] = {1.0f, -1.0f, 1.0f, -1.0f};
  const float sign_b[4] = {-1.0f, 1.0f, -1.0f, 1.0f};

  float inv[16];   /* the cofactor matrix, column-major */
  for (int i = 0; i < 4; ++i) {
    const float inv0 = (vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i];
    const float inv1 = (vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i];
    const float inv2 = (vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i];
    const float inv3 = (vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i];
    inv[i] = inv0 * sign_a[i];
    inv[4 + i] = inv1 * sign_b[i];
    inv[8 + i] = inv2 * sign_a[i];
    inv[12 + i] = inv3 * sign_b[i];
  }
  /* x_axis * (first row of the cofactor matrix), summed left to right */
  const float dot1 = ((m00 * inv[0] + m01 * inv[4]) + m02 * inv[8]) + m03 * inv[12];
  if (dot1 == 0.0f) return 0;
  const float rcp_det = 1.0f / dot1;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * rcp_det;
  return 1;
}

/* glam Mat4 * Vec4 = ((col0*x + col1*y) + col2*z) + col3*w, per component, no FMA.
 * Call sites src/main.rs:71,74,77 */
void ro_mat4_mul_vec4(const float m[16], const float v[4], float out[4]) {
  for (int i = 0; i < 4; ++i) {
    float r = m[i] * v[0];
    r = r + m[4 + i] * v[1];
    r = r + m[8 + i] * v[2];
    r = r + m[12 + i] * v[3];
    out[i] = r;
  }
}

/* ------------------------------------------------------------------------------------------
 * Camera, src/main.rs:57-84
 * ---------------------------------------------------------------------------------------- */

/* Camera::new, src/main.rs:58-68.  f32::to_radians is x * (PI_f32 / 180.0_f32). */
void ro_camera_new(ro_vec3 position, ro_vec3 look_at, float fov_deg, float aspect, rt_camera* out) {
  float view[16], proj[16];
  const float pi_f32 = 3.14159265358979323846f;
  float fov_rad = fov_deg * (pi_f32 / 180.0f);
  ro_mat4_look_at_rh(position, look_at, v3(0.0f, 1.0f, 0.0f), view);
  ro_mat4_perspective_rh(fov_rad, aspect, 0.1f, 100.0f, proj);
  ro_mat4_inverse(view, out->view_inv);
  ro_mat4_inverse(proj, out->proj_inv);
}

/* Camera::get_ray, src/main.rs:70-83.  The w component is dropped (truncate), never divided. */
ro_ray ro_camera_get_ray(const rt_camera* cam, float u, float v) {
  float in[4], o4[4], d4[4], w4[4];
  in[0] = 0.0f; in[1] = 0.0f; in[2] = 0.0f; in[3] = 1.0f;
  ro_mat4_mul_vec4(cam->view_inv, in, o4);
  in[0] = u; in[1] = v; in[2] = 0.0f; in[3] = 1.0f;
  ro_mat4_mul_vec4(cam->proj_inv, in, d4);
  in[0] = d4[0]; in[1] = d4[1]; in[2] = d4[2]; in[3] = 0.0f;
  ro_mat4_mul_vec4(cam->view_inv, in, w4);
  ro_ray r;
  r.origin = v3(o4[0], o4[1], o4[2]);
  r.dir = v3_normalize(v3(w4[0], w4[1], w4[2]));
  return r;
}

/* ------------------------------------------------------------------------------------------
 * Intersection, src/main.rs:97-106, 127-160, 174-189
 * ---------------------------------------------------------------------------------------- */

/* Sphere::hit, src/main.rs:128-159 (+ HitRecord::set_face_normal, src/main.rs:98-105) */
int ro_sphere_hit(const rt_sphere* s, const ro_ray* ray, float t_min, float t_max, ro_hit* rec) {
  ro_vec3 position = v3(s->x, s->y, s->z);
  ro_vec3 oc = v3_sub(ray->origin, position);
  float a = v3_length_squared(ray->dir);
  float half_b = v3_dot(oc, ray->dir);
  float c = v3_length_squared(oc) - s->radius * s->radius;
  float discriminant = half_b * half_b - a * c;

  if (discriminant < 0.0f) return 0;

  float sqrtd = sqrtf(discriminant);

  float root = (-half_b - sqrtd) / a;
  if (root < t_min || t_max < root) {
    root = (-half_b + sqrtd) / a;
    if (root < t_min || t_max < root) return 0;
  }

  rec->t = root;
  rec->point = v3_add(ray->origin, v3_scale(ray->dir, rec->t)); /* Ray::at, src/main.rs:46-48 */
  ro_vec3 normal = v3_div(v3_sub(rec->point, position), s->radius);
  rec->front_face = v3_dot(ray->dir, normal) < 0.0f;
  rec->normal = rec->front_face ? normal : v3_neg(normal);
  rec->material_id = s->material_id;
  return 1;
}

/* World::hit, src/main.rs:175-189: storage order, shrinking t_max; a later sphere at exactly
 * the same t replaces the earlier one (the rejection test is the strict `t_max < root`). */
int ro_world_hit(const ro_world* w, const ro_ray* ray, float t_min, float t_max, ro_hit* rec) {
  float closest = t_max;
  int found = 0;
  for (uint32_t i = 0; i < w->n_spheres; ++i) {
    ro_hit record;
    memset(&record, 0, sizeof record);
    if (ro_sphere_hit(&w->spheres[i], ray, t_min, closest, &record)) {
      closest = record.t;
      record.sphere = i;
      *rec = record;
      found = 1;
    }
  }
  return found;
}

/* ------------------------------------------------------------------------------------------
 * Scatter math, src/main.rs:162-166, 266-286, 312-323
 * ---------------------------------------------------------------------------------------- */

/* reflectance, src/main.rs:162-166 */
float ro_reflectance(float cosine, float ref_index, int pow5_mode) {
  float r0 = (1.0f - ref_index) / (1.0f + ref_index);
  r0 = r0 * r0;
  float x = 1.0f - cosine;
  float p;
  if (pow5_mode == RO_POW5_MUL) {
    float x2 = x * x;
    float x4 = x2 * x2;
    p = x4 * x;
  } else {
    p = powf(x, 5.0f);
  }
  return r0 + (1.0f - r0) * p;
}

/* reflect, src/main.rs:312-315 */
ro_vec3 ro_reflect(ro_vec3 input, ro_vec3 normal) {
  float dn = 2.0f * v3_dot(input, normal);
  return v3_sub(input, v3_scale(normal, dn));
}

/* refract, src/main.rs:317-323 */
ro_vec3 ro_refract(ro_vec3 uv, ro_vec3 n, float etai_over_etat) {
  float cos_theta = fminf(v3_dot(v3_neg(uv), n), 1.0f);
  ro_vec3 inner = v3_add(uv, v3_scale(n, cos_theta));
  ro_vec3 out_perp = v3_scale(inner, etai_over_etat);
  float k = -sqrtf(fabsf(1.0f - v3_length_squared(out_perp)));
  ro_vec3 out_parallel = v3_scale(n, k);
  return v3_add(out_perp, out_parallel);
}

/* random_vec3, src/main.rs:266-271 */
ro_vec3 ro_random_vec3(ro_rng* rng) {
  float x = ro_random_f32(rng) * 2.0f - 1.0f;
  float y = ro_random_f32(rng) * 2.0f - 1.0f;
  float z = ro_random_f32(rng) * 2.0f - 1.0f;
  return v3(x, y, z);
}

/* random_in_unit_sphere_vec3, src/main.rs:273-282 */
ro_vec3 ro_random_in_unit_sphere_vec3(ro_rng* rng) {
  for (;;) {
    ro_vec3 v = ro_random_vec3(rng);
    if (v3_length_squared(v) >= 1.0f) continue;
    return v;
  }
}

/* random_unit_vec3, src/main.rs:284-286 */
ro_vec3 ro_random_unit_vec3(ro_rng* rng) {
  return v3_normalize(ro_random_in_unit_sphere_vec3(rng));
}

/* Sky gradient, src/main.rs:256-262 */
ro_vec3 ro_sky(const ro_ray* ray) {
  ro_vec3 dir = v3_normalize(ray->dir);
  float t = 0.5f * (dir.y + 1.0f);
  ro_vec3 color1 = v3(1.0f, 1.0f, 1.0f);
  ro_vec3 color2 = v3(0.5f, 0.7f, 1.0f);
  return v3_add(v3_scale(color1, 1.0f - t), v3_scale(color2, t));
}

/* ------------------------------------------------------------------------------------------
 * World::shoot_ray, src/main.rs:191-263, as written: recursive
 * ---------------------------------------------------------------------------------------- */
ro_vec3 ro_shoot_ray(const ro_world* w, const ro_ray* ray, uint32_t depth, ro_rng* rng,
                     ro_counters* ctr) {
  if (depth == 0) return v3(0.0f, 0.0f, 0.0f);

  ro_hit record;
  if (ctr) ctr->rays++;
  if (ro_world_hit(w, ray, 0.001f, FLT_MAX, &record)) {
    const rt_material* material = &w->materials[record.material_id];
    ro_vec3 material_color = v3(material->r, material->g, material->b);
    ro_vec3 final_color = material_color;

    if (material->flags & RT_MAT_DIELECTRIC) {
      float ir = material->ir;
      float refraction_ratio = record.front_face ? 1.0f / ir : ir;

      ro_vec3 d = v3_normalize(ray->dir);
      float cos_theta = fminf(v3_dot(v3_neg(d), record.normal), 1.0f);
      float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);

      int cannot_refract = (refraction_ratio * sin_theta) > 1.0f;

      /* `||` short-circuits: no random number is drawn when cannot_refract is true */
      ro_vec3 dir;
      if (cannot_refract ||
          ro_reflectance(cos_theta, refraction_ratio, RO_POW5_LIBM) > ro_random_f32(rng)) {
        dir = ro_reflect(d, record.normal);
      } else {
        dir = ro_refract(d, record.normal, refraction_ratio);
      }

      ro_ray new_ray = {record.point, dir};
      ro_vec3 attenuation = v3(1.0f, 1.0f, 1.0f);
      final_color = v3_mul(attenuation, ro_shoot_ray(w, &new_ray, depth - 1, rng, ctr));
    } else if (material->flags & RT_MAT_METALLIC) {
      ro_vec3 reflected = ro_reflect(ray->dir, record.normal);
      float fuzz = 1.0f - material->metallic_strength;
      ro_vec3 ray_dir = v3_add(reflected, v3_scale(ro_random_in_unit_sphere_vec3(rng), fuzz));
      ro_ray scattered_ray = {record.point, ray_dir};

      ro_vec3 color;
      if (v3_dot(reflected, record.normal) > 0.0f) {
        ro_vec3 attenuation = material_color;
        color = v3_mul(attenuation, ro_shoot_ray(w, &scattered_ray, depth - 1, rng, ctr));
      } else {
        color = v3(0.0f, 0.0f, 0.0f);
      }
      final_color = v3_mul(final_color, color);
    } else {
      ro_vec3 dir = v3_add(record.normal, ro_random_unit_vec3(rng));
      ro_ray new_ray = {record.point, dir};
      ro_vec3 diffuse = ro_shoot_ray(w, &new_ray, depth - 1, rng, ctr);
      final_color = v3_mul(final_color, diffuse);
    }
    return final_color;
  }
  return ro_sky(ray);
}

/* ------------------------------------------------------------------------------------------
 * One segment of shoot_ray (World::hit + the material branch), and the loop built from it.
 * Follows src/main.rs:196-251 for the decisions and the order of random draws.
 * ---------------------------------------------------------------------------------------- */
void ro_bounce(const ro_world* w, const ro_ray* ray, ro_rng* rng, int pow5_mode, rt_ray_out* out) {
  memset(out, 0, sizeof *out);
  uint32_t draws0 = rng->draws;
  ro_hit record;
  if (!ro_world_hit(w, ray, 0.001f, FLT_MAX, &record)) {
    ro_vec3 sky = ro_sky(ray);
    out->hit = 0;
    out->kind = RT_KIND_MISS;
    out->ar = sky.x; out->ag = sky.y; out->ab = sky.z;
    return;
  }
  const rt_material* material = &w->materials[record.material_id];
  out->hit = 1;
  out->sphere = record.sphere;
  out->material = record.material_id;
  out->front_face = (uint32_t)record.front_face;
  out->t = record.t;
  out->px = record.point.x; out->py = record.point.y; out->pz = record.point.z;
  out->nx = record.normal.x; out->ny = record.normal.y; out->nz = record.normal.z;

  ro_vec3 dir;
  ro_vec3 atten;
  uint32_t scattered = 1;
  if (material->flags & RT_MAT_DIELECTRIC) {
    out->kind = RT_KIND_DIELECTRIC;
    float ir = material->ir;
    float refraction_ratio = record.front_face ? 1.0f / ir : ir;
    ro_vec3 d = v3_normalize(ray->dir);
    float cos_theta = fminf(v3_dot(v3_neg(d), record.normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    int cannot_refract = (refraction_ratio * sin_theta) > 1.0f;
    if (cannot_refract ||
        ro_reflectance(cos_theta, refraction_ratio, pow5_mode) > ro_random_f32(rng)) {
      dir = ro_reflect(d, record.normal);
    } else {
      dir = ro_refract(d, record.normal, refraction_ratio);
    }
    atten = v3(1.0f, 1.0f, 1.0f);
  } else if (material->flags & RT_MAT_METALLIC) {
    out->kind = RT_KIND_METAL;
    ro_vec3 c = v3(material->r, material->g, material->b);
    ro_vec3 reflected = ro_reflect(ray->dir, record.normal);
    float fuzz = 1.0f - material->metallic_strength;
    dir = v3_add(reflected, v3_scale(ro_random_in_unit_sphere_vec3(rng), fuzz));
    if (v3_dot(reflected, record.normal) > 0.0f) {
      atten = v3_mul(c, c); /* the albedo is applied twice, src/main.rs:199,236-237,242 */
    } else {
      atten = v3(0.0f, 0.0f, 0.0f);
      scattered = 0;
    }
  } else {
    out->kind = RT_KIND_DIFFUSE;
    dir = v3_add(record.normal, ro_random_unit_vec3(rng));
    atten = v3(material->r, material->g, material->b);
  }
  out->scattered = scattered;
  out->sx = dir.x; out->sy = dir.y; out->sz = dir.z;
  out->ar = atten.x; out->ag = atten.y; out->ab = atten.z;
  out->draws = rng->draws - draws0;
}

ro_vec3 ro_shoot_ray_iter(const ro_world* w, const ro_ray* ray, uint32_t depth, ro_rng* rng,
                          int pow5_mode, ro_counters* ctr) {
  ro_vec3 throughput = v3(1.0f, 1.0f, 1.0f);
  ro_ray cur = *ray;
  for (;;) {
    if (depth == 0) return v3(0.0f, 0.0f, 0.0f);
    rt_ray_out b;
    if (ctr) ctr->rays++;
    ro_bounce(w, &cur, rng, pow5_mode, &b);
    if (!b.hit) return v3_mul(throughput, v3(b.ar, b.ag, b.ab));
    if (!b.scattered) return v3(0.0f, 0.0f, 0.0f);
    if (b.kind != RT_KIND_DIELECTRIC) throughput = v3_mul(throughput, v3(b.ar, b.ag, b.ab));
    cur.origin = v3(b.px, b.py, b.pz);
    cur.dir = v3(b.sx, b.sy, b.sz);
    depth--;
  }
}

/* ------------------------------------------------------------------------------------------
 * Tile renderer, src/cpu/job.rs and src/cpu/mod.rs
 * ---------------------------------------------------------------------------------------- */

/* execute_job, src/cpu/job.rs:36-74.  The two jitter draws and every draw of the path come
 * from the stream keyed by (seed, pixel, sample). */
void ro_execute_job(const rt_tile* job, const ro_job_data* data, float* out, ro_counters* ctr) {
  uint32_t image_width = data->image_width;
  uint32_t image_height = data->image_height;
  size_t k = 0;
  for (uint32_t yoff = 0; yoff < job->height; ++yoff) {
    for (uint32_t xoff = 0; xoff < job->width; ++xoff) {
      uint32_t x = job->x + xoff;
      uint32_t y = job->y + yoff;
      ro_vec3 pixel_color = v3(0.0f, 0.0f, 0.0f);
      for (uint32_t sample = 0; sample < data->samples_per_pixel; ++sample) {
        ro_rng rng;
        ro_rng_counter(&rng, data->seed, y * image_width + x, sample);
        float u = ((float)x + ro_random_f32(&rng)) / (float)image_width;
        float v = ((float)y + ro_random_f32(&rng)) / (float)image_height;
        u = u * 2.0f - 1.0f;
        v = v * 2.0f - 1.0f;
        ro_ray ray = ro_camera_get_ray(data->camera, u, v);
        ro_vec3 color;
        if (data->mode == RO_MODE_RECURSIVE) {
          color = ro_shoot_ray(data->world, &ray, data->max_depth, &rng, ctr);
        } else {
          color = ro_shoot_ray_iter(data->world, &ray, data->max_depth, &rng,
                                    data->mode == RO_MODE_ITER_MUL ? RO_POW5_MUL : RO_POW5_LIBM,
                                    ctr);
        }
        if (ctr) ctr->samples++;
        pixel_color = v3_add(pixel_color, color);
      }
      out[3 * k + 0] = pixel_color.x;
      out[3 * k + 1] = pixel_color.y;
      out[3 * k + 2] = pixel_color.z;
      k++;
    }
  }
}

/* The sample loop of execute_job (src/cpu/job.rs:48-61) carried on from an existing sum:
 * `inout` holds the tile's sums after samples [0, sample_begin) and receives them after samples
 * [0, sample_begin + n_samples), `pixel_color += color` continuing in sample order.  The
 * reference renders in one pass; this is the same loop cut in two, so running it over
 * [0, a) and then [a, b) gives bit for bit what ro_execute_job gives for b samples. */
void ro_accumulate_job(const rt_tile* job, const ro_job_data* data, uint32_t sample_begin,
                       uint32_t n_samples, float* inout, ro_counters* ctr) {
  uint32_t image_width = data->image_width;
  uint32_t image_height = data->image_height;
  size_t k = 0;
  for (uint32_t yoff = 0; yoff < job->height; ++yoff) {
    for (uint32_t xoff = 0; xoff < job->width; ++xoff) {
      uint32_t x = job->x + xoff;
      uint32_t y = job->y + yoff;
      ro_vec3 pixel_color = v3(inout[3 * k + 0], inout[3 * k + 1], inout[3 * k + 2]);
      for (uint32_t sample = sample_begin; sample < sample_begin + n_samples; ++sample) {
        ro_rng rng;
        ro_rng_counter(&rng, data->seed, y * image_width + x, sample);
        float u = ((float)x + ro_random_f32(&rng)) / (float)image_width;
        float v = ((float)y + ro_random_f32(&rng)) / (float)image_height;
        u = u * 2.0f - 1.0f;
        v = v * 2.0f - 1.0f;
        ro_ray ray = ro_camera_get_ray(data->camera, u, v);
        ro_vec3 color;
        if (data->mode == RO_MODE_RECURSIVE) {
          color = ro_shoot_ray(data->world, &ray, data->max_depth, &rng, ctr);
        } else {
          color = ro_shoot_ray_iter(data->world, &ray, data->max_depth, &rng,
                                    data->mode == RO_MODE_ITER_MUL ? RO_POW5_MUL : RO_POW5_LIBM,
                                    ctr);
        }
        if (ctr) ctr->samples++;
        pixel_color = v3_add(pixel_color, color);
      }
      inout[3 * k + 0] = pixel_color.x;
      inout[3 * k + 1] = pixel_color.y;
      inout[3 * k + 2] = pixel_color.z;
      k++;
    }
  }
}

/* create_job_queue, src/cpu/job.rs:76-152: full tiles row-major, bottom strip, right strip,
 * corner.  The corner's start_y uses tile_width (src/cpu/job.rs:140), kept as written. */
uint32_t ro_create_job_queue(uint32_t image_width, uint32_t image_height, uint32_t tile_width,
                             uint32_t tile_height, rt_tile* out, uint32_t cap) {
  uint32_t n = 0;
#define RO_PUSH(X, Y, W, H)                                         \
  do {                                                              \
    if (n < cap) { out[n].x = (X); out[n].y = (Y); out[n].width = (W); out[n].height = (H); } \
    n++;                                                            \
  } while (0)
  for (uint32_t tile_y = 0; tile_y < image_height / tile_height; ++tile_y)
    for (uint32_t tile_x = 0; tile_x < image_width / tile_width; ++tile_x)
      RO_PUSH(tile_x * tile_width, tile_y * tile_height, tile_width, tile_height);

  uint32_t num_tiles_x = image_width / tile_width;
  uint32_t num_tiles_y = image_height / tile_height;

  if ((image_height % tile_height) > 0) {
    for (uint32_t i = 0; i < num_tiles_x; ++i) {
      uint32_t tw = tile_width;
      uint32_t th = image_height % tile_height;
      uint32_t start_y = num_tiles_y * tile_height;
      RO_PUSH(i * tw, start_y, tw, th);
    }
  }
  if ((image_width % tile_width) > 0) {
    for (uint32_t i = 0; i < num_tiles_y; ++i) {
      uint32_t tw = image_width % tile_width;
      uint32_t th = tile_height;
      uint32_t start_x = num_tiles_x * tile_width;
      RO_PUSH(start_x, i * th, tw, th);
    }
  }
  uint32_t tw = image_width % tile_width;
  uint32_t th = image_height % tile_height;
  if (tw != 0 && th != 0) {
    uint32_t start_x = num_tiles_x * tile_width;
    uint32_t start_y = num_tiles_y * tile_width;
    RO_PUSH(start_x, start_y, tw, th);
  }
#undef RO_PUSH
  return n;
}

/* debug_write_job_queue_to_framebuffer, src/cpu/job.rs:154-176 */
void ro_debug_write_job_queue_to_framebuffer(uint32_t image_width, uint32_t image_height,
                                             const rt_tile* jobs, uint32_t n_jobs, float* out_rgb) {
  memset(out_rgb, 0, (size_t)image_width * image_height * 3 * sizeof(float));
  for (uint32_t j = 0; j < n_jobs; ++j) {
    const rt_tile* job = &jobs[j];
    for (uint32_t yoff = 0; yoff < job->height; ++yoff) {
      for (uint32_t xoff = 0; xoff < job->width; ++xoff) {
        size_t px = job->x + xoff, py = job->y + yoff;
        float r = (float)xoff / (float)job->width;
        float g = (float)yoff / (float)job->height;
        float* o = out_rgb + 3 * (px + py * image_width);
        o[0] = r; o[1] = g; o[2] = 0.0f;
      }
    }
  }
}

/* f32::clamp of Rust: NaN stays NaN */
static inline float rust_clamp(float v, float lo, float hi) {
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}

/* compose_final_framebuffer, src/cpu/mod.rs:51-83 */
void ro_compose_final_framebuffer(uint32_t image_width, uint32_t image_height,
                                  uint32_t samples_per_pixel, const rt_tile* jobs,
                                  uint32_t n_jobs, const float* tile_sums, float* out_rgb) {
  memset(out_rgb, 0, (size_t)image_width * image_height * 3 * sizeof(float));
  size_t base = 0;
  for (uint32_t j = 0; j < n_jobs; ++j) {
    const rt_tile* result = &jobs[j];
    for (uint32_t yoff = 0; yoff < result->height; ++yoff) {
      for (uint32_t xoff = 0; xoff < result->width; ++xoff) {
        size_t x = result->x + xoff, y = result->y + yoff;
        const float* color = tile_sums + 3 * (base + xoff + (size_t)yoff * result->width);
        float scale = 1.0f / (float)samples_per_pixel;
        float r = sqrtf(color[0] * scale);
        float g = sqrtf(color[1] * scale);
        float b = sqrtf(color[2] * scale);
        float* o = out_rgb + 3 * (x + y * image_width);
        o[0] = rust_clamp(r, 0.0f, 0.999f);
        o[1] = rust_clamp(g, 0.0f, 0.999f);
        o[2] = rust_clamp(b, 0.0f, 0.999f);
      }
    }
    base += (size_t)result->width * result->height;
  }
}

static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static size_t* tile_offsets(const rt_tile* jobs, uint32_t n_jobs, size_t* total) {
  size_t* off = (size_t*)malloc(sizeof(size_t) * (n_jobs + 1));
  size_t acc = 0;
  for (uint32_t j = 0; j < n_jobs; ++j) {
    off[j] = acc;
    acc += (size_t)jobs[j].width * jobs[j].height;
  }
  off[n_jobs] = acc;
  *total = acc;
  return off;
}

/* dispatch_work_cpu_singlethreaded, src/cpu/single.rs:10-50 (progress printing left out) */
double ro_dispatch_singlethreaded(const rt_tile* jobs, uint32_t n_jobs, const ro_job_data* data,
                                  float* out_rgb, float* tile_sums, ro_counters* ctr) {
  size_t total;
  size_t* off = tile_offsets(jobs, n_jobs, &total);
  float* sums = tile_sums ? tile_sums : (float*)malloc(sizeof(float) * 3 * total);
  ro_counters local = {0, 0};
  double t0 = now_seconds();
  for (uint32_t j = 0; j < n_jobs; ++j) ro_execute_job(&jobs[j], data, sums + 3 * off[j], &local);
  double elapsed = now_seconds() - t0;
  if (out_rgb)
    ro_compose_final_framebuffer(data->image_width, data->image_height, data->samples_per_pixel,
                                 jobs, n_jobs, sums, out_rgb);
  if (ctr) { ctr->rays += local.rays; ctr->samples += local.samples; }
  if (!tile_sums) free(sums);
  free(off);
  return elapsed;
}

/* dispatch_work_cpu_multithreaded, src/cpu/multi.rs:30-172: num_threads workers pop tiles
 * from one mutex-guarded queue until it is empty.  The main thread joins the workers instead
 * of polling every 500 ms (src/cpu/multi.rs:105-121), so the measured time carries no polling
 * slack.  The timer starts before the workers are spawned and stops after the last join, like
 * src/cpu/multi.rs:146-163; the resolve is outside it. */
typedef struct worker_shared {
  pthread_mutex_t lock;
  uint32_t next_job;
  uint32_t n_jobs;
  const rt_tile* jobs;
  const size_t* off;
  const ro_job_data* data;
  float* sums;
  ro_counters total;
} worker_shared;

static void* worker_main(void* arg) {
  worker_shared* sh = (worker_shared*)arg;
  ro_counters local = {0, 0};
  for (;;) {
    pthread_mutex_lock(&sh->lock);
    uint32_t j = sh->next_job;
    if (j < sh->n_jobs) sh->next_job = j + 1;
    pthread_mutex_unlock(&sh->lock);
    if (j >= sh->n_jobs) break; /* queue empty, src/cpu/multi.rs:57-60 */
    ro_execute_job(&sh->jobs[j], sh->data, sh->sums + 3 * sh->off[j], &local);
  }
  pthread_mutex_lock(&sh->lock);
  sh->total.rays += local.rays;
  sh->total.samples += local.samples;
  pthread_mutex_unlock(&sh->lock);
  return NULL;
}

double ro_dispatch_multithreaded(const rt_tile* jobs, uint32_t n_jobs, const ro_job_data* data,
                                 uint32_t num_threads, float* out_rgb, float* tile_sums,
                                 ro_counters* ctr) {
  if (num_threads == 0) num_threads = 1;
  size_t total;
  size_t* off = tile_offsets(jobs, n_jobs, &total);
  float* sums = tile_sums ? tile_sums : (float*)malloc(sizeof(float) * 3 * total);
  worker_shared sh;
  pthread_mutex_init(&sh.lock, NULL);
  sh.next_job = 0;
  sh.n_jobs = n_jobs;
  sh.jobs = jobs;
  sh.off = off;
  sh.data = data;
  sh.sums = sums;
  sh.total.rays = 0;
  sh.total.samples = 0;
  pthread_t* threads = (pthread_t*)malloc(sizeof(pthread_t) * num_threads);

  double t0 = now_seconds();
  for (uint32_t i = 0; i < num_threads; ++i) pthread_create(&threads[i], NULL, worker_main, &sh);
  for (uint32_t i = 0; i < num_threads; ++i) pthread_join(threads[i], NULL);
  double elapsed = now_seconds() - t0;

  if (out_rgb)
    ro_compose_final_framebuffer(data->image_width, data->image_height, data->samples_per_pixel,
                                 jobs, n_jobs, sums, out_rgb);
  if (ctr) { ctr->rays += sh.total.rays; ctr->samples += sh.total.samples; }
  pthread_mutex_destroy(&sh.lock);
  free(threads);
  if (!tile_sums) free(sums);
  free(off);
  return elapsed;
}

/* Moments of the estimator per pixel (see rt_oracle.h).  The sample loop is execute_job's
 * (src/cpu/job.rs:48-59) with the reference's recursive shoot_ray. */
typedef struct moments_shared {
  pthread_mutex_t lock;
  uint32_t next_row;
  const ro_job_data* data;
  int rng_kind;
  double* sum;
  double* sumsq;
  ro_counters total;
} moments_shared;

static void* moments_worker(void* arg) {
  moments_shared* sh = (moments_shared*)arg;
  const ro_job_data* data = sh->data;
  ro_counters local = {0, 0};
  for (;;) {
    pthread_mutex_lock(&sh->lock);
    uint32_t y = sh->next_row;
    if (y < data->image_height) sh->next_row = y + 1;
    pthread_mutex_unlock(&sh->lock);
    if (y >= data->image_height) break;
    ro_chacha generator;
    ro_chacha12_seed_from(&generator, data->seed, y, 0x726f7773u);
    for (uint32_t x = 0; x < data->image_width; ++x) {
      double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
      for (uint32_t sample = 0; sample < data->samples_per_pixel; ++sample) {
        ro_rng rng;
        if (sh->rng_kind == RO_RNG_CHACHA12) ro_rng_chacha(&rng, &generator);
        else ro_rng_counter(&rng, data->seed, y * data->image_width + x, sample);
        float u = ((float)x + ro_random_f32(&rng)) / (float)data->image_width;
        float v = ((float)y + ro_random_f32(&rng)) / (float)data->image_height;
        u = u * 2.0f - 1.0f;
        v = v * 2.0f - 1.0f;
        ro_ray ray = ro_camera_get_ray(data->camera, u, v);
        ro_vec3 c = ro_shoot_ray(data->world, &ray, data->max_depth, &rng, &local);
        local.samples++;
        s1[0] += c.x; s1[1] += c.y; s1[2] += c.z;
        s2[0] += (double)c.x * c.x; s2[1] += (double)c.y * c.y; s2[2] += (double)c.z * c.z;
      }
      size_t k = 3 * ((size_t)y * data->image_width + x);
      for (int i = 0; i < 3; ++i) { sh->sum[k + i] = s1[i]; sh->sumsq[k + i] = s2[i]; }
    }
  }
  pthread_mutex_lock(&sh->lock);
  sh->total.rays += local.rays;
  sh->total.samples += local.samples;
  pthread_mutex_unlock(&sh->lock);
  return NULL;
}

void ro_render_moments(const ro_job_data* data, int rng_kind, uint32_t num_threads, double* sum,
                       double* sumsq, ro_counters* ctr) {
  if (num_threads == 0) num_threads = 1;
  moments_shared sh;
  pthread_mutex_init(&sh.lock, NULL);
  sh.next_row = 0;
  sh.data = data;
  sh.rng_kind = rng_kind;
  sh.sum = sum;
  sh.sumsq = sumsq;
  sh.total.rays = 0;
  sh.total.samples = 0;
  pthread_t* threads = (pthread_t*)malloc(sizeof(pthread_t) * num_threads);
  for (uint32_t i = 0; i < num_threads; ++i) pthread_create(&threads[i], NULL, moments_worker, &sh);
  for (uint32_t i = 0; i < num_threads; ++i) pthread_join(threads[i], NULL);
  if (ctr) { ctr->rays += sh.total.rays; ctr->samples += sh.total.samples; }
  pthread_mutex_destroy(&sh.lock);
  free(threads);
}

/* ------------------------------------------------------------------------------------------
 * Scenes, src/main.rs:326-519.  The reference draws from the unseeded thread RNG; here the
 * draws come, in the same order, from the stream (seed, RT_RNG_SCENE_STREAM, 0).
 * ---------------------------------------------------------------------------------------- */
static void set_material(rt_material* m, float r, float g, float b, int metallic, float strength,
                         int dielectric, float ir) {
  m->r = r; m->g = g; m->b = b;
  m->flags = (metallic ? RT_MAT_METALLIC : 0u) | (dielectric ? RT_MAT_DIELECTRIC : 0u);
  m->metallic_strength = strength;
  m->ir = ir;
}

static void set_sphere(rt_sphere* s, float x, float y, float z, float radius, uint32_t material) {
  s->x = x; s->y = y; s->z = z; s->radius = radius; s->material_id = material;
}

uint32_t ro_create_random_world_ex(uint32_t seed, int lo, int hi, float diffuse_below,
                                   float metal_below, int big_all_dielectric,
                                   rt_sphere* spheres, rt_material* materials) {
  ro_rng rng;
  ro_rng_counter(&rng, seed, RT_RNG_SCENE_STREAM, 0);
  uint32_t n = 0;
  for (int ai = lo; ai < hi; ++ai) {     /* `for a in -4..4`, src/main.rs:337 */
    for (int bi = lo; bi < hi; ++bi) {   /* `for b in -4..4`, src/main.rs:338 */
      float a = (float)ai;
      float b = (float)bi;
      float x = a + 0.9f * ro_random_f32(&rng);
      float y = 0.2f;
      float z = b + 0.9f * ro_random_f32(&rng);
      float choose_mat = ro_random_f32(&rng);
      if (choose_mat < diffuse_below) {             /* src/main.rs:349-358 */
        ro_vec3 c = v3_scale(ro_random_vec3(&rng), 0.9f);
        set_material(&materials[n], c.x, c.y, c.z, 0, 0.0f, 0, 0.0f);
      } else if (choose_mat < metal_below) {        /* src/main.rs:359-369 */
        ro_vec3 c = ro_random_vec3(&rng);
        float strength = ro_random_f32(&rng);
        set_material(&materials[n], c.x, c.y, c.z, 1, strength, 0, 0.0f);
      } else {                                      /* src/main.rs:370-378 */
        set_material(&materials[n], 1.0f, 1.0f, 1.0f, 0, 0.0f, 1, 1.5f);
      }
      set_sphere(&spheres[n], x, y, z, 0.2f, n);
      n++;
    }
  }
  /* ground, src/main.rs:391-405 */
  set_material(&materials[n], 0.5f, 0.5f, 0.5f, 0, 0.0f, 0, 0.0f);
  set_sphere(&spheres[n], 0.0f, -1000.0f, 0.0f, 1000.0f, n);
  n++;
  /* "Sphere 1", src/main.rs:409-423: dielectric ir 1.4 (its colour is never used) */
  set_material(&materials[n], 0.1f, 0.1f, 0.6f, 0, 0.0f, 1, big_all_dielectric ? 1.5f : 1.4f);
  set_sphere(&spheres[n], 0.0f, 1.0f, 0.0f, 1.0f, n);
  n++;
  /* "Sphere 2", src/main.rs:427-441: diffuse */
  if (big_all_dielectric) set_material(&materials[n], 0.1f, 0.1f, 0.6f, 0, 0.0f, 1, 1.5f);
  else set_material(&materials[n], 0.1f, 0.1f, 0.6f, 0, 0.0f, 0, 0.0f);
  set_sphere(&spheres[n], -4.0f, 1.0f, 0.0f, 1.0f, n);
  n++;
  /* "Sphere 3", src/main.rs:445-460: metal, strength 1.0 */
  if (big_all_dielectric) set_material(&materials[n], 0.7f, 0.6f, 0.5f, 0, 0.0f, 1, 1.5f);
  else set_material(&materials[n], 0.7f, 0.6f, 0.5f, 1, 1.0f, 0, 0.0f);
  set_sphere(&spheres[n], 4.0f, 1.0f, 0.0f, 1.0f, n);
  n++;
  return n;
}

/* create_random_world, src/main.rs:326-466 with its own literals */
uint32_t ro_create_random_world(uint32_t seed, rt_sphere* spheres, rt_material* materials) {
  return ro_create_random_world_ex(seed, -4, 4, 0.8f, 0.95f, 0, spheres, materials);
}

/* create_simple_world, src/main.rs:468-519 */
uint32_t ro_create_simple_world(rt_sphere* spheres, rt_material* materials) {
  set_material(&materials[0], 0.5f, 0.5f, 0.5f, 0, 0.0f, 0, 0.0f);
  set_sphere(&spheres[0], 0.0f, -1000.0f, 0.0f, 1000.0f, 0);
  set_material(&materials[1], 0.7f, 0.6f, 0.5f, 1, 0.1f, 0, 0.0f);
  set_sphere(&spheres[1], 4.0f, 1.0f, 0.0f, 1.0f, 1);
  return 2;
}

/* ------------------------------------------------------------------------------------------
 * write_framebuffer_to_image, src/main.rs:521-546.  The bmp crate (0.5.0, absent) writes an
 * uncompressed 24-bit BITMAPINFOHEADER file, rows bottom-up and padded to 4 bytes, pixels BGR.
 * The reference puts framebuffer row y at image row H-1-y, so framebuffer row 0 is the
 * bottom row of the picture = the first row stored in the file.
 * ---------------------------------------------------------------------------------------- */
uint8_t ro_f32_as_u8(float v) {
  if (v != v) return 0;       /* NaN -> 0 */
  if (v <= 0.0f) return 0;    /* saturate */
  if (v >= 255.0f) return 255;
  return (uint8_t)v;          /* truncation toward zero */
}

static void put_u32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static void put_u16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

int ro_write_framebuffer_to_image(const char* path, uint32_t image_width, uint32_t image_height,
                                  const float* rgb) {
  uint32_t row_bytes = (image_width * 3 + 3) & ~3u;
  uint32_t data_size = row_bytes * image_height;
  uint8_t header[54];
  memset(header, 0, sizeof header);
  header[0] = 'B'; header[1] = 'M';
  put_u32(header + 2, 54 + data_size);
  put_u32(header + 10, 54);
  put_u32(header + 14, 40);
  put_u32(header + 18, image_width);
  put_u32(header + 22, image_height);
  put_u16(header + 26, 1);
  put_u16(header + 28, 24);
  put_u32(header + 34, data_size);
  put_u32(header + 38, 1000);
  put_u32(header + 42, 1000);
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  uint8_t* row = (uint8_t*)calloc(row_bytes, 1);
  int ok = fwrite(header, 1, sizeof header, f) == sizeof header;
  for (uint32_t y = 0; ok && y < image_height; ++y) {
    for (uint32_t x = 0; x < image_width; ++x) {
      const float* pixel = rgb + 3 * ((size_t)x + (size_t)y * image_width);
      row[3 * x + 2] = ro_f32_as_u8(pixel[0] * 255.0f);
      row[3 * x + 1] = ro_f32_as_u8(pixel[1] * 255.0f);
      row[3 * x + 0] = ro_f32_as_u8(pixel[2] * 255.0f);
    }
    ok = fwrite(row, 1, row_bytes, f) == row_bytes;
  }
  free(row);
  if (fclose(f) != 0) ok = 0;
  return ok ? 0 : -1;
}
