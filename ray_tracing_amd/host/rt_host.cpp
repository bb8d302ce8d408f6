// rt_host.cpp - see rt_host.hpp.  Pure host code (g++, no HIP): everything that renders goes
// through the C ABI of librt_hip.so.
#include "rt_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/rt_rng.h"

namespace rt_host {

namespace {

// ---- the few glam operations Camera::new needs (glam 0.21.2, column-major Mat4) -------------
struct Mat4 {
  float c[16];   // c[4*col + row]
};

inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) {
  return Vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline Vec3 normalize(Vec3 v) {   // v * (1 / |v|)
  const float rcp = 1.0f / std::sqrt(dot(v, v));
  return Vec3(v.x * rcp, v.y * rcp, v.z * rcp);
}

// Mat4::look_at_rh(eye, center, up), call site src/main.rs:59-60
Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up) {
  const Vec3 f = normalize(center - eye);
  const Vec3 s = normalize(cross(f, up));
  const Vec3 u = cross(s, f);
  Mat4 m = {{s.x, u.x, -f.x, 0.0f,
             s.y, u.y, -f.y, 0.0f,
             s.z, u.z, -f.z, 0.0f,
             -dot(s, eye), -dot(u, eye), dot(f, eye), 1.0f}};
  return m;
}

// Mat4::perspective_rh(fov_y, aspect, near, far), depth 0..1, call site src/main.rs:61-62
Mat4 perspective_rh(float fov_y_radians, float aspect_ratio, float z_near, float z_far) {
  const float sin_fov = std::sin(0.5f * fov_y_radians);
  const float cos_fov = std::cos(0.5f * fov_y_radians);
  const float h = cos_fov / sin_fov;
  const float w = h / aspect_ratio;
  const float r = z_far / (z_near - z_far);
  Mat4 m;
  std::memset(&m, 0, sizeof m);
  m.c[0] = w;
  m.c[5] = h;
  m.c[10] = r;
  m.c[11] = -1.0f;
  m.c[14] = r * z_near;
  return m;
}

// Mat4::inverse as glam 0.21.2's scalar implementation writes it (call sites src/main.rs:65-66):
// eighteen 2x2 sub-determinants, four cofactor columns each evaluated as a * f - b * g + c * h
// from left to right, the signs applied by a multiplication, the determinant as the first column
// times the first row of the cofactors summed left to right, and every entry times 1 / det.
// c[4 * C + R] is column C, row R.  (The crate's SSE2 path adds the determinant's four terms in
// another order; that build is not reproduced here.)
Mat4 inverse(const Mat4& in) {
  const float* m = in.c;
  // sub[k] = m[a] * m[b] - m[c] * m[d] for the index quadruples below, in glam's order
  static const int kSub[18][4] = {
      {10, 15, 14, 11}, {6, 15, 14, 7}, {6, 11, 10, 7},   // coef00, 02, 03
      {9, 15, 13, 11},  {5, 15, 13, 7}, {5, 11, 9, 7},    // coef04, 06, 07
      {9, 14, 13, 10},  {5, 14, 13, 6}, {5, 10, 9, 6},    // coef08, 10, 11
      {8, 15, 12, 11},  {4, 15, 12, 7}, {4, 11, 8, 7},    // coef12, 14, 15
      {8, 14, 12, 10},  {4, 14, 12, 6}, {4, 10, 8, 6},    // coef16, 18, 19
      {8, 13, 12, 9},   {4, 13, 12, 5}, {4, 9, 8, 5}};    // coef20, 22, 23
  float sub[18];
  for (int k = 0; k < 18; ++k) sub[k] = m[kSub[k][0]] * m[kSub[k][1]] - m[kSub[k][2]] * m[kSub[k][3]];
  // fac_g = (first, first, second, third) of group g; vec_r = (m[4 + r], m[r], m[r], m[r])
  Mat4 cof;
  for (int i = 0; i < 4; ++i) {
    const auto fac = [&](int g) { return sub[3 * g + (i == 0 ? 0 : i - 1)]; };
    const auto vec = [&](int r) { return i == 0 ? m[4 + r] : m[r]; };
    const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
    cof.c[i] = ((vec(1) * fac(0) - vec(2) * fac(1)) + vec(3) * fac(2)) * sa;
    cof.c[4 + i] = ((vec(0) * fac(0) - vec(2) * fac(3)) + vec(3) * fac(4)) * sb;
    cof.c[8 + i] = ((vec(0) * fac(1) - vec(1) * fac(3)) + vec(3) * fac(5)) * sa;
    cof.c[12 + i] = ((vec(0) * fac(2) - vec(1) * fac(4)) + vec(2) * fac(5)) * sb;
  }
  const float det = ((m[0] * cof.c[0] + m[1] * cof.c[4]) + m[2] * cof.c[8]) + m[3] * cof.c[12];
  if (det == 0.0f) throw std::runtime_error("Mat4::inverse: singular matrix");
  const float rcp_det = 1.0f / det;
  Mat4 o;
  for (int k = 0; k < 16; ++k) o.c[k] = cof.c[k] * rcp_det;
  return o;
}

// Sequential draws of scene construction: the stream (seed, RT_RNG_SCENE_STREAM, 0).
struct SceneRng {
  rt_rng g;
  explicit SceneRng(uint32_t seed) : g(rt_rng_counter_init(seed, RT_RNG_SCENE_STREAM, 0)) {}
  float next() { return rt_rng_counter_next_f32(&g); }  // rand::random::<f32>()
  Vec3 vec3() {                                         // random_vec3, src/main.rs:266-271
    const float x = next() * 2.0f - 1.0f;
    const float y = next() * 2.0f - 1.0f;
    const float z = next() * 2.0f - 1.0f;
    return Vec3(x, y, z);
  }
};

Material make_material(Vec3 color, bool metallic, float strength, bool dielectric, float ir) {
  Material m;
  m.color = color;
  m.metallic = metallic;
  m.metallic_strength = strength;
  m.dielectric = dielectric;
  m.ir = ir;
  return m;
}

void require(int rc, const char* what) {
  if (rc != RT_OK)
    throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}

rt_request make_request(const cpu::Dispatch& d, uint32_t seed, uint32_t flags,
                        std::vector<rt_sphere>& s, std::vector<rt_material>& m,
                        std::vector<rt_tile>& tiles) {
  d.world.to_abi(s, m);
  tiles.clear();
  for (const cpu::TileJob& j : d.job_queue) {
    rt_tile t = {(uint32_t)j.x, (uint32_t)j.y, (uint32_t)j.width, (uint32_t)j.height};
    tiles.push_back(t);
  }
  rt_request r;
  std::memset(&r, 0, sizeof r);
  r.image_width = (uint32_t)d.image_width;
  r.image_height = (uint32_t)d.image_height;
  r.max_depth = (uint32_t)d.max_depth;
  r.samples_per_pixel = (uint32_t)d.samples_per_pixel;
  r.seed = seed;
  r.flags = flags;
  r.camera = d.camera.m;
  r.spheres = s.data();
  r.n_spheres = (uint32_t)s.size();
  r.materials = m.data();
  r.n_materials = (uint32_t)m.size();
  r.tiles = tiles.data();
  r.n_tiles = (uint32_t)tiles.size();
  return r;
}

}  // namespace

// Camera::new, src/main.rs:58-68.  f32::to_radians(x) = x * (PI / 180).
Camera Camera::create(Vec3 position, Vec3 look_at, float fov_degrees, float aspect_ratio) {
  const float pi = 3.14159265358979323846f;
  const Mat4 view = look_at_rh(position, look_at, Vec3(0.0f, 1.0f, 0.0f));
  const Mat4 proj = perspective_rh(fov_degrees * (pi / 180.0f), aspect_ratio, 0.1f, 100.0f);
  const Mat4 vi = inverse(view), pi_ = inverse(proj);
  Camera c;
  std::memcpy(c.m.view_inv, vi.c, sizeof vi.c);
  std::memcpy(c.m.proj_inv, pi_.c, sizeof pi_.c);
  return c;
}

void World::to_abi(std::vector<rt_sphere>& s, std::vector<rt_material>& m) const {
  s.resize(spheres.size());
  m.resize(materials.size());
  for (std::size_t i = 0; i < spheres.size(); ++i) {
    s[i].x = spheres[i].position.x;
    s[i].y = spheres[i].position.y;
    s[i].z = spheres[i].position.z;
    s[i].radius = spheres[i].radius;
    s[i].material_id = (uint32_t)spheres[i].material_id;
  }
  for (std::size_t i = 0; i < materials.size(); ++i) {
    m[i].r = materials[i].color.x;
    m[i].g = materials[i].color.y;
    m[i].b = materials[i].color.z;
    m[i].flags = (materials[i].metallic ? RT_MAT_METALLIC : 0u) |
                 (materials[i].dielectric ? RT_MAT_DIELECTRIC : 0u);
    m[i].metallic_strength = materials[i].metallic_strength;
    m[i].ir = materials[i].ir;
  }
}

// create_random_world, src/main.rs:326-466.  Draw order per cell: x, z, choose_mat, then the
// colour (3 draws) and, for metal, the strength.  Small spheres first, then the ground, then
// the three unit spheres.  The reference has no overlap test against the big spheres.
World create_random_world(const RandomWorldParams& p) {
  World w;
  SceneRng rng(p.seed);
  for (int ai = p.grid_lo; ai < p.grid_hi; ++ai) {
    for (int bi = p.grid_lo; bi < p.grid_hi; ++bi) {
      const float a = (float)ai, b = (float)bi;
      const float x = a + 0.9f * rng.next();
      const float y = 0.2f;
      const float z = b + 0.9f * rng.next();
      const float choose_mat = rng.next();
      std::size_t material;
      if (choose_mat < p.diffuse_below) {
        const Vec3 c = rng.vec3();
        material = w.submit_material(
            make_material(Vec3(c.x * 0.9f, c.y * 0.9f, c.z * 0.9f), false, 0.0f, false, 0.0f));
      } else if (choose_mat < p.metal_below) {
        const Vec3 c = rng.vec3();
        const float strength = rng.next();
        material = w.submit_material(make_material(c, true, strength, false, 0.0f));
      } else {
        material = w.submit_material(make_material(Vec3(1.0f, 1.0f, 1.0f), false, 0.0f, true, 1.5f));
      }
      Sphere s;
      s.position = Vec3(x, y, z);
      s.radius = 0.2f;
      s.material_id = material;
      w.spheres.push_back(s);
    }
  }
  auto push = [&w](Vec3 pos, float radius, const Material& m) {
    Sphere s;
    s.position = pos;
    s.radius = radius;
    s.material_id = w.submit_material(m);
    w.spheres.push_back(s);
  };
  const bool glass = p.big_spheres_all_dielectric;
  // ground, src/main.rs:391-405
  push(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, make_material(Vec3(0.5f, 0.5f, 0.5f), false, 0.0f, false, 0.0f));
  // "Sphere 1", src/main.rs:409-423: dielectric, ir 1.4
  push(Vec3(0.0f, 1.0f, 0.0f), 1.0f,
       make_material(Vec3(0.1f, 0.1f, 0.6f), false, 0.0f, true, glass ? 1.5f : 1.4f));
  // "Sphere 2", src/main.rs:427-441: diffuse
  push(Vec3(-4.0f, 1.0f, 0.0f), 1.0f,
       glass ? make_material(Vec3(0.1f, 0.1f, 0.6f), false, 0.0f, true, 1.5f)
             : make_material(Vec3(0.1f, 0.1f, 0.6f), false, 0.0f, false, 0.0f));
  // "Sphere 3", src/main.rs:445-460: metal, strength 1.0
  push(Vec3(4.0f, 1.0f, 0.0f), 1.0f,
       glass ? make_material(Vec3(0.7f, 0.6f, 0.5f), false, 0.0f, true, 1.5f)
             : make_material(Vec3(0.7f, 0.6f, 0.5f), true, 1.0f, false, 0.0f));
  return w;
}

// create_simple_world, src/main.rs:468-519
World create_simple_world() {
  World w;
  Sphere ground;
  ground.position = Vec3(0.0f, -1000.0f, 0.0f);
  ground.radius = 1000.0f;
  ground.material_id = w.submit_material(make_material(Vec3(0.5f, 0.5f, 0.5f), false, 0.0f, false, 0.0f));
  w.spheres.push_back(ground);
  Sphere ball;
  ball.position = Vec3(4.0f, 1.0f, 0.0f);
  ball.radius = 1.0f;
  ball.material_id = w.submit_material(make_material(Vec3(0.7f, 0.6f, 0.5f), true, 0.1f, false, 0.0f));
  w.spheres.push_back(ball);
  return w;
}

namespace cpu {

// create_job_queue, src/cpu/job.rs:76-152
std::deque<TileJob> create_job_queue(std::size_t image_width, std::size_t image_height,
                                     std::size_t tile_width, std::size_t tile_height) {
  if (tile_width == 0 || tile_height == 0) throw std::runtime_error("attempt to divide by zero");
  std::deque<TileJob> queue;
  const std::size_t num_tiles_x = image_width / tile_width;
  const std::size_t num_tiles_y = image_height / tile_height;
  const std::size_t rest_w = image_width % tile_width;
  const std::size_t rest_h = image_height % tile_height;
  for (std::size_t ty = 0; ty < num_tiles_y; ++ty)
    for (std::size_t tx = 0; tx < num_tiles_x; ++tx)
      queue.push_back(TileJob{tx * tile_width, ty * tile_height, tile_width, tile_height});
  if (rest_h > 0)   // bottom strip
    for (std::size_t i = 0; i < num_tiles_x; ++i)
      queue.push_back(TileJob{i * tile_width, num_tiles_y * tile_height, tile_width, rest_h});
  if (rest_w > 0)   // right strip
    for (std::size_t i = 0; i < num_tiles_y; ++i)
      queue.push_back(TileJob{num_tiles_x * tile_width, i * tile_height, rest_w, tile_height});
  if (rest_w != 0 && rest_h != 0)   // corner; start_y uses tile_width as in src/cpu/job.rs:140
    queue.push_back(TileJob{num_tiles_x * tile_width, num_tiles_y * tile_width, rest_w, rest_h});
  return queue;
}

// debug_write_job_queue_to_framebuffer, src/cpu/job.rs:154-176
std::vector<Vec3> debug_write_job_queue_to_framebuffer(std::size_t image_width,
                                                       std::size_t image_height,
                                                       const std::deque<TileJob>& job_queue) {
  std::vector<Vec3> result(image_width * image_height);
  for (const TileJob& job : job_queue)
    for (std::size_t yoff = 0; yoff < job.height; ++yoff)
      for (std::size_t xoff = 0; xoff < job.width; ++xoff) {
        const float r = (float)xoff / (float)job.width;
        const float g = (float)yoff / (float)job.height;
        result.at((job.x + xoff) + (job.y + yoff) * image_width) = Vec3(r, g, 0.0f);
      }
  return result;
}

Dispatch Dispatch::create(std::deque<TileJob> job_queue, std::size_t image_width,
                          std::size_t image_height, std::size_t max_depth,
                          std::size_t samples_per_pixel, Camera camera, World world) {
  Dispatch d;
  d.job_queue = std::move(job_queue);
  d.image_width = image_width;
  d.image_height = image_height;
  d.max_depth = max_depth;
  d.samples_per_pixel = samples_per_pixel;
  d.camera = camera;
  d.world = std::move(world);
  return d;
}

static inline float clamp_like_rust(float v, float lo, float hi) {   // NaN stays NaN
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}

// compose_final_framebuffer, src/cpu/mod.rs:51-83
std::vector<Vec3> compose_final_framebuffer(std::size_t image_width, std::size_t image_height,
                                            std::size_t samples_per_pixel,
                                            const std::vector<JobResult>& job_results) {
  std::vector<Vec3> framebuffer(image_width * image_height);
  const float scale = 1.0f / (float)samples_per_pixel;
  for (const JobResult& result : job_results)
    for (std::size_t yoff = 0; yoff < result.height; ++yoff)
      for (std::size_t xoff = 0; xoff < result.width; ++xoff) {
        const Vec3 color = result.pixel_colors.at(xoff + yoff * result.width);
        const float r = clamp_like_rust(std::sqrt(color.x * scale), 0.0f, 0.999f);
        const float g = clamp_like_rust(std::sqrt(color.y * scale), 0.0f, 0.999f);
        const float b = clamp_like_rust(std::sqrt(color.z * scale), 0.0f, 0.999f);
        framebuffer.at((result.x + xoff) + (result.y + yoff) * image_width) = Vec3(r, g, b);
      }
  return framebuffer;
}

std::vector<Vec3> dispatch_work_gpu(const Dispatch& dispatch, uint32_t seed, uint32_t flags,
                                    int num_gpus) {
  std::vector<rt_sphere> s;
  std::vector<rt_material> m;
  std::vector<rt_tile> tiles;
  const rt_request req = make_request(dispatch, seed, flags, s, m, tiles);
  static_assert(sizeof(Vec3) == 3 * sizeof(float), "Vec3 must be three packed floats");
  std::vector<Vec3> framebuffer(dispatch.image_width * dispatch.image_height);
  if (num_gpus > 1) {
    std::vector<int> devices(num_gpus);
    for (int k = 0; k < num_gpus; ++k) devices[k] = k;
    require(rt_render_multi(&req, devices.data(), num_gpus,
                            reinterpret_cast<float*>(framebuffer.data())),
            "rt_render_multi");
  } else {
    require(rt_render(&req, reinterpret_cast<float*>(framebuffer.data())), "rt_render");
  }
  return framebuffer;
}

std::vector<Vec3> dispatch_work_gpu_progressive(const Dispatch& dispatch,
                                                const std::vector<std::size_t>& passes,
                                                uint32_t seed, uint32_t flags,
                                                bool (*on_pass)(std::size_t)) {
  std::vector<rt_sphere> s;
  std::vector<rt_material> m;
  std::vector<rt_tile> tiles;
  rt_request req = make_request(dispatch, seed, flags, s, m, tiles);
  std::size_t total = 0;
  for (const rt_tile& t : tiles) total += (std::size_t)t.width * t.height;
  std::vector<float> sums(total * 3, 0.0f);
  std::size_t done = 0;
  for (std::size_t n : passes) {
    if (n == 0) continue;
    req.samples_per_pixel = (uint32_t)n;
    require(rt_accumulate_tiles(&req, (uint32_t)done, sums.data()), "rt_accumulate_tiles");
    done += n;
    if (on_pass && !on_pass(done)) break;
  }
  if (done == 0) throw std::runtime_error("dispatch_work_gpu_progressive: no samples rendered");
  std::vector<Vec3> framebuffer(dispatch.image_width * dispatch.image_height);
  require(rt_compose(tiles.data(), (uint32_t)tiles.size(), sums.data(), (uint32_t)dispatch.image_width,
                     (uint32_t)dispatch.image_height, (uint32_t)done,
                     reinterpret_cast<float*>(framebuffer.data())),
          "rt_compose");
  return framebuffer;
}

std::vector<JobResult> execute_jobs_gpu(const Dispatch& dispatch, uint32_t seed, uint32_t flags) {
  std::vector<rt_sphere> s;
  std::vector<rt_material> m;
  std::vector<rt_tile> tiles;
  const rt_request req = make_request(dispatch, seed, flags, s, m, tiles);
  std::size_t total = 0;
  for (const rt_tile& t : tiles) total += (std::size_t)t.width * t.height;
  std::vector<float> sums(total * 3);
  require(rt_render_tiles(&req, sums.data()), "rt_render_tiles");
  std::vector<JobResult> results;
  std::size_t base = 0;
  for (const TileJob& j : dispatch.job_queue) {
    JobResult r;
    r.x = j.x; r.y = j.y; r.width = j.width; r.height = j.height;
    r.pixel_colors.resize(j.width * j.height);
    std::memcpy(static_cast<void*>(r.pixel_colors.data()), sums.data() + 3 * base, r.pixel_colors.size() * sizeof(Vec3));
    base += j.width * j.height;
    results.push_back(std::move(r));
  }
  return results;
}

}  // namespace cpu

// `(pixel.x * 255.0) as u8`, src/main.rs:534-536: saturating, NaN -> 0
static inline uint8_t as_u8(float v) {
  if (!(v > 0.0f)) return 0;
  if (v >= 255.0f) return 255;
  return (uint8_t)v;
}

// write_framebuffer_to_image, src/main.rs:521-546.  The bmp crate writes an uncompressed
// 24-bit BMP (rows stored bottom-up, padded to 4 bytes, pixels B,G,R).  The reference puts
// framebuffer row y at picture row H-1-y, so framebuffer row 0 is the first row stored.
void write_framebuffer_to_image(const std::string& path, std::size_t image_width,
                                std::size_t image_height, const std::vector<Vec3>& framebuffer) {
  if (framebuffer.size() < image_width * image_height)
    throw std::runtime_error("index out of bounds: framebuffer is smaller than the image");
  const uint32_t row_bytes = (uint32_t)((image_width * 3 + 3) & ~(std::size_t)3);
  const uint32_t data_size = row_bytes * (uint32_t)image_height;
  uint8_t header[54] = {0};
  auto put32 = [&header](int at, uint32_t v) {
    for (int i = 0; i < 4; ++i) header[at + i] = (uint8_t)(v >> (8 * i));
  };
  header[0] = 'B'; header[1] = 'M';
  put32(2, 54 + data_size);
  put32(10, 54);
  put32(14, 40);
  put32(18, (uint32_t)image_width);
  put32(22, (uint32_t)image_height);
  header[26] = 1;
  header[28] = 24;
  put32(34, data_size);
  put32(38, 1000);
  put32(42, 1000);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("Failed to save image: cannot open " + path);
  std::vector<uint8_t> row(row_bytes, 0);
  bool ok = std::fwrite(header, 1, sizeof header, f) == sizeof header;
  for (std::size_t y = 0; ok && y < image_height; ++y) {
    for (std::size_t x = 0; x < image_width; ++x) {
      const Vec3 pixel = framebuffer[x + y * image_width];
      row[3 * x + 2] = as_u8(pixel.x * 255.0f);
      row[3 * x + 1] = as_u8(pixel.y * 255.0f);
      row[3 * x + 0] = as_u8(pixel.z * 255.0f);
    }
    ok = std::fwrite(row.data(), 1, row_bytes, f) == row_bytes;
  }
  if (std::fclose(f) != 0) ok = false;
  if (!ok) throw std::runtime_error("Failed to save image: write error on " + path);
}

}  // namespace rt_host

// ---- C surface of the mirror, for the Python binding and the tests ---------------------------
// Errors (C++ exceptions above) come back as -1 with the message in rth_last_error().
namespace {
thread_local std::string g_host_error;
template <class F>
int guarded(F f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return -1;
  }
}
rt_host::World world_from_abi(const rt_sphere* s, uint32_t ns, const rt_material* m, uint32_t nm) {
  rt_host::World w;
  for (uint32_t i = 0; i < nm; ++i) {
    rt_host::Material mm;
    mm.color = rt_host::Vec3(m[i].r, m[i].g, m[i].b);
    mm.metallic = (m[i].flags & RT_MAT_METALLIC) != 0;
    mm.dielectric = (m[i].flags & RT_MAT_DIELECTRIC) != 0;
    mm.metallic_strength = m[i].metallic_strength;
    mm.ir = m[i].ir;
    w.materials.push_back(mm);
  }
  for (uint32_t i = 0; i < ns; ++i) {
    rt_host::Sphere ss;
    ss.position = rt_host::Vec3(s[i].x, s[i].y, s[i].z);
    ss.radius = s[i].radius;
    ss.material_id = s[i].material_id;
    w.spheres.push_back(ss);
  }
  return w;
}
std::deque<rt_host::cpu::TileJob> queue_from_abi(const rt_tile* t, uint32_t n) {
  std::deque<rt_host::cpu::TileJob> q;
  for (uint32_t i = 0; i < n; ++i) q.push_back(rt_host::cpu::TileJob{t[i].x, t[i].y, t[i].width, t[i].height});
  return q;
}
}  // namespace

extern "C" {

const char* rth_last_error(void) { return g_host_error.c_str(); }

int rth_camera_new(const float position[3], const float look_at[3], float fov_degrees,
                   float aspect_ratio, rt_camera* out) {
  return guarded([&] {
    *out = rt_host::Camera::create(rt_host::Vec3(position[0], position[1], position[2]),
                                   rt_host::Vec3(look_at[0], look_at[1], look_at[2]), fov_degrees,
                                   aspect_ratio).m;
  });
}

// Writes (grid_hi-grid_lo)^2 + 4 spheres and materials; returns the count or -1.
int rth_create_random_world(uint32_t seed, int grid_lo, int grid_hi, float diffuse_below,
                            float metal_below, int big_all_dielectric, rt_sphere* spheres,
                            rt_material* materials, uint32_t capacity) {
  int n = -1;
  int rc = guarded([&] {
    rt_host::RandomWorldParams p;
    p.seed = seed;
    p.grid_lo = grid_lo;
    p.grid_hi = grid_hi;
    p.diffuse_below = diffuse_below;
    p.metal_below = metal_below;
    p.big_spheres_all_dielectric = big_all_dielectric != 0;
    const rt_host::World w = rt_host::create_random_world(p);
    if (w.spheres.size() > capacity) throw std::runtime_error("capacity too small");
    std::vector<rt_sphere> s;
    std::vector<rt_material> m;
    w.to_abi(s, m);
    std::memcpy(spheres, s.data(), s.size() * sizeof(rt_sphere));
    std::memcpy(materials, m.data(), m.size() * sizeof(rt_material));
    n = (int)s.size();
  });
  return rc ? rc : n;
}

int rth_create_simple_world(rt_sphere* spheres, rt_material* materials) {
  return guarded([&] {
    std::vector<rt_sphere> s;
    std::vector<rt_material> m;
    rt_host::create_simple_world().to_abi(s, m);
    std::memcpy(spheres, s.data(), s.size() * sizeof(rt_sphere));
    std::memcpy(materials, m.data(), m.size() * sizeof(rt_material));
  });
}

// Returns the number of jobs (writes at most `capacity`), or -1.
int rth_create_job_queue(uint32_t image_width, uint32_t image_height, uint32_t tile_width,
                         uint32_t tile_height, rt_tile* out, uint32_t capacity) {
  int n = -1;
  int rc = guarded([&] {
    const auto q = rt_host::cpu::create_job_queue(image_width, image_height, tile_width, tile_height);
    uint32_t i = 0;
    for (const auto& j : q) {
      if (i < capacity) out[i] = rt_tile{(uint32_t)j.x, (uint32_t)j.y, (uint32_t)j.width, (uint32_t)j.height};
      ++i;
    }
    n = (int)q.size();
  });
  return rc ? rc : n;
}

int rth_debug_write_job_queue_to_framebuffer(uint32_t image_width, uint32_t image_height,
                                             const rt_tile* jobs, uint32_t n_jobs, float* out_rgb) {
  return guarded([&] {
    const auto fb = rt_host::cpu::debug_write_job_queue_to_framebuffer(image_width, image_height,
                                                                       queue_from_abi(jobs, n_jobs));
    std::memcpy(out_rgb, fb.data(), fb.size() * sizeof(rt_host::Vec3));
  });
}

int rth_compose_final_framebuffer(uint32_t image_width, uint32_t image_height,
                                  uint32_t samples_per_pixel, const rt_tile* jobs, uint32_t n_jobs,
                                  const float* tile_sums, float* out_rgb) {
  return guarded([&] {
    std::vector<rt_host::cpu::JobResult> results;
    std::size_t base = 0;
    for (uint32_t i = 0; i < n_jobs; ++i) {
      rt_host::cpu::JobResult r;
      r.x = jobs[i].x; r.y = jobs[i].y; r.width = jobs[i].width; r.height = jobs[i].height;
      r.pixel_colors.resize(r.width * r.height);
      std::memcpy(static_cast<void*>(r.pixel_colors.data()), tile_sums + 3 * base, r.pixel_colors.size() * sizeof(rt_host::Vec3));
      base += r.width * r.height;
      results.push_back(std::move(r));
    }
    const auto fb = rt_host::cpu::compose_final_framebuffer(image_width, image_height,
                                                            samples_per_pixel, results);
    std::memcpy(out_rgb, fb.data(), fb.size() * sizeof(rt_host::Vec3));
  });
}

int rth_write_framebuffer_to_image(const char* path, uint32_t image_width, uint32_t image_height,
                                   const float* rgb) {
  return guarded([&] {
    std::vector<rt_host::Vec3> fb((std::size_t)image_width * image_height);
    std::memcpy(static_cast<void*>(fb.data()), rgb, fb.size() * sizeof(rt_host::Vec3));
    rt_host::write_framebuffer_to_image(path, image_width, image_height, fb);
  });
}

// cpu::dispatch_work_gpu over POD inputs: Dispatch::new(...) then dispatch_work_gpu(dispatch).
int rth_dispatch_work_gpu(const rt_tile* jobs, uint32_t n_jobs, uint32_t image_width,
                          uint32_t image_height, uint32_t max_depth, uint32_t samples_per_pixel,
                          const rt_camera* camera, const rt_sphere* spheres, uint32_t n_spheres,
                          const rt_material* materials, uint32_t n_materials, uint32_t seed,
                          uint32_t flags, float* out_rgb) {
  return guarded([&] {
    rt_host::Camera cam;
    cam.m = *camera;
    const auto d = rt_host::cpu::Dispatch::create(
        queue_from_abi(jobs, n_jobs), image_width, image_height, max_depth, samples_per_pixel, cam,
        world_from_abi(spheres, n_spheres, materials, n_materials));
    const auto fb = rt_host::cpu::dispatch_work_gpu(d, seed, flags);
    std::memcpy(out_rgb, fb.data(), fb.size() * sizeof(rt_host::Vec3));
  });
}

// cpu::dispatch_work_gpu_progressive over POD inputs; passes[k] samples per pixel in pass k.
int rth_dispatch_work_gpu_progressive(const rt_tile* jobs, uint32_t n_jobs, uint32_t image_width,
                                      uint32_t image_height, uint32_t max_depth,
                                      const uint32_t* passes, uint32_t n_passes,
                                      const rt_camera* camera, const rt_sphere* spheres,
                                      uint32_t n_spheres, const rt_material* materials,
                                      uint32_t n_materials, uint32_t seed, uint32_t flags,
                                      float* out_rgb) {
  return guarded([&] {
    rt_host::Camera cam;
    cam.m = *camera;
    const auto d = rt_host::cpu::Dispatch::create(
        queue_from_abi(jobs, n_jobs), image_width, image_height, max_depth, 1, cam,
        world_from_abi(spheres, n_spheres, materials, n_materials));
    std::vector<std::size_t> p(passes, passes + n_passes);
    const auto fb = rt_host::cpu::dispatch_work_gpu_progressive(d, p, seed, flags);
    std::memcpy(out_rgb, fb.data(), fb.size() * sizeof(rt_host::Vec3));
  });
}

}  // extern "C"
