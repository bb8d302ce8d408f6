// rt_host.hpp - C++ mirror of the reference's host-side types and of its `cpu` module's public
// surface, above the C ABI of librt_hip.so.
//
// The reference is a Rust program; no Rust toolchain exists where this is built, so the host
// side a Rust caller would keep is restated here in C++ with the reference's names and
// argument meaning:
//   Camera::new                      src/main.rs:58-68
//   World, Material, Sphere          src/main.rs:108-125, 168-172
//   create_random_world / _simple_   src/main.rs:326-519
//   cpu::create_job_queue            src/cpu/job.rs:76-152
//   cpu::debug_write_job_queue_...   src/cpu/job.rs:154-176
//   cpu::Dispatch, Dispatch::new     src/cpu/mod.rs:19-49
//   compose_final_framebuffer        src/cpu/mod.rs:51-83
//   write_framebuffer_to_image       src/main.rs:521-546
// and one sibling of dispatch_work_cpu_multithreaded (src/cpu/multi.rs:131-172):
//   cpu::dispatch_work_gpu(dispatch, seed) -> Vec<Vec3>, the stub the reference left at
//   src/main.rs:940.  It renders through rt_render(); there is no CPU renderer here.
#pragma once

#include <cstdint>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt_host {

struct Vec3 {
  float x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

// struct Camera, src/main.rs:51-55: the two inverse matrices, column-major
struct Camera {
  rt_camera m;
  static Camera create(Vec3 position, Vec3 look_at, float fov_degrees, float aspect_ratio);
};

// struct Material / Sphere / World, src/main.rs:108-125, 168-172
struct Material {
  Vec3 color;
  bool metallic = false;
  float metallic_strength = 0;
  bool dielectric = false;
  float ir = 0;
};

struct Sphere {
  Vec3 position;
  float radius = 0;
  std::size_t material_id = 0;
};

struct World {
  std::vector<Material> materials;
  std::vector<Sphere> spheres;
  std::size_t submit_material(const Material& m) {   // the closure at src/main.rs:330-335
    materials.push_back(m);
    return materials.size() - 1;
  }
  void to_abi(std::vector<rt_sphere>& s, std::vector<rt_material>& m) const;
};

// Parameters the reference hard-codes in create_random_world (src/main.rs:337-338, 349, 359,
// 409-460).  The defaults reproduce it; the benchmark scenes change the grid and thresholds.
struct RandomWorldParams {
  uint32_t seed = 1;
  int grid_lo = -4, grid_hi = 4;
  float diffuse_below = 0.8f;
  float metal_below = 0.95f;
  bool big_spheres_all_dielectric = false;
};

World create_random_world(const RandomWorldParams& p = RandomWorldParams());
World create_simple_world();

namespace cpu {

// struct TileJob, src/cpu/job.rs:6-13
struct TileJob {
  std::size_t x = 0, y = 0, width = 0, height = 0;
};

std::deque<TileJob> create_job_queue(std::size_t image_width, std::size_t image_height,
                                     std::size_t tile_width, std::size_t tile_height);

std::vector<Vec3> debug_write_job_queue_to_framebuffer(std::size_t image_width,
                                                       std::size_t image_height,
                                                       const std::deque<TileJob>& job_queue);

// struct JobResult, src/cpu/job.rs:15-23
struct JobResult {
  std::size_t x = 0, y = 0, width = 0, height = 0;
  std::vector<Vec3> pixel_colors;
};

// struct Dispatch, src/cpu/mod.rs:19-30
struct Dispatch {
  std::deque<TileJob> job_queue;
  std::size_t image_width = 0, image_height = 0;
  std::size_t max_depth = 0, samples_per_pixel = 0;
  Camera camera;
  World world;
  static Dispatch create(std::deque<TileJob> job_queue, std::size_t image_width,
                         std::size_t image_height, std::size_t max_depth,
                         std::size_t samples_per_pixel, Camera camera, World world);
};

std::vector<Vec3> compose_final_framebuffer(std::size_t image_width, std::size_t image_height,
                                            std::size_t samples_per_pixel,
                                            const std::vector<JobResult>& job_results);

// Whole frame on the GPU; same return contract as dispatch_work_cpu_multithreaded.  Throws
// std::runtime_error where the reference would panic.
// num_gpus plays the part of dispatch_work_cpu_multithreaded's num_threads (src/cpu/multi.rs:132):
// with more than one, this one process drives devices 0..num_gpus-1 through rt_render_multi.
std::vector<Vec3> dispatch_work_gpu(const Dispatch& dispatch, uint32_t seed, uint32_t flags = 0,
                                    int num_gpus = 1);
// The tile loop on the GPU: one JobResult (un-normalised sums) per job of the queue.
std::vector<JobResult> execute_jobs_gpu(const Dispatch& dispatch, uint32_t seed, uint32_t flags = 0);
// The same contract rendered in passes of `passes[k]` samples per pixel that add into the
// per-tile sums (`pixel_color += color`, src/cpu/job.rs:58), composed once at the end with the
// number of samples actually rendered (src/cpu/mod.rs:67); dispatch.samples_per_pixel is not
// used.  `on_pass(samples_done)` may return false to stop early.
std::vector<Vec3> dispatch_work_gpu_progressive(const Dispatch& dispatch,
                                                const std::vector<std::size_t>& passes,
                                                uint32_t seed, uint32_t flags = 0,
                                                bool (*on_pass)(std::size_t) = nullptr);

}  // namespace cpu

void write_framebuffer_to_image(const std::string& path, std::size_t image_width,
                                std::size_t image_height, const std::vector<Vec3>& framebuffer);

}  // namespace rt_host
