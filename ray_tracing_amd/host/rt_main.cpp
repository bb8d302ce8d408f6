// rt_main.cpp - the reference's main() (src/main.rs:841-941) with its CPU-renderer half kept
// and the renderer swapped for the GPU: same hard-coded camera, create_random_world(), 64x64
// job queue, debug_tiles.bmp, then `gpu.bmp` where the reference wrote singlethreaded.bmp /
// multithreaded.bmp (src/main.rs:905-937; this library has no CPU renderer, so those two files
// are not written here: tests/reference_driver.py writes them from the CPU oracle for
// comparison).  The wgpu experiment that precedes `return;` at src/main.rs:885 is not
// reproduced.
//
// usage: rt_main [width=1920] [spp=100] [max_depth=4] [grid=4] [seed=1] [outdir=.] [flags=0] [gpus=1]
//        (flags: RT_FLAG_* of include/rt_abi.h, e.g. 4 = the opt-in hierarchy)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rt_host.hpp"

int main(int argc, char** argv) {
  using namespace rt_host;
  const std::size_t image_width = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1920;
  const std::size_t samples_per_pixel = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 100;  // SAMPLES_PER_PIXEL, src/main.rs:6
  const std::size_t max_depth = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 4;             // MAX_DEPTH, src/main.rs:7
  const int grid = argc > 4 ? std::atoi(argv[4]) : 4;
  const uint32_t seed = argc > 5 ? (uint32_t)std::strtoul(argv[5], nullptr, 10) : 1;
  const std::string outdir = argc > 6 ? argv[6] : ".";
  const uint32_t flags = argc > 7 ? (uint32_t)std::strtoul(argv[7], nullptr, 10) : 0;
  const int num_gpus = argc > 8 ? std::atoi(argv[8]) : 1;   // num_cpus::get() in the reference

  const float aspect_ratio = 16.0f / 9.0f;
  const std::size_t image_height = (std::size_t)((float)image_width / aspect_ratio);
  std::printf("Width: %zu Height: %zu\n", image_width, image_height);

  try {
    const Camera camera = Camera::create(Vec3(13.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), 20.0f, aspect_ratio);
    RandomWorldParams wp;
    wp.seed = seed;
    wp.grid_lo = -grid;
    wp.grid_hi = grid;
    const World world = create_random_world(wp);

    const auto job_queue = cpu::create_job_queue(image_width, image_height, 64, 64);
    write_framebuffer_to_image(outdir + "/debug_tiles.bmp", image_width, image_height,
                               cpu::debug_write_job_queue_to_framebuffer(image_width, image_height, job_queue));

    if (rt_init(0) != RT_OK) {
      std::fprintf(stderr, "rt_init: %s\n", rt_last_error());
      return 1;
    }
    std::printf("-- GPU (%zu spheres) --\n", world.spheres.size());
    const auto dispatch = cpu::Dispatch::create(job_queue, image_width, image_height, max_depth,
                                                samples_per_pixel, camera, world);
    const auto t0 = std::chrono::steady_clock::now();
    const auto framebuffer = cpu::dispatch_work_gpu(dispatch, seed, flags, num_gpus);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // The reference's "Finished Jobs: xx.xx" line (src/cpu/multi.rs:111-114) polls the result
    // vector every 500 ms; the GPU delivers the whole frame in one launch, so there is no
    // intermediate state to report and no such line is printed.
    std::printf("Time: %.2f s (%lld ms)\n", secs, (long long)(secs * 1000.0));   // src/cpu/multi.rs:164-165
    std::printf("%.1f Msamples/s\n", (double)image_width * image_height * samples_per_pixel / secs * 1e-6);
    write_framebuffer_to_image(outdir + "/gpu.bmp", image_width, image_height, framebuffer);
    rt_shutdown();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
