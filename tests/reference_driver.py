"""The CPU half of the reference's main() (src/main.rs:889-937) from the CPU oracle.  TEST SIDE.

Writes, into `outdir`, the three files the reference's driver writes when its `return;`
(src/main.rs:885) is removed:

    debug_tiles.bmp       debug_write_job_queue_to_framebuffer      src/main.rs:896-903
    singlethreaded.bmp    dispatch_work_cpu_singlethreaded          src/main.rs:905-919
    multithreaded.bmp     dispatch_work_cpu_multithreaded           src/main.rs:921-937

with the same camera, create_random_world(), 64x64 job queue and print-outs.  It exists so that
ray_tracing_amd/rt_main's debug_tiles.bmp and gpu.bmp can be compared with what the CPU renderer
produces; it is not part of the product (which has no CPU renderer).

    python tests/reference_driver.py <width> <spp> <max_depth> <grid> <seed> <outdir> [skip_single]
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from ray_tracing_amd import abi  # noqa: E402


def write_bmp(path, w, h, fb):
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    assert O.lib().ro_write_framebuffer_to_image(os.fsencode(path), w, h, abi.ptr(fb)) == 0


def main(argv):
    width = int(argv[1]) if len(argv) > 1 else 1920
    spp = int(argv[2]) if len(argv) > 2 else 100          # SAMPLES_PER_PIXEL, src/main.rs:6
    depth = int(argv[3]) if len(argv) > 3 else 4          # MAX_DEPTH, src/main.rs:7
    grid = int(argv[4]) if len(argv) > 4 else 4
    seed = int(argv[5]) if len(argv) > 5 else 1
    outdir = argv[6] if len(argv) > 6 else "."
    skip_single = len(argv) > 7 and argv[7] == "skip_single"
    aspect = np.float32(16.0) / np.float32(9.0)
    height = int(np.float32(width) / aspect)              # `(image_width as f32 / aspect_ratio) as usize`
    print("Width: %d Height: %d" % (width, height))
    cores = os.cpu_count() or 1
    print("Num Cores: %d" % cores)                        # src/main.rs:851
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, float(aspect))
    s, m = O.random_world(seed, -grid, grid)
    tiles = O.job_queue(width, height, 64, 64)
    dbg = np.zeros((height, width, 3), dtype=np.float32)
    O.lib().ro_debug_write_job_queue_to_framebuffer(width, height, abi.ptr(tiles), len(tiles),
                                                    abi.ptr(dbg))
    write_bmp(os.path.join(outdir, "debug_tiles.bmp"), width, height, dbg)
    if not skip_single:
        print("-- Singlethreaded --")
        fb, secs, _ = O.render(s, m, cam, width, height, spp, depth, seed, mode=O.RO_MODE_RECURSIVE,
                               threads=1, tiles=tiles)
        print("Time: %.2f s (%d ms)" % (secs, int(secs * 1000)))      # src/cpu/single.rs:43-44
        write_bmp(os.path.join(outdir, "singlethreaded.bmp"), width, height, fb)
    print("-- Multithreaded --")
    fb, secs, _ = O.render(s, m, cam, width, height, spp, depth, seed, mode=O.RO_MODE_RECURSIVE,
                           threads=cores, tiles=tiles)
    print("Time: %.2f s (%d ms)" % (secs, int(secs * 1000)))          # src/cpu/multi.rs:164-165
    write_bmp(os.path.join(outdir, "multithreaded.bmp"), width, height, fb)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
