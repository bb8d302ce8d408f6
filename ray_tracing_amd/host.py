"""Python face of the C++ host mirror (librt_host.so, sources in ray_tracing_amd/host/).

Names and argument meaning follow the reference program so that code written against it reads
the same here:

    reference (Rust)                                   here
    Camera::new(position, look_at, fov, aspect)        Camera.new(...)              src/main.rs:58
    create_random_world() / create_simple_world()      same names                   src/main.rs:326,468
    cpu::create_job_queue(w, h, tw, th)                create_job_queue(...)        src/cpu/job.rs:76
    cpu::debug_write_job_queue_to_framebuffer(...)     same name                    src/cpu/job.rs:154
    cpu::Dispatch::new(queue, w, h, depth, spp, camera, world)   Dispatch.new(...)  src/cpu/mod.rs:33
    cpu::dispatch_work_cpu_multithreaded(dispatch, n)  dispatch_work_gpu(dispatch, num_gpus)
                                                        (src/cpu/multi.rs:131; GPU sibling)
    write_framebuffer_to_image(path, w, h, fb)         same name                    src/main.rs:521

All arithmetic lives in the C++ library; this module only marshals numpy arrays.
"""
import ctypes as C
import os

import numpy as np

from . import abi, hip

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "librt_host.so")

HOST_EXPORTS = (
    "rth_last_error", "rth_camera_new", "rth_create_random_world", "rth_create_simple_world",
    "rth_create_job_queue", "rth_debug_write_job_queue_to_framebuffer",
    "rth_compose_final_framebuffer", "rth_write_framebuffer_to_image", "rth_dispatch_work_gpu",
    "rth_dispatch_work_gpu_progressive",
)

_lib = None


class HostError(RuntimeError):
    """The C++ mirror threw where the reference would have panicked."""


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C ray_tracing_amd/host`" % HOST_LIB_PATH)
    hip.lib()   # librt_host.so links librt_hip.so; load it (and the shared HIP runtime) first
    L = C.CDLL(HOST_LIB_PATH)
    f, u32, vp, i = C.c_float, C.c_uint32, C.c_void_p, C.c_int
    L.rth_last_error.restype = C.c_char_p
    L.rth_camera_new.argtypes = [C.POINTER(f), C.POINTER(f), f, f, C.POINTER(abi.RtCamera)]
    L.rth_create_random_world.argtypes = [u32, i, i, f, f, i, vp, vp, u32]
    L.rth_create_simple_world.argtypes = [vp, vp]
    L.rth_create_job_queue.argtypes = [u32, u32, u32, u32, vp, u32]
    L.rth_debug_write_job_queue_to_framebuffer.argtypes = [u32, u32, vp, u32, vp]
    L.rth_compose_final_framebuffer.argtypes = [u32, u32, u32, vp, u32, vp, vp]
    L.rth_write_framebuffer_to_image.argtypes = [C.c_char_p, u32, u32, vp]
    L.rth_dispatch_work_gpu.argtypes = [vp, u32, u32, u32, u32, u32, C.POINTER(abi.RtCamera),
                                        vp, u32, vp, u32, u32, u32, vp]
    L.rth_dispatch_work_gpu_progressive.argtypes = [vp, u32, u32, u32, u32, vp, u32,
                                                    C.POINTER(abi.RtCamera), vp, u32, vp, u32,
                                                    u32, u32, vp]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise HostError(lib().rth_last_error().decode("utf-8", "replace"))
    return rc


class Camera:
    """struct Camera (src/main.rs:51-55): inverse view and projection matrices."""

    def __init__(self, rt_camera):
        self.m = rt_camera

    @classmethod
    def new(cls, position, look_at, fov, aspect_ratio):
        cam = abi.RtCamera()
        pos = (C.c_float * 3)(*[float(v) for v in position])
        tgt = (C.c_float * 3)(*[float(v) for v in look_at])
        _check(lib().rth_camera_new(pos, tgt, float(fov), float(aspect_ratio), C.byref(cam)))
        return cls(cam)

    @property
    def view_matrix_inv(self):
        return np.array(self.m.view_inv[:], dtype=np.float32)

    @property
    def projection_matrix_inv(self):
        return np.array(self.m.proj_inv[:], dtype=np.float32)


class World:
    """struct World (src/main.rs:168-172) as two structured arrays."""

    def __init__(self, spheres, materials):
        self.spheres = np.ascontiguousarray(spheres, dtype=abi.SPHERE_DTYPE)
        self.materials = np.ascontiguousarray(materials, dtype=abi.MATERIAL_DTYPE)


def create_random_world(seed=1, grid=(-4, 4), diffuse_below=0.8, metal_below=0.95,
                        big_spheres_all_dielectric=False):
    """create_random_world (src/main.rs:326-466).  The defaults are the reference's literals;
    `grid=(-11, 11)` gives the 488-sphere book scene, `(-50, 50)` the 10 004-sphere one."""
    lo, hi = grid
    n = (hi - lo) ** 2 + 4
    s = np.zeros(n, dtype=abi.SPHERE_DTYPE)
    m = np.zeros(n, dtype=abi.MATERIAL_DTYPE)
    got = _check(lib().rth_create_random_world(int(seed) & 0xFFFFFFFF, lo, hi, diffuse_below,
                                               metal_below, int(big_spheres_all_dielectric),
                                               abi.ptr(s), abi.ptr(m), n))
    assert got == n
    return World(s, m)


def create_simple_world():
    s = np.zeros(2, dtype=abi.SPHERE_DTYPE)
    m = np.zeros(2, dtype=abi.MATERIAL_DTYPE)
    _check(lib().rth_create_simple_world(abi.ptr(s), abi.ptr(m)))
    return World(s, m)


def create_job_queue(image_width, image_height, tile_width, tile_height):
    """cpu::create_job_queue (src/cpu/job.rs:76-152) -> array of TileJob (abi.TILE_DTYPE)."""
    n = _check(lib().rth_create_job_queue(image_width, image_height, tile_width, tile_height, None, 0))
    tiles = np.zeros(n, dtype=abi.TILE_DTYPE)
    _check(lib().rth_create_job_queue(image_width, image_height, tile_width, tile_height,
                                      abi.ptr(tiles), n))
    return tiles


def debug_write_job_queue_to_framebuffer(image_width, image_height, job_queue):
    tiles = np.ascontiguousarray(job_queue, dtype=abi.TILE_DTYPE)
    out = np.zeros((image_height, image_width, 3), dtype=np.float32)
    _check(lib().rth_debug_write_job_queue_to_framebuffer(image_width, image_height, abi.ptr(tiles),
                                                          len(tiles), abi.ptr(out)))
    return out


class Dispatch:
    """struct Dispatch (src/cpu/mod.rs:19-30)."""

    def __init__(self, job_queue, image_width, image_height, max_depth, samples_per_pixel, camera,
                 world):
        self.job_queue = np.ascontiguousarray(job_queue, dtype=abi.TILE_DTYPE)
        self.image_width = int(image_width)
        self.image_height = int(image_height)
        self.max_depth = int(max_depth)
        self.samples_per_pixel = int(samples_per_pixel)
        self.camera = camera
        self.world = world

    new = classmethod(lambda cls, *a: cls(*a))


def compose_final_framebuffer(image_width, image_height, samples_per_pixel, job_queue, tile_sums):
    """compose_final_framebuffer (src/cpu/mod.rs:51-83) over concatenated per-tile sums."""
    tiles = np.ascontiguousarray(job_queue, dtype=abi.TILE_DTYPE)
    sums = np.ascontiguousarray(tile_sums, dtype=np.float32)
    out = np.zeros((image_height, image_width, 3), dtype=np.float32)
    _check(lib().rth_compose_final_framebuffer(image_width, image_height, samples_per_pixel,
                                               abi.ptr(tiles), len(tiles), abi.ptr(sums), abi.ptr(out)))
    return out


def write_framebuffer_to_image(path, image_width, image_height, framebuffer):
    fb = np.ascontiguousarray(framebuffer, dtype=np.float32)
    assert fb.size == image_width * image_height * 3
    _check(lib().rth_write_framebuffer_to_image(os.fsencode(path), image_width, image_height, abi.ptr(fb)))


def dispatch_work_gpu(dispatch, num_gpus=1, seed=1, flags=0):
    """GPU sibling of cpu::dispatch_work_cpu_multithreaded (src/cpu/multi.rs:131-172): returns
    the resolved framebuffer, shape (H, W, 3), fb[y, x] with y = 0 at the bottom.

    num_gpus == 1 renders through the C++ mirror (Dispatch -> rt_render).  num_gpus > 1 must be
    called by every rank of an initialised torch.distributed job; rows are sharded over the
    ranks and rank 0 receives the gathered frame (the others get None): see
    ray_tracing_amd.multi_gpu."""
    if num_gpus > 1:
        from . import multi_gpu
        return multi_gpu.dispatch_work_gpu_sharded(dispatch, seed=seed, flags=flags)
    d = dispatch
    out = np.zeros((d.image_height, d.image_width, 3), dtype=np.float32)
    _check(lib().rth_dispatch_work_gpu(abi.ptr(d.job_queue), len(d.job_queue), d.image_width,
                                       d.image_height, d.max_depth, d.samples_per_pixel,
                                       C.byref(d.camera.m), abi.ptr(d.world.spheres),
                                       len(d.world.spheres), abi.ptr(d.world.materials),
                                       len(d.world.materials), int(seed) & 0xFFFFFFFF, int(flags),
                                       abi.ptr(out)))
    return out


def dispatch_work_gpu_progressive_host(dispatch, passes, seed=1, flags=0):
    """cpu::dispatch_work_gpu_progressive of the C++ mirror: the passes go through the
    host-pointer entry point rt_accumulate_tiles and the frame is composed at the end."""
    d = dispatch
    p = np.ascontiguousarray(passes, dtype=np.uint32)
    out = np.zeros((d.image_height, d.image_width, 3), dtype=np.float32)
    _check(lib().rth_dispatch_work_gpu_progressive(
        abi.ptr(d.job_queue), len(d.job_queue), d.image_width, d.image_height, d.max_depth,
        abi.ptr(p), len(p), C.byref(d.camera.m), abi.ptr(d.world.spheres), len(d.world.spheres),
        abi.ptr(d.world.materials), len(d.world.materials), int(seed) & 0xFFFFFFFF, int(flags),
        abi.ptr(out)))
    return out


def dispatch_work_gpu_progressive(dispatch, passes, seed=1, flags=0, time_budget_s=None,
                                  on_pass=None):
    """The dispatch_work_* contract rendered in several passes that add into one buffer of
    un-normalised sums (the additive JobResult.pixel_colors of src/cpu/job.rs:48-61, divided by
    the sample count only at the end, src/cpu/mod.rs:67).

    `passes` lists the samples per pixel of each pass; `dispatch.samples_per_pixel` is ignored in
    favour of what was actually rendered.  With `time_budget_s` the passes stop early once that
    much wall time has gone by (at least one pass always runs).  `on_pass(done_spp, sums)` is
    called after each pass with the device tensor of sums (for previews or checkpoints).
    Returns (framebuffer (H, W, 3) resolved with the rendered sample count, rendered spp).
    With abi.RT_FLAG_SEQUENTIAL_SUM in `flags` the result is bit-identical to a one-pass render
    of the same number of samples."""
    import time

    import torch

    from . import hip

    d = dispatch
    W, H = d.image_width, d.image_height
    device = torch.device("cuda", torch.cuda.current_device())
    scene = hip.Scene(d.world.spheres, d.world.materials)
    sums = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    tile = np.array([(0, 0, W, H)], dtype=abi.TILE_DTYPE)
    off, stride = np.zeros(1, dtype=np.uint64), np.full(1, W, dtype=np.uint32)
    done, t0 = 0, time.perf_counter()
    for n in passes:
        n = int(n)
        if n <= 0:
            continue
        frame = hip.make_frame(d.camera.m, W, H, n, d.max_depth, seed, flags)
        hip.accumulate_device_async(scene, frame, done, tile, off, stride, sums.data_ptr(), stream)
        done += n
        if on_pass is not None or time_budget_s is not None:
            torch.cuda.synchronize()
        if on_pass is not None:
            on_pass(done, sums)
        if time_budget_s is not None and time.perf_counter() - t0 >= time_budget_s:
            break
    if done == 0:
        raise ValueError("no samples rendered: `passes` is empty")
    rgb = torch.empty_like(sums)
    hip.resolve_device_async(sums.data_ptr(), W * H, done, rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    scene.close()
    return rgb.cpu().numpy(), done
