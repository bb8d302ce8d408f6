"""ctypes binding of librt_hip.so, the C ABI declared in include/rt_abi.h.

The library is the product: there is no Python or CPU fallback behind these calls.  Importing
this module only loads the shared object (which works without a GPU so that the exported
symbols can be checked); every compute call needs `init()` to have found a gfx950 device and
raises RtError otherwise.
"""
import ctypes as C
import importlib.util
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB lets a developer point at a diagnostic build of the same library.
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")

# every symbol include/rt_abi.h declares
EXPORTS = (
    "rt_init", "rt_shutdown", "rt_last_error", "rt_abi_version",
    "rt_render", "rt_render_multi", "rt_render_tiles", "rt_accumulate_tiles", "rt_compose",
    "rt_scene_create", "rt_scene_destroy",
    "rt_render_device_async", "rt_render_rows_device_async", "rt_render_rows_device_noalloc",
    "rt_accumulate_device_async", "rt_resolve_device_async",
    "rt_synchronize", "rt_get_stats", "rt_get_launch_ms", "rt_get_phase_cycles", "rt_trace_rays",
    "rt_scene_scan_info", "rt_selftest_sqrt",
)


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt error %d: %s" % (code, message))
        self.code = code


_lib = None


def _share_torch_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch wheels ship their own libamdhip64.so.7;
    if this library bound to /opt/rocm's copy first, a later `import torch` would find no GPU.
    So when torch is installed, load the runtime it ships (without importing torch) before
    librt_hip.so: the dynamic linker then resolves our libamdhip64.so.7 to that same copy, and
    device pointers / streams from torch are valid in our calls.  Without torch the system
    runtime is used."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load librt_hip.so.  Fails loudly if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C ray_tracing_amd/csrc` "
            "(there is no fallback implementation)" % LIB_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    u32, u64, vp, i = C.c_uint32, C.c_uint64, C.c_void_p, C.c_int
    L.rt_init.argtypes = [i]
    L.rt_shutdown.argtypes = []
    L.rt_last_error.argtypes = []
    L.rt_last_error.restype = C.c_char_p
    L.rt_abi_version.argtypes = []
    L.rt_render.argtypes = [C.POINTER(abi.RtRequest), vp]
    L.rt_render_multi.argtypes = [C.POINTER(abi.RtRequest), vp, i, vp]
    L.rt_render_tiles.argtypes = [C.POINTER(abi.RtRequest), vp]
    L.rt_accumulate_tiles.argtypes = [C.POINTER(abi.RtRequest), u32, vp]
    L.rt_compose.argtypes = [vp, u32, vp, u32, u32, u32, vp]
    L.rt_scene_create.argtypes = [vp, u32, vp, u32, C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_render_device_async.argtypes = [vp, C.POINTER(abi.RtFrame), vp, vp, vp, u32, i, vp, vp]
    L.rt_render_rows_device_async.argtypes = [vp, C.POINTER(abi.RtFrame), u32, u32, i, vp, vp]
    L.rt_render_rows_device_noalloc.argtypes = [vp, C.POINTER(abi.RtFrame), u32, u32, i, vp, vp]
    L.rt_accumulate_device_async.argtypes = [vp, C.POINTER(abi.RtFrame), u32, vp, vp, vp, u32, vp, vp]
    L.rt_resolve_device_async.argtypes = [vp, u64, u32, vp, vp]
    L.rt_synchronize.argtypes = [vp]
    L.rt_get_stats.argtypes = [C.POINTER(abi.RtStats)]
    L.rt_get_launch_ms.argtypes = [u32, vp]
    L.rt_get_launch_ms.restype = i
    L.rt_get_phase_cycles.argtypes = [vp]
    L.rt_trace_rays.argtypes = [vp, vp, u32, u32, u32, vp]
    L.rt_selftest_sqrt.argtypes = [vp]
    L.rt_scene_scan_info.argtypes = [vp, C.POINTER(abi.RtScanInfo)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))


def init(device=0):
    check(lib().rt_init(int(device)))


def shutdown():
    check(lib().rt_shutdown())


class Scene:
    """Device-resident World (reference src/main.rs:168-172)."""

    def __init__(self, spheres, materials):
        self.spheres = np.ascontiguousarray(spheres, dtype=abi.SPHERE_DTYPE)
        self.materials = np.ascontiguousarray(materials, dtype=abi.MATERIAL_DTYPE)
        h = C.c_void_p()
        check(lib().rt_scene_create(abi.ptr(self.spheres), len(self.spheres),
                                    abi.ptr(self.materials), len(self.materials), C.byref(h)))
        self.handle = h

    def scan_info(self):
        """Which filter World::hit's scan runs on for this scene (rt_scene_scan_info)."""
        info = abi.RtScanInfo()
        check(lib().rt_scene_scan_info(self.handle, C.byref(info)))
        return info

    def close(self):
        if self.handle is not None:
            lib().rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_frame(camera, width, height, spp, max_depth, seed, flags=0):
    f = abi.RtFrame()
    f.image_width, f.image_height = int(width), int(height)
    f.max_depth, f.samples_per_pixel = int(max_depth), int(spp)
    f.seed, f.flags = int(seed) & 0xFFFFFFFF, int(flags)
    f.camera = camera
    return f


def make_request(spheres, materials, camera, width, height, spp, max_depth, seed, flags=0,
                 tiles=None):
    """Returns (request, keepalive): the arrays must outlive the call that uses the request."""
    spheres = np.ascontiguousarray(spheres, dtype=abi.SPHERE_DTYPE)
    materials = np.ascontiguousarray(materials, dtype=abi.MATERIAL_DTYPE)
    r = abi.RtRequest()
    r.image_width, r.image_height = int(width), int(height)
    r.max_depth, r.samples_per_pixel = int(max_depth), int(spp)
    r.seed, r.flags = int(seed) & 0xFFFFFFFF, int(flags)
    r.camera = camera
    r.spheres, r.n_spheres = abi.ptr(spheres), len(spheres)
    r.materials, r.n_materials = abi.ptr(materials), len(materials)
    keep = [spheres, materials]
    if tiles is not None:
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        r.tiles, r.n_tiles = abi.ptr(tiles), len(tiles)
        keep.append(tiles)
    return r, keep


def render(spheres, materials, camera, width, height, spp, max_depth, seed, flags=0, tiles=None):
    """rt_render: resolved framebuffer, shape (H, W, 3), row 0 = bottom of the picture.  With
    `tiles`, exactly those tiles are rendered and every other pixel is (0, 0, 0)."""
    req, keep = make_request(spheres, materials, camera, width, height, spp, max_depth, seed, flags,
                             tiles)
    out = np.empty((height, width, 3), dtype=np.float32)
    check(lib().rt_render(C.byref(req), abi.ptr(out)))
    return out


def render_multi(spheres, materials, camera, width, height, spp, max_depth, seed, devices, flags=0,
                 tiles=None):
    """rt_render_multi: one process, several GPUs (a device may be listed more than once)."""
    req, keep = make_request(spheres, materials, camera, width, height, spp, max_depth, seed, flags,
                             tiles)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    out = np.empty((height, width, 3), dtype=np.float32)
    check(lib().rt_render_multi(C.byref(req), abi.ptr(dev), len(dev), abi.ptr(out)))
    return out


def render_tiles(spheres, materials, camera, width, height, spp, max_depth, seed, tiles, flags=0):
    """rt_render_tiles: concatenated per-tile un-normalised sums, shape (sum w*h, 3)."""
    req, keep = make_request(spheres, materials, camera, width, height, spp, max_depth, seed, flags,
                             tiles)
    t = keep[2]
    n = int((t["width"].astype(np.int64) * t["height"]).sum())
    out = np.empty((n, 3), dtype=np.float32)
    check(lib().rt_render_tiles(C.byref(req), abi.ptr(out)))
    return out


def accumulate_tiles(spheres, materials, camera, width, height, spp, max_depth, seed, tiles,
                     sample_begin, sums, flags=0):
    """rt_accumulate_tiles: adds samples [sample_begin, sample_begin + spp) to `sums` in place."""
    req, keep = make_request(spheres, materials, camera, width, height, spp, max_depth, seed, flags,
                             tiles)
    assert sums.dtype == np.float32 and sums.flags["C_CONTIGUOUS"]
    check(lib().rt_accumulate_tiles(C.byref(req), int(sample_begin), abi.ptr(sums)))
    return sums


def compose(tiles, tile_sums, width, height, spp):
    tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
    tile_sums = np.ascontiguousarray(tile_sums, dtype=np.float32)
    out = np.empty((height, width, 3), dtype=np.float32)
    check(lib().rt_compose(abi.ptr(tiles), len(tiles), abi.ptr(tile_sums), width, height, spp,
                           abi.ptr(out)))
    return out


def render_rows_device_async(scene, frame, row_begin, row_end, resolve, d_out_ptr, stream_ptr=None):
    check(lib().rt_render_rows_device_async(scene.handle, C.byref(frame), int(row_begin),
                                            int(row_end), int(bool(resolve)),
                                            C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr or 0)))


def render_rows_device_noalloc(scene, frame, row_begin, row_end, resolve, d_out_ptr, stream_ptr=None):
    """rt_render_rows_device_noalloc: the same rows, with counters and partial sums in the
    library's fixed device buffer (no allocation, no wait)."""
    check(lib().rt_render_rows_device_noalloc(scene.handle, C.byref(frame), int(row_begin),
                                              int(row_end), int(bool(resolve)),
                                              C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr or 0)))


def render_device_async(scene, frame, tiles, out_offset, out_stride, resolve, d_out_ptr,
                        stream_ptr=None):
    tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
    off = np.ascontiguousarray(out_offset, dtype=np.uint64)
    stride = np.ascontiguousarray(out_stride, dtype=np.uint32)
    assert len(off) == len(tiles) == len(stride)
    check(lib().rt_render_device_async(scene.handle, C.byref(frame), abi.ptr(tiles), abi.ptr(off),
                                       abi.ptr(stride), len(tiles), int(bool(resolve)),
                                       C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr or 0)))


def accumulate_device_async(scene, frame, sample_begin, tiles, out_offset, out_stride, d_sums_ptr,
                            stream_ptr=None):
    """rt_accumulate_device_async: adds samples [sample_begin, sample_begin + frame.spp) to the
    un-normalised sums at d_sums_ptr."""
    tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
    off = np.ascontiguousarray(out_offset, dtype=np.uint64)
    stride = np.ascontiguousarray(out_stride, dtype=np.uint32)
    assert len(off) == len(tiles) == len(stride)
    check(lib().rt_accumulate_device_async(scene.handle, C.byref(frame), int(sample_begin),
                                           abi.ptr(tiles), abi.ptr(off), abi.ptr(stride),
                                           len(tiles), C.c_void_p(d_sums_ptr),
                                           C.c_void_p(stream_ptr or 0)))


def resolve_device_async(d_sums_ptr, n_pixels, spp, d_rgb_ptr, stream_ptr=None):
    check(lib().rt_resolve_device_async(C.c_void_p(d_sums_ptr), int(n_pixels), int(spp),
                                        C.c_void_p(d_rgb_ptr), C.c_void_p(stream_ptr or 0)))


def synchronize(stream_ptr=None):
    check(lib().rt_synchronize(C.c_void_p(stream_ptr or 0)))


def get_stats():
    s = abi.RtStats()
    check(lib().rt_get_stats(C.byref(s)))
    return s


def get_launch_ms(n_last):
    out = np.zeros(int(n_last), dtype=np.float32)
    got = lib().rt_get_launch_ms(int(n_last), abi.ptr(out))
    if got < 0:
        check(got)
    return out[:got]


def get_phase_cycles():
    out = np.zeros(16, dtype=np.uint64)
    check(lib().rt_get_phase_cycles(abi.ptr(out)))
    return out


def selftest_sqrt():
    """(differing inputs of the guarded square root, of the unguarded one where it is used, of the
    unguarded one inside 0 < |x| < 2^-96, of 1 / sqrt(x)) over all 2^32 floats; see
    rt_selftest_sqrt."""
    out = np.zeros(4, dtype=np.uint64)
    check(lib().rt_selftest_sqrt(abi.ptr(out)))
    return tuple(int(v) for v in out)


def trace_rays(scene, rays, seed, flags=0):
    rays = np.ascontiguousarray(rays, dtype=abi.RAY_IN_DTYPE)
    out = np.zeros(len(rays), dtype=abi.RAY_OUT_DTYPE)
    check(lib().rt_trace_rays(scene.handle, abi.ptr(rays), len(rays), int(seed) & 0xFFFFFFFF,
                              int(flags), abi.ptr(out)))
    return out
