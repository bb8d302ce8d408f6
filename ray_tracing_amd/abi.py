"""ctypes / numpy mirrors of the POD types declared in include/rt_abi.h.

These carry the reference's scene types across the C ABI: Sphere (reference src/main.rs:119-125),
Material (src/main.rs:108-117), Camera (src/main.rs:51-55), TileJob (src/cpu/job.rs:6-13) and
Dispatch (src/cpu/mod.rs:19-30).  Layouts must match the header byte for byte;
tests/test_abi_library.py checks the sizes against a C program compiled from the header.
"""
import ctypes as C

import numpy as np

RT_ABI_VERSION = 4

RT_MAT_METALLIC = 1
RT_MAT_DIELECTRIC = 2

RT_FLAG_STRICT_FP = 1
RT_FLAG_SEQUENTIAL_SUM = 2
RT_FLAG_ACCEL = 4
RT_FLAG_SCAN_VALU = 8

RT_KIND_MISS = 0
RT_KIND_DIFFUSE = 1
RT_KIND_METAL = 2
RT_KIND_DIELECTRIC = 3

RT_RNG_SCENE_STREAM = 0xFFFFFFF1

SPHERE_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("radius", "<f4"), ("material_id", "<u4")]
)
MATERIAL_DTYPE = np.dtype(
    [("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("flags", "<u4"),
     ("metallic_strength", "<f4"), ("ir", "<f4")]
)
TILE_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("width", "<u4"), ("height", "<u4")])
RAY_IN_DTYPE = np.dtype([(n, "<f4") for n in ("ox", "oy", "oz", "dx", "dy", "dz")])
RAY_OUT_DTYPE = np.dtype(
    [(n, "<u4") for n in ("hit", "sphere", "material", "kind", "front_face", "scattered", "draws")]
    + [(n, "<f4") for n in ("t", "px", "py", "pz", "nx", "ny", "nz", "sx", "sy", "sz",
                            "ar", "ag", "ab")]
)


class RtCamera(C.Structure):
    _fields_ = [("view_inv", C.c_float * 16), ("proj_inv", C.c_float * 16)]


class RtRequest(C.Structure):
    _fields_ = [
        ("image_width", C.c_uint32),
        ("image_height", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("seed", C.c_uint32),
        ("flags", C.c_uint32),
        ("camera", RtCamera),
        ("spheres", C.c_void_p),
        ("n_spheres", C.c_uint32),
        ("materials", C.c_void_p),
        ("n_materials", C.c_uint32),
        ("tiles", C.c_void_p),
        ("n_tiles", C.c_uint32),
    ]


class RtFrame(C.Structure):
    _fields_ = [
        ("image_width", C.c_uint32),
        ("image_height", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("seed", C.c_uint32),
        ("flags", C.c_uint32),
        ("camera", RtCamera),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("samples", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("kernel_ms", C.c_float),
        ("rays_per_lane", C.c_uint32),
        ("slices", C.c_uint32),
        ("workgroups", C.c_uint32),
        ("threads_per_workgroup", C.c_uint32),
    ]


class RtScanInfo(C.Structure):
    _fields_ = [
        ("matrix_filter", C.c_uint32),
        ("listed_spheres", C.c_uint32),
        ("scale", C.c_float),
        ("origin_limit", C.c_float),
    ]


def camera_from_arrays(view_inv, proj_inv):
    cam = RtCamera()
    v = np.asarray(view_inv, dtype=np.float32).reshape(16)
    p = np.asarray(proj_inv, dtype=np.float32).reshape(16)
    for i in range(16):
        cam.view_inv[i] = float(v[i])
        cam.proj_inv[i] = float(p[i])
    return cam


def ptr(arr):
    """void* of a C-contiguous numpy array (the array must outlive the call)."""
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)
