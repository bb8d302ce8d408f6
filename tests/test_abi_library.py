"""The C-ABI library loads without a GPU and exports every symbol include/rt_abi.h declares;
the ctypes / numpy mirrors have the header's layout.  CPU only: no compute call is made."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from ray_tracing_amd import abi, hip, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_abi.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 15
    L = hip.lib()
    for n in names:
        assert hasattr(L, n), "librt_hip.so does not export %s" % n
    assert sorted(hip.EXPORTS) == names
    assert L.rt_abi_version() == abi.RT_ABI_VERSION


def test_host_mirror_exports():
    H = host.lib()
    for n in host.HOST_EXPORTS:
        assert hasattr(H, n)


def test_calls_fail_cleanly_before_init():
    L = hip.lib()
    st = abi.RtStats()
    assert L.rt_get_stats(C.byref(st)) == -2           # RT_ERR_NOT_INITIALIZED
    assert b"rt_init" in L.rt_last_error()
    assert L.rt_shutdown() == 0


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "rt_abi.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
        'sizeof(rt_sphere),sizeof(rt_material),sizeof(rt_camera),sizeof(rt_tile),'
        'sizeof(rt_request),sizeof(rt_frame),sizeof(rt_stats),sizeof(rt_ray_in),sizeof(rt_ray_out),'
        'offsetof(rt_request,spheres),offsetof(rt_request,tiles),sizeof(rt_scan_info));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True,
                                          text=True).stdout.split()]
    mine = [abi.SPHERE_DTYPE.itemsize, abi.MATERIAL_DTYPE.itemsize, C.sizeof(abi.RtCamera),
            abi.TILE_DTYPE.itemsize, C.sizeof(abi.RtRequest), C.sizeof(abi.RtFrame),
            C.sizeof(abi.RtStats), abi.RAY_IN_DTYPE.itemsize, abi.RAY_OUT_DTYPE.itemsize,
            abi.RtRequest.spheres.offset, abi.RtRequest.tiles.offset, C.sizeof(abi.RtScanInfo)]
    assert got == mine
    assert got[:4] == [20, 24, 128, 16]
