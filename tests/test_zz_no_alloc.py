"""rt_render_rows_device_noalloc against rt_render_rows_device_async, through the C ABI.

The entry point that may not allocate keeps its work counters and the waves' rings of parked
slice sums in one fixed device buffer of the library.  It must give the allocating entry point's
pixels bit for bit, on calls that follow each other (the second smaller than the first, so
that anything the first call left in the buffer would show), for frames of any size."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from ray_tracing_amd import abi, hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    hip.init(0)
    yield


def both(scene, cam, w, h, spp, flags, rows=None, resolve=True, seed=3, depth=50):
    r0, r1 = rows or (0, h)
    frame = hip.make_frame(cam, w, h, spp, depth, seed, flags)
    stream = torch.cuda.current_stream().cuda_stream
    ref = torch.full((r1 - r0, w, 3), -7.0, dtype=torch.float32, device="cuda:0")
    got = torch.full_like(ref, -9.0)
    hip.render_rows_device_async(scene, frame, r0, r1, resolve, ref.data_ptr(), stream)
    torch.cuda.synchronize()
    st_ref = hip.get_stats()
    hip.render_rows_device_noalloc(scene, frame, r0, r1, resolve, got.data_ptr(), stream)
    torch.cuda.synchronize()
    st = hip.get_stats()
    assert (st.rays, st.samples, st.slices) == (st_ref.rays, st_ref.samples, st_ref.slices)
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (w, h, spp, flags, rows)
    return st


def test_two_calls_in_a_row_the_second_smaller():
    s, m = O.random_world(1, -11, 11)
    scene = hip.Scene(s, m)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    # sliced samples: the waves' rings live in the fixed buffer
    st = both(scene, cam, 256, 144, 32, 0)
    assert st.slices > 1
    st = both(scene, cam, 96, 54, 8, 0)                      # smaller, fewer slices
    assert st.slices > 1
    both(scene, cam, 256, 144, 32, 0, rows=(10, 50), resolve=False)   # a band, un-normalised sums
    both(scene, cam, 64, 36, 5, abi.RT_FLAG_STRICT_FP)
    # one slice: no rings, only the counters
    st = both(scene, cam, 640, 360, 4, abi.RT_FLAG_SEQUENTIAL_SUM)
    assert st.slices == 1
    # the hierarchy was never built for this scene by an allocating call: refused, not built
    frame = hip.make_frame(cam, 64, 36, 2, 50, 1, abi.RT_FLAG_ACCEL)
    out = torch.zeros((36, 64, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    with pytest.raises(hip.RtError) as e:
        hip.render_rows_device_noalloc(scene, frame, 0, 36, True, out.data_ptr(), stream)
    assert e.value.code == -1 and "hierarchy" in str(e.value)
    both(scene, cam, 64, 36, 2, abi.RT_FLAG_ACCEL)           # the allocating call builds it first
    scene.close()


def test_the_whole_benchmark_frame_in_one_call():
    """1920x1080 at 100 spp is 50 slices a pixel: the waves' rings of parked slice sums fit the
    fixed buffer whatever the size of the frame (a wave holds fewer pixels open than it would with
    memory of the library's own choosing; the sums do not depend on that)."""
    s, m = O.random_world(1, -11, 11)
    scene = hip.Scene(s, m)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    st = both(scene, cam, 1920, 1080, 100, 0)
    assert st.slices == 50
    both(scene, cam, 1920, 1080, 100, 0, rows=(512, 560))    # a band of it, right after
    scene.close()


def test_two_calls_captured_into_a_graph_and_replayed_twice():
    """The use the entry point exists for: under stream capture nothing may allocate, wait or
    query.  Two calls (the two parts of a frame) are captured into one HIP graph; each replay must
    give the allocating entry point's pixels, and the counters of the last call are readable
    afterwards (its kernel time is not: no event pair is recorded on a captured stream)."""
    import math
    s, m = O.random_world(1, -11, 11)
    scene = hip.Scene(s, m)
    other = hip.Scene(s[:77], m)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    w, h, spp, cut = 256, 144, 32, 80
    frame = hip.make_frame(cam, w, h, spp, 50, 11, 0)
    ref = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    hip.render_rows_device_async(scene, frame, 0, h, True, ref.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    hip.render_rows_device_async(scene, frame, cut, h, True, ref[cut:].data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = hip.get_stats()                      # of the second part
    out = torch.zeros_like(ref)
    graph = torch.cuda.CUDAGraph()
    refused = None
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
        sp = torch.cuda.current_stream().cuda_stream
        hip.render_rows_device_noalloc(scene, frame, 0, cut, True, out.data_ptr(), sp)
        hip.render_rows_device_noalloc(scene, frame, cut, h, True, out[cut:].data_ptr(), sp)
        # a kernel variant and scene size this process has not launched yet (the vector filter on
        # a World of 77 spheres) would need a query first: refused while capturing, and nothing
        # is enqueued
        try:
            hip.render_rows_device_noalloc(other, hip.make_frame(cam, w, h, spp, 50, 11, abi.RT_FLAG_SCAN_VALU),
                                           0, cut, True, out.data_ptr(), sp)
        except hip.RtError as e:
            refused = e
    assert float(out.abs().max()) == 0.0        # capturing runs nothing
    for _ in range(2):
        out.fill_(-3.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    st = hip.get_stats()
    assert (st.rays, st.samples, st.slices) == (want.rays, want.samples, want.slices)
    assert math.isnan(st.kernel_ms)
    assert refused is not None and refused.code == -1 and "capture" in str(refused)
    scene.close()
    other.close()
