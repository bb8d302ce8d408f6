"""The benchmark configurations' own workloads on the GPU.

 * the 10 004-sphere brute-force scene (the scan streams it through LDS in nine chunks):
   per-ray records and a small image against the oracle, bit for bit, then the full
   1920x1080 @ 64 spp frame through properties that do not depend on size;
 * 3840x2160 @ 500 spp: the frame's own sample-slice plan (body + taper), the eight row
   shares of an 8-GPU run rendered one after the other and put back together, against the
   frame rendered whole;
 * the body + taper slice plan against the reference's sample order on one >= 2^20-pixel frame.
"""
import numpy as np
import pytest
import torch

import oracle_lib as O
from ray_tracing_amd import abi, hip, host, multi_gpu
from test_gpu_parity import FLOAT_FIELDS, INT_FIELDS, default_camera, make_rays, oracle_sums

pytestmark = pytest.mark.gpu

STRICT = abi.RT_FLAG_STRICT_FP
SEQ = abi.RT_FLAG_SEQUENTIAL_SUM


@pytest.fixture(scope="module", autouse=True)
def device():
    hip.init(0)
    yield


@pytest.fixture(scope="module")
def world10k():
    s, m = O.random_world(1, -50, 50)
    assert len(s) == 10004
    return s, m


def test_10k_brute_force_scan_per_ray_matches_world_hit(world10k):
    """World::hit's loop over all 10 004 spheres (src/main.rs:175-189) as five LDS chunks with the
    closest hit carried across them: every record field equals ro_bounce's, in the literal and
    in the conservative scan."""
    s, m = world10k
    cam = default_camera()
    rays = make_rays((s, m), cam, 700, seed=12)
    assert len(rays) >= 600
    ref = O.bounce(O.make_world(s, m), rays, seed=31, pow5_mode=O.RO_POW5_MUL)
    scene = hip.Scene(s, m)
    for flags in (STRICT, 0):
        got = hip.trace_rays(scene, rays, seed=31, flags=flags)
        for f in INT_FIELDS:
            assert np.array_equal(got[f], ref[f]), (flags, f)
        for f in FLOAT_FIELDS:
            assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), (flags, f)
    # hits spread over the chunks of 1200 spheres, the first and the last among them
    hit = ref["sphere"][ref["hit"] == 1]
    chunks = set(int(c) for c in np.unique(hit // 1200))
    assert {0, 8} <= chunks and len(chunks) >= 6, chunks
    scene.close()


def test_10k_brute_force_image_matches_the_oracle(world10k):
    s, m = world10k
    w, h, spp = 48, 27, 2
    cam = default_camera(w / h)
    tiles = O.job_queue(w, h, 64, 64)
    _, ref, ctr = oracle_sums(s, m, cam, w, h, spp, 50, 2, tiles)
    for flags in (SEQ, SEQ | STRICT):
        sums = hip.render_tiles(s, m, cam, w, h, spp, 50, 2, tiles, flags=flags)
        assert np.array_equal(sums.view(np.uint32), ref.view(np.uint32)), flags
        st = hip.get_stats()
        assert (st.rays, st.samples) == (ctr.rays, ctr.samples)
        assert st.sphere_tests == ctr.rays * 10004
        assert st.threads_per_workgroup == (512 if flags & STRICT else 1024)   # the chunked kernels


def test_10k_brute_force_full_size_frame_properties():
    """1920x1080 @ 64 spp on the chunked scan (what the benchmark's 10k configuration runs):
    determinism, counters, resolve range, and a strip rendered alone equals its rows."""
    w, h, spp = 1920, 1080, 64
    world = host.create_random_world(1, (-50, 50))
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=0)
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    hip.render_rows_device_async(scene, frame, 0, h, True, a.data_ptr(), stream)
    torch.cuda.synchronize()
    st = hip.get_stats()
    assert st.samples == w * h * spp and st.rays >= st.samples
    assert st.sphere_tests == st.rays * 10004 and st.threads_per_workgroup == 1024
    hip.render_rows_device_async(scene, frame, 0, h, True, b.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    strip = torch.zeros((8, w, 3), dtype=torch.float32, device="cuda:0")
    hip.render_rows_device_async(scene, frame, 300, 308, True, strip.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(strip.view(torch.int32), a[300:308].view(torch.int32))
    img = a.cpu().numpy()
    ok = ~np.isnan(img)
    assert img[ok].min() >= 0.0 and img[ok].max() <= np.float32(0.999)
    assert np.all(img[-1, :, 2] == np.float32(0.999))
    scene.close()


def test_headline_frame_at_its_own_size_and_sample_count():
    """1920x1080 @ 100 spp, max_depth 50, 488 spheres (the configuration the metric is quoted on):
    the frame twice (same bits), with the filter on the vector ALU and through the hierarchy (same
    bits), as the eight shares of an 8-GPU run (same bits, same counters), and ten of its rows
    with every pixel's samples added in the reference's order (within the association bound)."""
    w, h, spp = 1920, 1080, 100
    world = host.create_random_world(1, (-11, 11))
    assert len(world.spheres) == 488
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    stream = torch.cuda.current_stream().cuda_stream

    def whole(flags):
        out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
        frame = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=flags)
        hip.render_rows_device_async(scene, frame, 0, h, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        return out, hip.get_stats()

    full, st = whole(0)
    assert st.samples == w * h * spp and st.slices == 50 and st.threads_per_workgroup == 1024
    assert st.sphere_tests == st.rays * 488
    again, st2 = whole(0)
    assert torch.equal(again.view(torch.int32), full.view(torch.int32)) and st2.rays == st.rays
    for flags in (abi.RT_FLAG_SCAN_VALU, abi.RT_FLAG_ACCEL):
        other, sto = whole(flags)
        assert torch.equal(other.view(torch.int32), full.view(torch.int32)), flags
        assert sto.rays == st.rays
    frame = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=0)
    plan = multi_gpu.RowPlan(w, h, 8)
    stacked = torch.zeros((8, plan.rows_per_rank, w, 3), dtype=torch.float32, device="cuda:0")
    rays = 0
    for rank in range(8):
        tiles, offs, strides = plan.tiles_for(rank)
        hip.render_device_async(scene, frame, tiles, offs, strides, True, stacked[rank].data_ptr(), stream)
        torch.cuda.synchronize()
        rays += hip.get_stats().rays
    assert rays == st.rays
    row_map = torch.from_numpy(plan.frame_row_map()).to("cuda:0")
    assert torch.equal(stacked.reshape(-1, w, 3).index_select(0, row_map).view(torch.int32),
                       full.view(torch.int32))
    # rows 500..509 as un-normalised sums: sliced (50 items of 2 samples) against sequential
    sums = {}
    for name, flags in (("sliced", 0), ("sequential", SEQ)):
        out = torch.zeros((10, w, 3), dtype=torch.float32, device="cuda:0")
        fr = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=flags)
        hip.render_rows_device_async(scene, fr, 500, 510, False, out.data_ptr(), stream)
        torch.cuda.synchronize()
        sums[name] = out.cpu().numpy().astype(np.float64)
    err = np.abs(sums["sliced"] - sums["sequential"]) / np.maximum(np.abs(sums["sequential"]), 1.0)
    assert err.max() <= 2 * spp * 2.0 ** -24, err.max()
    resolved = np.sqrt(sums["sliced"].astype(np.float32) * np.float32(1.0 / spp))
    resolved = np.where(resolved > np.float32(0.999), np.float32(0.999), resolved).astype(np.float32)
    got = full[500:510].cpu().numpy()
    ok = ~np.isnan(resolved)
    assert np.array_equal(got[ok].view(np.uint32), resolved[ok].view(np.uint32))
    scene.close()


def test_dielectric_heavy_frame_at_1080p_1024spp():
    """The dielectric-heavy configuration at its throughput size, 1920x1080 @ 1024 spp,
    max_depth 50 (64 slices of 16 samples a pixel): the frame twice (same bits, same counters),
    with the filter on the vector ALU and through the hierarchy (same bits, same ray count), a
    strip of it rendered on its own (same bits), and every value inside the resolve's range."""
    w, h, spp = 1920, 1080, 1024
    world = host.create_random_world(1, (-11, 11), 0.1, 0.2, True)
    assert len(world.spheres) == 488
    glass = (world.materials["flags"] & abi.RT_MAT_DIELECTRIC) != 0
    assert glass.mean() > 0.75                            # four spheres in five are glass
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    stream = torch.cuda.current_stream().cuda_stream

    def rows(r0, r1, flags):
        out = torch.zeros((r1 - r0, w, 3), dtype=torch.float32, device="cuda:0")
        frame = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=flags)
        hip.render_rows_device_async(scene, frame, r0, r1, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        return out, hip.get_stats()

    full, st = rows(0, h, 0)
    assert st.samples == w * h * spp and st.slices == 64 and st.threads_per_workgroup == 1024
    assert st.rays > 3.3 * st.samples                     # glass keeps paths alive longer
    again, st2 = rows(0, h, 0)
    assert torch.equal(again.view(torch.int32), full.view(torch.int32))
    assert (st2.rays, st2.samples) == (st.rays, st.samples)
    for flags in (abi.RT_FLAG_SCAN_VALU, abi.RT_FLAG_ACCEL):
        other, sto = rows(0, h, flags)
        assert torch.equal(other.view(torch.int32), full.view(torch.int32)), flags
        assert sto.rays == st.rays
    strip, _ = rows(400, 440, 0)
    assert torch.equal(strip.view(torch.int32), full[400:440].view(torch.int32))
    img = full.cpu().numpy()
    ok = ~np.isnan(img)
    assert img[ok].min() >= 0.0 and img[ok].max() <= np.float32(0.999)
    scene.close()


def test_4k_500spp_eight_row_shares_reassemble_to_the_whole_frame():
    """3840x2160 @ 500 spp (54 slices of 9 samples and a taper of 8): the frame rendered whole,
    then as the eight shares RowPlan deals to 8 GPUs, one after the other on this GPU through the
    entry point the ranks use, reassembled with the plan's row map: the same bits, the same
    counters."""
    w, h, spp = 3840, 2160, 500
    world = host.create_random_world(1, (-11, 11))
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=0)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    hip.render_rows_device_async(scene, frame, 0, h, True, full.data_ptr(), stream)
    torch.cuda.synchronize()
    st = hip.get_stats()
    assert st.samples == w * h * spp and st.slices == 62
    total_rays = st.rays
    plan = multi_gpu.RowPlan(w, h, 8)
    stacked = torch.zeros((8, plan.rows_per_rank, w, 3), dtype=torch.float32, device="cuda:0")
    rays = 0
    for rank in range(8):
        tiles, offs, strides = plan.tiles_for(rank)
        hip.render_device_async(scene, frame, tiles, offs, strides, True, stacked[rank].data_ptr(),
                                stream)
        torch.cuda.synchronize()
        s = hip.get_stats()
        assert s.samples == plan.rows_of(rank) * w * spp
        rays += s.rays
    assert rays == total_rays
    row_map = torch.from_numpy(plan.frame_row_map()).to("cuda:0")
    again = stacked.reshape(-1, w, 3).index_select(0, row_map)
    assert torch.equal(again.view(torch.int32), full.view(torch.int32))
    # and the whole frame reproduces itself
    hip.render_rows_device_async(scene, frame, 0, h, True, again.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), full.view(torch.int32))
    scene.close()


@pytest.mark.parametrize("flags", [0, SEQ])
def test_row_block_lists_regular_ragged_and_irregular(flags):
    """Tile lists made of row blocks: at a constant pitch (what the trace kernel takes as one
    tile with a rule for its rows), the same with a short last block, and at an uneven pitch
    (looked up tile by tile).  Each equals the same rows of the frame rendered whole."""
    w, h, spp = 160, 100, 6
    world = host.create_random_world(1, (-4, 4))
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(camera.m, w, h, spp, 50, 3, flags=flags)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    hip.render_rows_device_async(scene, frame, 0, h, True, full.data_ptr(), stream)
    torch.cuda.synchronize()
    plan = multi_gpu.RowPlan(w, h, 3)                 # 12 blocks of 8 rows and one of 4
    lists = [plan.tiles_for(r) for r in range(3)]     # rank 0's last block is the short one
    assert int(lists[0][0]["height"][-1]) == 4
    # uneven pitch: blocks 0, 1, 4, 5 and 11
    blocks = [0, 1, 4, 5, 11]
    t = np.array([(0, 8 * b, w, 8) for b in blocks], dtype=abi.TILE_DTYPE)
    lists.append((t, np.arange(len(blocks), dtype=np.uint64) * 8 * w, np.full(len(blocks), w, dtype=np.uint32)))
    # constant pitch in the image but not in the output (a gap after the second block)
    t = np.array([(0, 16 * k, w, 8) for k in range(4)], dtype=abi.TILE_DTYPE)
    lists.append((t, np.array([0, 8 * w, 24 * w, 32 * w], dtype=np.uint64), np.full(4, w, dtype=np.uint32)))
    for tiles, offs, strides in lists:
        rows = int(offs[-1]) // w + int(tiles["height"][-1])
        out = torch.full((rows, w, 3), -1.0, dtype=torch.float32, device="cuda:0")
        hip.render_device_async(scene, frame, tiles, offs, strides, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        for k in range(len(tiles)):
            y0, th, r0 = int(tiles["y"][k]), int(tiles["height"][k]), int(offs[k]) // w
            assert torch.equal(out[r0:r0 + th].view(torch.int32), full[y0:y0 + th].view(torch.int32)), (k, y0)
    scene.close()


def test_body_and_taper_slices_agree_with_the_reference_sample_order():
    """A frame of more than 2^20 pixels whose samples are cut into full-size items plus the
    tapered tail (4,4,2,2,1,1,1,1), against the same frame with every pixel's samples added in
    the reference's order (`pixel_color += color`, src/cpu/job.rs:58).  The paths are the same
    (equal ray counts), so the two sums of n = 300 floats differ only by the association of the
    additions: each is within (n-1) u of the exact sum relative to the sum of the terms'
    magnitudes (u = 2^-24), so they are within 2 n u = 3.6e-5 of that sum of each other.  The sum
    of magnitudes is not known per pixel: it is at most n (a sample's radiance is a product of
    albedos, below 1 in magnitude, and the sky) and about the pixel's sum itself wherever no
    negative albedo makes the terms cancel, so the bound proper is asserted against n and the
    bulk of the pixels (median, 99.9 %) against the pixel's own sum; the test prints what it finds."""
    w, h, spp = 1920, 1080, 300
    world = host.create_random_world(1, (-11, 11))
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    stream = torch.cuda.current_stream().cuda_stream
    outs = {}
    for name, flags in (("sliced", 0), ("sequential", SEQ)):
        out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
        frame = hip.make_frame(camera.m, w, h, spp, 50, 1, flags=flags)
        hip.render_rows_device_async(scene, frame, 0, h, False, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        outs[name] = (out.cpu().numpy().astype(np.float64), st)
    (a, st_a), (b, st_b) = outs["sliced"], outs["sequential"]
    assert st_a.slices == 56 and st_b.slices == 1          # 48 items of 6 samples + the taper of 8
    assert (st_a.rays, st_a.samples) == (st_b.rays, st_b.samples)
    # relative to the magnitude of what was added up (at least 1: a pixel whose negative-albedo
    # samples cancel has a small sum of larger terms)
    scale = np.maximum(np.abs(b), 1.0)
    err = np.abs(a - b) / scale
    print("sliced vs sequential sums, relative difference: max %.3g, median %.3g, 99.9%% %.3g"
          % (err.max(), np.median(err), np.quantile(err, 0.999)))
    assert np.abs(a - b).max() <= 2 * spp * 2.0 ** -24 * spp
    assert np.quantile(err, 0.999) <= 2 * spp * 2.0 ** -24
    assert np.median(err) <= 2e-6
    scene.close()
