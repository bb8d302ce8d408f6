#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on the 1080p RTIOW-final scene (488 spheres, 100 spp,
max_depth 50) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one full frame: every pixel's `spp` paths through the HIP kernel behind the C ABI
(rt_render_device_async), scene resident in HBM.  With N > 1 the frame's rows are dealt to the
ranks in blocks of 8, each rank renders its blocks and one RCCL gather brings the framebuffer
to rank 0 (strong scaling: the frame is fixed).  Rank 0 prints ONE JSON line.

  value      whole-job Msamples/s = W*H*spp*steps / max-over-ranks wall time
  roofline   dominant kernel (trace_kernel).  Its ray/sphere filter runs on the matrix pipe
             (v_mfma_f32_16x16x32_f16, 32 slots = 64 flop per test as issued), so the bound is
             the dense f16 MFMA peak: achieved = 64 flop * n_spheres * rays / kernel time, with
             rays counted by the kernel itself and kernel time from HIP events on the launch
             stream (rank 0's launches).  `fp32_form` prices the same launch the way the
             reference states the test (17 flop, src/main.rs:131-135) against the FP32
             vector-ALU peak, the figure the metric names
  valu_scan  the same frame with the filter kept on the vector ALU (RT_FLAG_SCAN_VALU: seven
             fused f32 instructions and a compare per test), with its own roofline against the
             FP32 vector-ALU peak (N = 1 only)
  cpu_baseline  the CPU restatement of the reference's multithreaded tile renderer
             (oracle/, kind "port") timed on this host's cores on a reduced sample of the
             same workload (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3   # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz (MI355X)
FLOP_PER_TEST = 17              # one ray/sphere discriminant, src/main.rs:131-135 with a hoisted
FLOP_PER_TEST_EXECUTED = 14     # what the vector scan issues per test: 7 fused multiply-adds
                                # (and one compare, not counted)
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense f16 matrix peak: 1024 SIMDs x 16x16x32x2 flop / 16 clk x 2.4 GHz
FLOP_PER_TEST_MFMA = 64         # one (sphere, ray) entry of a 16x16x32 product: 32 slots x 2 flop

CONFIGS = {
    # the configuration the headline metric is quoted on
    "rtiow488_1080p_100spp_d50": dict(grid=(-11, 11), width=1920, height=1080, spp=100, depth=50),
    # same scene at 4K, 500 spp
    "rtiow488_2160p_500spp_d50": dict(grid=(-11, 11), width=3840, height=2160, spp=500, depth=50),
    # 10 004 spheres, LDS-chunked scan
    "rtiow10k_1080p_64spp_d50": dict(grid=(-50, 50), width=1920, height=1080, spp=64, depth=50),
    # dielectric-heavy
    "glass488_1080p_1024spp_d50": dict(grid=(-11, 11), width=1920, height=1080, spp=1024, depth=50,
                                       diffuse_below=0.1, metal_below=0.2, glass=True),
}
DEFAULT_CONFIG = "rtiow488_1080p_100spp_d50"


def logical_cores():
    """What num_cpus::get() (the reference's thread count, src/main.rs:923) reports on Linux:
    the CPUs this process may run on, capped by a cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max ..."
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                                # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(cfg, seed):
    """Time the oracle's dispatch_work_cpu_multithreaded on a bounded sample of the workload:
    the same scene, camera, depth, resolution and 64x64 tile queue (510 tiles at 1080p, so that
    every thread has tiles to pop, as in the reference) at a bounded number of samples per pixel
    (cost is linear in spp), on every logical core the process may use, like the reference's
    `num_cpus::get()` (src/main.rs:923, src/cpu/multi.rs:131)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = logical_cores()
    w, h = cfg["width"], cfg["height"]
    lo, hi = cfg["grid"]
    # about 10-30 s of CPU work in total: ~0.3 Msamples/s per core at 488 spheres
    per_core_rate = 0.3e6 * 488.0 / ((hi - lo) ** 2 + 4)
    spp = int(max(1, min(cfg["spp"], round(20.0 * per_core_rate * min(cores, 16) / (w * h)))))
    s, m = O.random_world(1, lo, hi, cfg.get("diffuse_below", 0.8), cfg.get("metal_below", 0.95),
                          1 if cfg.get("glass") else 0)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    _, secs, ctr = O.render(s, m, cam, w, h, spp, cfg["depth"], seed, mode=O.RO_MODE_RECURSIVE,
                            threads=cores)
    return {
        "value": round(ctr.samples / secs * 1e-6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "cores_source": "CPUs this process may run on, capped by its cgroup quota "
                        "(num_cpus::get() semantics); os.cpu_count() = %d" % (os.cpu_count() or 0),
        "kind": "port",
        "sample": "%dx%d @ %d spp, same scene/camera/max_depth/tile queue (%d samples, %.1f s, "
                  "%.3f rays/sample)" % (w, h, spp, ctr.samples, secs, ctr.rays / max(ctr.samples, 1)),
    }


def drop_in_leg(hip, scene_host, camera, W, H, spp, depth, seed, steps, kernel_ms, n_devices):
    """The call a host of the reference makes: rt_render with host pointers in and out (world
    arrays in host memory, framebuffer in pageable host memory), wall time per call.  The first
    call uploads the World and allocates; later calls on the same World reuse both, which is the
    steady state timed here.  Where the process sees several GPUs, rt_render_multi over all of
    them is timed the same way."""
    import numpy as np
    t0 = time.perf_counter()
    fb = hip.render(scene_host.spheres, scene_host.materials, camera.m, W, H, spp, depth, seed)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        fb = hip.render(scene_host.spheres, scene_host.materials, camera.m, W, H, spp, depth, seed)
    per_call = (time.perf_counter() - t0) / steps
    out = {
        "entry": "rt_render (host pointers in, pageable host framebuffer out)",
        "ms_per_call": round(per_call * 1e3, 3),
        "first_call_ms": round(first * 1e3, 3),
        "overhead_vs_kernel_ms": round(per_call * 1e3 - kernel_ms, 3),
        "value": round(W * H * spp / per_call * 1e-6, 2),
        "unit": "Msamples/s",
        "mean_rgb": [round(float(v), 5) for v in np.nanmean(fb, axis=(0, 1))],
    }
    if n_devices > 1:
        devices = list(range(n_devices))
        many = hip.render_multi(scene_host.spheres, scene_host.materials, camera.m, W, H, spp,
                                depth, seed, devices)
        t0 = time.perf_counter()
        for _ in range(steps):
            many = hip.render_multi(scene_host.spheres, scene_host.materials, camera.m, W, H, spp,
                                    depth, seed, devices)
        per_call = (time.perf_counter() - t0) / steps
        same = np.array_equal(np.nan_to_num(many, nan=-1.0).view(np.uint32),
                              np.nan_to_num(fb, nan=-1.0).view(np.uint32))
        out["multi"] = {
            "entry": "rt_render_multi (one process, %d GPUs)" % n_devices,
            "ms_per_call": round(per_call * 1e3, 3),
            "value": round(W * H * spp / per_call * 1e-6, 2),
            "unit": "Msamples/s",
            "same_pixels_as_rt_render": bool(same),
        }
    return out


def profiled_traffic(config, kernel):
    """Bytes per launch of `kernel` that left the L2 for memory, from the committed rocprofv3 PMC
    passes of this same command (profiles/: FETCH_SIZE and WRITE_SIZE, in KiB, each collected in
    a pass of its own), and where they come from.  FETCH_SIZE is doubled as the gfx950 correction
    for wide reads prescribes, which makes the figure an upper bound for this kernel's narrow
    reads.  The counters sit on the L2's memory side and count what the 256 MiB Infinity Cache
    then serves as well: nearly all of this kernel's traffic is the waves' rings of parked slice
    sums (79 MB in all for this frame), written and read back within a fraction of a millisecond.
    (None, reason) when no profile of `config` is in the tree: the counters cannot be read from
    inside a timed run."""
    import csv
    for rnd in ("round3", "round2", "round1"):
        path = os.path.join(ROOT, "profiles", rnd, config + "_pmc_per_launch.csv")
        if os.path.exists(path):
            break
    else:
        return None, "no committed PMC profile of this configuration"
    kib = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["kernel"] == kernel and row["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                kib[row["counter"]] = float(row["value_per_launch"])
    rel = os.path.relpath(path, ROOT)
    if len(kib) != 2:
        return None, rel + " has no FETCH_SIZE / WRITE_SIZE rows for " + kernel
    commit = ""
    note = os.path.join(os.path.dirname(path), config + "_profiled_commit.txt")
    if os.path.exists(note):
        commit = ", library built from commit " + open(note).read().strip()
    return (int((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024),
            "%s (rocprofv3 --pmc, bytes per launch of an earlier run of this command%s)" % (rel, commit))


def profiled_issue_bound(config, kernel, kernel_ms):
    """What share of the SIMDs' cycles the kernel's vector and matrix instructions need just to
    issue: instruction counts per launch from the committed rocprofv3 PMC passes of this command
    (SQ_INSTS_VALU, SQ_INSTS_MFMA), priced at what a SIMD was measured to issue them at with four
    waves (2.2 cycles a vector instruction, 21.6 a v_mfma_f32_16x16x32_f16, no overlap between the
    two: tools/ubench), against this run's kernel time at the nominal clock.  The lower figure
    takes SQ_INSTS_VALU to include the matrix instructions.  None without a profile."""
    import csv
    for rnd in ("round3", "round2"):
        path = os.path.join(ROOT, "profiles", rnd, config + "_pmc_per_launch.csv")
        if os.path.exists(path):
            break
    else:
        return None
    if not kernel_ms > 0:
        return None
    count = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["kernel"] == kernel and row["counter"] in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA"):
                count[row["counter"]] = float(row["value_per_launch"])
    if len(count) != 2:
        return None
    simd_cycles = kernel_ms * 1e-3 * 2.4e9 * 256 * 4
    mfma = count["SQ_INSTS_MFMA"] * 21.6
    return {
        "valu_insts_per_launch": count["SQ_INSTS_VALU"],
        "mfma_insts_per_launch": count["SQ_INSTS_MFMA"],
        "cycles_per_valu": 2.2,
        "cycles_per_mfma": 21.6,
        "frac_low": round(((count["SQ_INSTS_VALU"] - count["SQ_INSTS_MFMA"]) * 2.2 + mfma) / simd_cycles, 4),
        "frac_high": round((count["SQ_INSTS_VALU"] * 2.2 + mfma) / simd_cycles, 4),
        "note": "share of all SIMD-cycles (nominal clock) that issuing the launch's vector and matrix "
                "instructions takes; the two kinds do not overlap on a SIMD "
                "(profiles/round2/ubench_mfma_valu_mix.txt)",
        "source": os.path.relpath(path, ROOT),
    }


def valu_roofline(achieved, traffic, traffic_source, kernel_name, kernel_ms, kernel_ms_source, st):
    """Roofline of a launch whose filter ran on the vector ALU, against the FP32 VALU peak."""
    return {
        "bound": "valu_fp32",
        "achieved": round(achieved, 3),
        "peak": PEAK_FP32_VALU_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "kernel": kernel_name,
        "kernel_ms": round(kernel_ms, 3),
        "kernel_ms_source": kernel_ms_source,
        "rays_per_launch": int(st.rays),
        "rays_per_sample": round(st.rays / max(st.samples, 1), 4),
        "flop_per_ray_sphere_test": FLOP_PER_TEST,
        "frac_executed": round(achieved / PEAK_FP32_VALU_TFLOPS
                               * FLOP_PER_TEST_EXECUTED / FLOP_PER_TEST, 4),
        "frac_executed_note": "the same launch priced at the 14 flop (7 FMA, plus a "
                              "compare that is not counted) per test the scan actually "
                              "issues instead of the reference's 17",
        "rank": 0,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames enqueued on alternating streams so that one frame's drain overlaps "
                         "the next frame's start (default: 1 on one GPU, 2 on several)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on GPU 0, gloo gather through host memory (plumbing check)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    import torch
    import torch.distributed as dist

    from ray_tracing_amd import hip, host, multi_gpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    # --rehearse-on-one-gpu: every rank renders on GPU 0 and the gather goes through gloo and
    # host memory.  It exercises this file's N > 1 code on a one-GPU box; its numbers mean nothing.
    rehearsal = args.rehearse_on_one_gpu
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    hip.init(device_index)

    # host side: the reference's scene, camera and request, from the product's host mirror
    W, H, spp, depth = cfg["width"], cfg["height"], cfg["spp"], cfg["depth"]
    scene_host = host.create_random_world(1, cfg["grid"], cfg.get("diffuse_below", 0.8),
                                          cfg.get("metal_below", 0.95), cfg.get("glass", False))
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, W / H)
    scene = hip.Scene(scene_host.spheres, scene_host.materials)   # resident in HBM from here on
    frame = hip.make_frame(camera.m, W, H, spp, depth, args.seed, flags=0)

    # Frames in flight.  One frame at a time is the plain reading of "a step" and what `value`
    # reports.  With the frame split over N GPUs a rank's share is short enough that the time
    # its last deep paths take to drain shows; a host that renders a sequence of frames can put
    # consecutive frames on alternating streams (each with its own output buffers) so that the
    # next frame's workgroups take over the compute units as the previous frame's retire.  That
    # is pipelined throughput, not a frame's latency, and is reported beside `value` as
    # `pipelined` (N > 1 only, or when --frames-in-flight asks for it).
    plan = multi_gpu.RowPlan(W, H, world)

    def timed_run(n_flight):
        streams = ([torch.cuda.current_stream()] if n_flight == 1
                   else [torch.cuda.Stream(device=device) for _ in range(n_flight)])
        shardeds = [multi_gpu.ShardedFrame(plan, device, rank, stage_through_host=rehearsal)
                    for _ in range(n_flight)] if world > 1 else None
        fulls = [torch.zeros((H, W, 3), dtype=torch.float32, device=device)
                 for _ in range(n_flight)] if world == 1 else None
        step_no = [0]

        def step():
            i = step_no[0] % n_flight
            step_no[0] += 1
            with torch.cuda.stream(streams[i]):
                raw = streams[i].cuda_stream
                if world == 1:
                    hip.render_rows_device_async(scene, frame, 0, H, True, fulls[i].data_ptr(), raw)
                    return fulls[i]

                def render_fn(tiles, offs, strides, out):
                    hip.render_device_async(scene, frame, tiles, offs, strides, True, out.data_ptr(), raw)

                return shardeds[i].render_and_gather(render_fn)

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fb = step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, fb

    n_flight = args.frames_in_flight or 1
    elapsed, fb = timed_run(n_flight)
    launch_ms = hip.get_launch_ms(min(args.steps, 64))
    st = hip.get_stats()                       # this rank's last launch
    if n_flight == 1:
        kernel_ms = float(launch_ms.mean()) if len(launch_ms) else float("nan")
        kernel_ms_source = "HIP events around each launch on its stream"
    else:
        # overlapping launches: an event pair also spans the wait for the other frame, so the
        # per-launch time is the timed region divided by its launches (includes the gather)
        kernel_ms = elapsed / args.steps * 1e3
        kernel_ms_source = "timed region / launches (frames overlap, so event pairs do not isolate a launch)"
    pipelined = None
    if world > 1 and not args.frames_in_flight:
        p_elapsed, _ = timed_run(2)
        pipelined = {
            "frames_in_flight": 2,
            "value": round(W * H * spp * args.steps / p_elapsed * 1e-6, 2),
            "unit": "Msamples/s",
            "ms_per_step": round(p_elapsed / args.steps * 1e3, 3),
            "note": "throughput with consecutive frames overlapped on two streams; ms_per_step "
                    "here is the timed region divided by its frames, not one frame's latency",
        }
    n_spheres = len(scene_host.spheres)
    flop = FLOP_PER_TEST * n_spheres * st.rays
    achieved = flop / (kernel_ms * 1e-3) * 1e-12 if kernel_ms > 0 else 0.0
    matrix = st.threads_per_workgroup == 1024      # the launch ran the matrix filter
    achieved_mfma = achieved * FLOP_PER_TEST_MFMA / FLOP_PER_TEST

    if matrix:
        kernel_name = "rt::trace_kernel<false,%s,false,true>" % ("true" if n_spheres > 512 else "false")
    else:
        kernel_name = "rt::trace_kernel<false,%s,false,false>" % ("true" if n_spheres > 1136 else "false")
    if rank == 0:
        traffic, traffic_source = (profiled_traffic(args.config, kernel_name) if world == 1
                                   else (None, "not reported for N > 1"))
        samples_per_step = W * H * spp
        value = samples_per_step * args.steps / elapsed * 1e-6
        line = {
            "metric": "Msamples/sec at 1080p RTIOW final scene" if args.config == DEFAULT_CONFIG
                      else "Msamples/sec, " + args.config,
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL on one GPU, not a result",
            "config": {
                "workload": args.config,
                "spheres": n_spheres,
                "image": "%dx%d" % (W, H),
                "samples_per_pixel": spp,
                "max_depth": depth,
                "frames_in_flight": n_flight,   # of `value`; see `pipelined` for two
                "sharding": "rows in blocks of %d dealt round-robin over %d rank(s)%s"
                            % (plan.block_rows, world, ", RCCL gather to rank 0" if world > 1 else ""),
            },
            "roofline": valu_roofline(achieved, traffic, traffic_source, kernel_name, kernel_ms,
                                      kernel_ms_source, st) if not matrix else {
                "bound": "mfma",
                "achieved": round(achieved_mfma, 3),
                "peak": PEAK_F16_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_mfma / PEAK_F16_MFMA_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kernel_name,
                "kernel_ms": round(kernel_ms, 3),
                "kernel_ms_source": kernel_ms_source,
                "rays_per_launch": int(st.rays),
                "rays_per_sample": round(st.rays / max(st.samples, 1), 4),
                "flop_per_ray_sphere_test": FLOP_PER_TEST_MFMA,
                "flop_note": "one (sphere, ray) entry of v_mfma_f32_16x16x32_f16: 32 slots (ten "
                             "terms of the discriminant's bilinear form, three split-f16 products "
                             "each, and the threshold) x 2 flop; frac is also the share of time "
                             "the matrix pipe is busy at the nominal clock",
                "issue_bound": profiled_issue_bound(args.config, kernel_name, kernel_ms) if world == 1 else None,
                "fp32_form": {
                    "flop_per_ray_sphere_test": FLOP_PER_TEST,
                    "achieved": round(achieved, 3),
                    "peak": PEAK_FP32_VALU_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                    "note": "the same launch priced as the reference states the test (17 flop) "
                            "against the FP32 vector-ALU peak; it can pass 1 because the filter "
                            "no longer runs on the vector ALU (see valu_scan for the kernel that does)",
                },
                "rank": 0,
            },
        }
        if pipelined is not None:
            line["pipelined"] = pipelined
        if world == 1:
            line["drop_in"] = drop_in_leg(hip, scene_host, camera, W, H, spp, depth, args.seed,
                                          args.steps, kernel_ms, torch.cuda.device_count())
        if world == 1 and matrix:
            # The same frame with the filter on the vector ALU (the kernel the FP32 VALU roofline
            # describes), timed by HIP events around each launch.
            from ray_tracing_amd import abi
            vframe = hip.make_frame(camera.m, W, H, spp, depth, args.seed, flags=abi.RT_FLAG_SCAN_VALU)
            vout = torch.zeros_like(fb)
            raw = torch.cuda.current_stream().cuda_stream
            hip.render_rows_device_async(scene, vframe, 0, H, True, vout.data_ptr(), raw)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(args.steps):
                hip.render_rows_device_async(scene, vframe, 0, H, True, vout.data_ptr(), raw)
            torch.cuda.synchronize()
            tv = (time.perf_counter() - tv) / args.steps
            v_ms = hip.get_launch_ms(min(args.steps, 64))
            v_st = hip.get_stats()
            v_kernel_ms = float(v_ms.mean()) if len(v_ms) else float("nan")
            v_name = "rt::trace_kernel<false,%s,false,false>" % ("true" if n_spheres > 1136 else "false")
            v_traffic, v_source = profiled_traffic(args.config, v_name)
            v_achieved = FLOP_PER_TEST * n_spheres * v_st.rays / (v_kernel_ms * 1e-3) * 1e-12
            line["valu_scan"] = {
                "value": round(samples_per_step / tv * 1e-6, 2),
                "unit": "Msamples/s",
                "ms_per_step": round(tv * 1e3, 3),
                "same_pixels_as_headline": bool(torch.equal(vout.view(torch.int32), fb.view(torch.int32))),
                "roofline": valu_roofline(v_achieved, v_traffic, v_source, v_name, v_kernel_ms,
                                          "HIP events around each launch on its stream", v_st),
                "note": "RT_FLAG_SCAN_VALU: the ray/sphere filter as seven fused f32 instructions and "
                        "a compare per test on the vector ALU, everything else the same kernel code",
            }
        if world == 1:
            # The opt-in mode that walks a bounding-volume hierarchy instead of testing every
            # sphere (RT_FLAG_ACCEL): same pixels, bit for bit, but not the brute-force work the
            # roofline above is counted in, so it is reported beside the headline, not as it.
            from ray_tracing_amd import abi
            aframe = hip.make_frame(camera.m, W, H, spp, depth, args.seed, flags=abi.RT_FLAG_ACCEL)
            aout = torch.zeros_like(fb)
            raw = torch.cuda.current_stream().cuda_stream
            hip.render_rows_device_async(scene, aframe, 0, H, True, aout.data_ptr(), raw)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(args.steps):
                hip.render_rows_device_async(scene, aframe, 0, H, True, aout.data_ptr(), raw)
            torch.cuda.synchronize()
            ta = (time.perf_counter() - ta) / args.steps
            line["accel"] = {
                "value": round(samples_per_step / ta * 1e-6, 2),
                "unit": "Msamples/s",
                "ms_per_step": round(ta * 1e3, 3),
                "same_pixels_as_headline": bool(torch.equal(aout.view(torch.int32),
                                                            fb.view(torch.int32))),
                "note": "RT_FLAG_ACCEL (opt-in BVH traversal); not the brute-force path the "
                        "metric and roofline are defined on",
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.seed)
        else:
            line["cpu_baseline"] = None
        if fb is not None:
            import numpy as np
            img = fb.detach().cpu().numpy()
            line["config"]["mean_rgb"] = [round(float(v), 5) for v in np.nanmean(img, axis=(0, 1))]
        print(json.dumps(line), flush=True)

    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
