"""Row sharding of one frame over the GPUs of a node, one process per GPU, and the gather of
the framebuffer to rank 0 with torch.distributed (backend "nccl" = RCCL over xGMI).

The reference spreads 64x64 tiles over worker threads that share one result vector
(src/cpu/multi.rs:30-62, 140-141) and then scatters the tiles into the framebuffer
(src/cpu/mod.rs:60-80).  Here the unit is a block of BLOCK_ROWS full-width rows, dealt
round-robin to the ranks (sky rows are cheap and ground rows expensive, so contiguous bands
would be unbalanced); each rank renders its blocks into one compact device buffer, one RCCL
gather brings the buffers to rank 0, and one index_select puts the rows in frame order.  The
path has no other exchange step.  Pixels do not depend on which rank renders them (the random
stream is keyed by the global pixel index), so the gathered frame is bit-identical to a
single-GPU render.

`render_fn` is injectable so that the sharding, gather and reassembly can be exercised on the
CPU with the gloo backend.
"""
import numpy as np

from . import abi

BLOCK_ROWS = 8


class RowPlan:
    """Which rows each rank renders and where they sit in its compact buffer."""

    def __init__(self, image_width, image_height, world_size, block_rows=BLOCK_ROWS):
        self.width = int(image_width)
        self.height = int(image_height)
        self.world_size = int(world_size)
        self.block_rows = int(block_rows)
        self.n_blocks = (self.height + self.block_rows - 1) // self.block_rows
        self.blocks_per_rank = (self.n_blocks + self.world_size - 1) // self.world_size
        self.rows_per_rank = self.blocks_per_rank * self.block_rows   # padded buffer height

    def tiles_for(self, rank):
        """(tiles, out_offset, out_stride) of rank `rank`, for rt_render_device_async."""
        tiles, offs = [], []
        for local, b in enumerate(range(rank, self.n_blocks, self.world_size)):
            y0 = b * self.block_rows
            rows = min(self.block_rows, self.height - y0)
            tiles.append((0, y0, self.width, rows))
            offs.append(local * self.block_rows * self.width)
        t = np.array(tiles, dtype=abi.TILE_DTYPE) if tiles else np.zeros(0, dtype=abi.TILE_DTYPE)
        return (t, np.array(offs, dtype=np.uint64), np.full(len(tiles), self.width, dtype=np.uint32))

    def rows_of(self, rank):
        return int(sum(min(self.block_rows, self.height - b * self.block_rows)
                       for b in range(rank, self.n_blocks, self.world_size)))

    def frame_row_map(self):
        """For frame row y, its row index in the concatenation of the ranks' buffers."""
        idx = np.empty(self.height, dtype=np.int64)
        for y in range(self.height):
            b = y // self.block_rows
            rank, local = b % self.world_size, b // self.world_size
            idx[y] = rank * self.rows_per_rank + local * self.block_rows + (y - b * self.block_rows)
        return idx


class ShardedFrame:
    """Buffers and index map of one sharded frame; reusable across renders."""

    def __init__(self, plan, device, rank, stage_through_host=False):
        """`stage_through_host` gathers host copies (for a backend without device collectives,
        such as gloo driving GPU renders in a rehearsal); RCCL gathers device buffers directly."""
        import torch
        self.plan = plan
        self.rank = rank
        self.stage_through_host = stage_through_host
        self.local = torch.zeros((plan.rows_per_rank, plan.width, 3), dtype=torch.float32,
                                 device=device)
        self.tiles, self.offsets, self.strides = plan.tiles_for(rank)
        if rank == 0:
            # the gather writes straight into the slices of one buffer, which is already the
            # concatenation of the ranks' compact buffers that row_map indexes
            self.stacked = torch.zeros((plan.world_size,) + tuple(self.local.shape),
                                       dtype=torch.float32,
                                       device="cpu" if stage_through_host else device)
            self.gathered = list(self.stacked.unbind(0))
            self.row_map = torch.from_numpy(plan.frame_row_map()).to(device)
        else:
            self.stacked = None
            self.gathered = None
            self.row_map = None

    def render_and_gather(self, render_fn):
        """render_fn(tiles, out_offset, out_stride, local_tensor) enqueues this rank's blocks.
        Returns the (H, W, 3) frame on rank 0, None elsewhere.  Asynchronous with respect to
        the host on a GPU (the caller synchronises)."""
        import torch
        import torch.distributed as dist
        if len(self.tiles):
            render_fn(self.tiles, self.offsets, self.strides, self.local)
        if self.plan.world_size == 1:
            stacked = self.local
        elif self.stage_through_host:
            dist.gather(self.local.cpu(), self.gathered, dst=0)   # .cpu() waits for the render
            if self.rank != 0:
                return None
            stacked = self.stacked.to(self.local.device)
        else:
            dist.gather(self.local, self.gathered, dst=0)
            if self.rank != 0:
                return None
            stacked = self.stacked
        flat = stacked.reshape(-1, self.plan.width, 3)
        return flat.index_select(0, self.row_map)


def dispatch_work_gpu_sharded(dispatch, seed=1, flags=0):
    """cpu::dispatch_work_* contract over all ranks of the current process group: every rank
    calls this with the same Dispatch; rank 0 returns the resolved (H, W, 3) framebuffer as a
    numpy array, the other ranks None."""
    import torch
    import torch.distributed as dist

    from . import hip

    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cuda", torch.cuda.current_device())
    plan = RowPlan(dispatch.image_width, dispatch.image_height, world)
    frame = hip.make_frame(dispatch.camera.m, dispatch.image_width, dispatch.image_height,
                           dispatch.samples_per_pixel, dispatch.max_depth, seed, flags)
    scene = hip.Scene(dispatch.world.spheres, dispatch.world.materials)
    sharded = ShardedFrame(plan, device, rank)
    stream = torch.cuda.current_stream().cuda_stream

    def render_fn(tiles, offs, strides, out):
        hip.render_device_async(scene, frame, tiles, offs, strides, True, out.data_ptr(), stream)

    fb = sharded.render_and_gather(render_fn)
    torch.cuda.synchronize()
    scene.close()
    return fb.cpu().numpy() if fb is not None else None


def sample_ranges(samples_per_pixel, world_size):
    """[begin, end) of the samples each rank renders when a frame's SAMPLES are split."""
    spp, n = int(samples_per_pixel), int(world_size)
    return [((r * spp) // n, ((r + 1) * spp) // n) for r in range(n)]


def dispatch_work_gpu_sample_split(dispatch, seed=1, flags=0, accumulate_fn=None, device=None):
    """Alternative sharding for images too small to split by rows: every rank renders the WHOLE
    frame for its share of the samples (rt_accumulate_device_async on a zeroed buffer), the
    un-normalised sums are gathered to rank 0 and added there in rank order, then resolved.
    The exchange moves world_size full frames, so row sharding is the default; this one exists
    because sums are additive (src/cpu/job.rs:48-61).  The result depends on the number of ranks
    only through the association of the per-rank sums.

    `accumulate_fn(sample_begin, n_samples, sums_tensor)` is injectable for CPU tests."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    d = dispatch
    W, H = d.image_width, d.image_height
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    begin, end = sample_ranges(d.samples_per_pixel, world)[rank]
    sums = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
    scene = None
    if accumulate_fn is None:
        from . import hip
        scene = hip.Scene(d.world.spheres, d.world.materials)
        stream = torch.cuda.current_stream().cuda_stream
        tile = np.array([(0, 0, W, H)], dtype=abi.TILE_DTYPE)
        off, stride = np.zeros(1, dtype=np.uint64), np.full(1, W, dtype=np.uint32)

        def accumulate_fn(b, n, t):
            frame = hip.make_frame(d.camera.m, W, H, n, d.max_depth, seed, flags)
            hip.accumulate_device_async(scene, frame, b, tile, off, stride, t.data_ptr(), stream)

    if end > begin:
        accumulate_fn(begin, end - begin, sums)
    gathered = [torch.zeros_like(sums) for _ in range(world)] if rank == 0 else None
    dist.gather(sums, gathered, dst=0)
    out = None
    if rank == 0:
        total = gathered[0]
        for r in range(1, world):
            total = total + gathered[r]
        x = torch.sqrt(total * (1.0 / float(d.samples_per_pixel)))
        # Rust's f32::clamp: NaN stays NaN (src/cpu/mod.rs:73-75)
        x = torch.where(x < 0.0, torch.zeros_like(x), torch.where(x > 0.999, torch.full_like(x, 0.999), x))
        out = x.cpu().numpy()
    if device.type == "cuda":
        torch.cuda.synchronize()
    if scene is not None:
        scene.close()
    return out
