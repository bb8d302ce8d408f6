"""ray_tracing_amd - MI355X (gfx950) path tracer behind the tile-renderer interface of
Nanoteck137/ray_tracing.

  abi        ctypes / numpy mirrors of include/rt_abi.h
  hip        binding of librt_hip.so (HIP kernels + C ABI); the only thing that renders
  host       the reference's host-side names (Camera, World, create_job_queue, Dispatch,
             dispatch_work_gpu, write_framebuffer_to_image) over the C++ mirror librt_host.so
  multi_gpu  row sharding over ranks and the RCCL gather of the framebuffer
"""
from . import abi  # noqa: F401

__all__ = ["abi", "hip", "host", "multi_gpu"]
