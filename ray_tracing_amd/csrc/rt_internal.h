// Internal declarations shared by the kernel file and the host side of librt_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/rt_abi.h"

namespace rt {

constexpr int kWave = 64;
#ifndef RT_THREADS
#define RT_THREADS 512
#endif
constexpr int kThreads = RT_THREADS;    // 8 waves per workgroup: chunked scan, hierarchy, hooks
#ifndef RT_THREADS_SINGLE
#define RT_THREADS_SINGLE 256
#endif
// Workgroup of the scan over an LDS-resident sphere list.  A workgroup gives its registers and
// LDS back only when its last wave ends, so four waves instead of eight let the next launch on
// another stream move in sooner while this one drains (same speed for a launch on its own).
constexpr int kThreadsSingle = RT_THREADS_SINGLE;
#ifndef RT_THREADS_ACCEL
#define RT_THREADS_ACCEL 512
#endif
constexpr int kThreadsAccel = RT_THREADS_ACCEL;   // workgroup of the hierarchy traversal
// The scan on the matrix pipe: one workgroup of 16 waves per CU shares the sphere rows (64 bytes a
// sphere), which two smaller workgroups could not both hold beside the waves' candidate lists.
#ifndef RT_THREADS_MFMA
#define RT_THREADS_MFMA 1024
#endif
constexpr int kThreadsMfma = RT_THREADS_MFMA;
constexpr uint32_t kMfmaChunk = 512;               // spheres resident in LDS at a time
constexpr uint32_t kMfmaBlock = 16;                // spheres per MFMA (rows of a 16x16x32 product)
constexpr uint32_t kMfmaRows = kMfmaChunk + kMfmaBlock;   // + one block of read-ahead padding
constexpr uint32_t kMfmaMaxAlways = 8;             // spheres out of f16 range, listed for every ray
constexpr int trace_threads(bool multi, bool accel, bool mfma = false) {
  return accel ? kThreadsAccel : mfma ? kThreadsMfma : multi ? kThreads : kThreadsSingle;
}
#ifndef RT_RAYS_PER_LANE
#define RT_RAYS_PER_LANE 2
#endif
constexpr int kRaysPerLane = RT_RAYS_PER_LANE;   // independent paths blocked into one lane
constexpr uint32_t kListEntries = 512;  // candidates a wave can list between two replays
constexpr int kScanUnroll = 4;          // spheres per unrolled block of the scan
constexpr uint32_t kChunkSpheres = 1136; // spheres resident in LDS at a time: with the read-ahead
                                         // padding and the waves' candidate lists, two workgroups
                                         // of kThreads fill a CU's LDS
constexpr size_t kLdsPerWorkgroup = 160 * 1024 / 2;   // two workgroups share a CU's 160 KB
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kItemsPerGrab = 64;  // pixels a wave reserves at a time when a pixel is one item
#ifndef RT_GRAB_ITEMS
#define RT_GRAB_ITEMS 64
#endif
constexpr uint32_t kGrabItems = RT_GRAB_ITEMS;   // sliced samples: a wave reserves as many whole
constexpr uint32_t kGrabPixelsMax = 4;  // pixels (1 to kGrabPixelsMax) as make up to kGrabItems items
#ifndef RT_CHUNK_GRABS
#define RT_CHUNK_GRABS 8
#endif
#ifndef RT_GUIDE_FACTOR
#define RT_GUIDE_FACTOR 16
#endif
constexpr uint32_t kChunkGrabs = RT_CHUNK_GRABS; // a workgroup's chunk of the frame: this many
                                                 // reservations for each of its waves, at most
// Sliced samples: every slice of a pixel is run by lanes of ONE wave (a wave reserves whole
// pixels), each slice's sum is parked in the wave's own ring in device memory, and once a pixel's
// last slice is in, the wave adds the parked sums in slice order and writes the pixel.  A wave
// holds at most kRingSlots pixels open; per wave in LDS: where each open pixel goes in the output
// and how many of its slices are in (one byte each).
constexpr uint32_t kRingSlots = 32;
constexpr size_t kSlotTableBytes = kRingSlots * sizeof(uint32_t) + kRingSlots;
#ifndef RT_MERGE_BATCH
#define RT_MERGE_BATCH 8
#endif
#ifndef RT_MERGE_EVERY
#define RT_MERGE_EVERY 4
#endif
constexpr uint32_t kMergeBatch = RT_MERGE_BATCH;   // pixels that wait for each other before a wave merges them
constexpr uint32_t kMergeEvery = RT_MERGE_EVERY;   // a wave looks for ready pixels every so many steps (a power of two)
constexpr uint32_t kTargetItems = 1u << 20; // below this many pixels, split samples over lanes
constexpr uint32_t kSamplesPerItem = 8;     // samples one full-size work item runs
#ifndef RT_MAX_SLICES
#define RT_MAX_SLICES 64
#endif
constexpr uint32_t kMaxSlices = RT_MAX_SLICES;  // planes of the partial-sum buffer, at most
constexpr uint32_t kLaneSlots = 256u * 2u * kThreads * kRaysPerLane;   // paths in flight per GPU
#ifndef RT_TAIL_PATHS
#define RT_TAIL_PATHS 16
#endif
#ifndef RT_TAIL_PATHS_MFMA
#define RT_TAIL_PATHS_MFMA 4
#endif
constexpr uint32_t kTailPathsMfma = RT_TAIL_PATHS_MFMA;   // the same for the matrix scan, whose
                                                          // step costs the same for 128 paths as for 5
constexpr uint32_t kTailPaths = RT_TAIL_PATHS;  // a wave with no work left and at most this many
                                                // live paths traces them with the spheres spread
                                                // across its lanes

// Division of a 32-bit number by a divisor fixed on the host (Granlund and Montgomery's
// round-up method): q = (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(n, mul), exact for every
// 32-bit n and every divisor from 1 to 2^32 - 1.  Five instructions where the general division
// the compiler emits takes about thirty.
struct Divider {
  uint32_t mul = 1;
  uint16_t sh1 = 0, sh2 = 0;
};
inline Divider make_divider(uint32_t d) {
  Divider v;
  if (d == 0) d = 1;
  uint32_t l = 0;                                   // ceil(log2 d)
  while (l < 32 && (1ull << l) < d) ++l;
  v.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  v.sh1 = l < 1 ? (uint16_t)l : 1;
  v.sh2 = l < 1 ? 0 : (uint16_t)(l - 1);
  return v;
}
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ uint32_t fast_div(uint32_t n, const Divider& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t t = __umulhi(n, v.mul);
#else
  const uint32_t t = (uint32_t)(((uint64_t)n * v.mul) >> 32);
#endif
  return (t + ((n - t) >> v.sh1)) >> v.sh2;
}
// a / b, correctly rounded, for a divisor whose correctly rounded reciprocal y = 1.0f / b is at
// hand: five instructions of the fast kind in place of the division's expansion (two scalings,
// the hardware's reciprocal, six more).  q = a * y is within 1.5 ulp of a / b; one step
// q += (a - b q) y brings it within an ulp (the residual a - b q then is exact), and the second
// such step rounds to the nearest (Markstein's theorem).  Holds where nothing overflows or
// underflows, which a, b of the sample loop (0 or 2^-24 <= a <= 2^31, 1 <= b <= 2^31) cannot.
__host__ __device__ __forceinline__ float div_by(float a, float b, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}
#endif

// One tile of the request as the kernel sees it.
struct TileDev {
  uint32_t x, y, w, h;
  uint32_t first_pixel;   // exclusive prefix sum of w*h over the tile list
  uint32_t out_stride;    // pixels between rows of this tile in the output
  uint64_t out_offset;    // pixel index of the tile's first pixel in the output
  Divider wdiv;           // division by w (make_divider(w)); set_tile_width fills both
};
inline void set_tile_width(TileDev& t, uint32_t w) {
  t.w = w;
  t.wdiv = make_divider(w);
}

// Device copy of a World (reference src/main.rs:168-172), split hot/cold:
//   trig[i] = (cx, cy, cz, -k'/2)               read by every ray for every sphere; k' is
//             |c|^2 - r^2 minus a rounding slack (see trigger_constant in rt_api.cpp)
//   geom[i] = (cx, cy, cz, radius*radius)       read when a sphere is replayed exactly
//   surf[2i] = (r, g, b, flags bits), surf[2i+1] = (metallic_strength, ir, radius, material_id
//             bits): the sphere's radius and a copy of its material, read once per hit; for a
//             dielectric, whose colour nothing reads, (1/ir, r0 of 1/ir, r0 of ir) instead of
//             (r, g, b), r0 being reflectance's ((1 - x) / (1 + x))^2; for a metal (r*r, g*g, b*b)
//             and 1 - metallic_strength instead of (r, g, b) and metallic_strength
// Optional bounding-volume hierarchy (RT_FLAG_ACCEL; built by build_bvh in rt_bvh.cpp).
constexpr uint32_t kBvhLeaf = 0x80000000u;
struct BvhNode {
  float lo[3];
  uint32_t a;    // interior: left child; leaf: first slot in leaf_geom / leaf_index
  float hi[3];
  uint32_t b;    // interior: right child; leaf: kBvhLeaf | number of spheres
};
struct BvhInfo {
  uint32_t n_big = 0;      // leaf slots [0, n_big) hold the spheres kept out of the tree
  uint32_t n_nodes = 0;    // 0: the tree is empty (every sphere is "big", or there is none)
  float centre[3] = {0, 0, 0};   // a ball (centre, extent) holds every tree sphere's centre
  float extent = 0;
  float r_min = 0, r_max = 0;    // smallest / largest radius among the tree's spheres
};
struct BvhHost {
  std::vector<BvhNode> nodes;
  std::vector<float4> leaf_geom;     // (cx, cy, cz, r*r), big spheres first, then tree order
  std::vector<uint32_t> leaf_index;  // original sphere number of each slot
  uint32_t depth = 0;                // levels of the tree (the traversal stack must hold them)
  BvhInfo info;
};
void build_bvh(const rt_sphere* spheres, uint32_t n, BvhHost* out);

struct SceneDev {
  const float4* trig;
  const float4* geom;
  const float4* surf;
  uint32_t n_spheres;
  uint32_t n_materials;
  // RT_FLAG_ACCEL
  const BvhNode* bvh;
  const float4* leaf_geom;
  const uint32_t* leaf_index;
  BvhInfo bvh_info;
  // The scan on the matrix pipe (see build_mfma_scene in rt_api.cpp).  Spheres in `mfma` order:
  // those inside the f16 range sorted along a space-filling curve, then the n_always others.
  const uint4* mfma_rows;       // [chunk][4][kMfmaRows] x 8 f16: the spheres' rows, by quarters
  const float4* mfma_geom;      // [position] (cx, cy, cz, r*r) for the exact replay
  const uint32_t* mfma_index;   // [position] -> sphere number in the World
  uint32_t mfma_ok;             // 0: this scene keeps to the vector-ALU scan
  uint32_t mfma_always;         // the last mfma_always positions are listed for every ray
  float mfma_scale;             // power of two applied to centres, radii and ray origins
};

struct Counters {
  uint32_t next_pixel;  // work-distribution cursor: the first pixel no wave has reserved yet
  uint32_t pad;
  unsigned long long rays;     // World::hit calls
  unsigned long long samples;  // shoot_ray calls from the pixel loop
  // Wave-cycles per phase and drain statistics (see rt_get_phase_cycles); filled only by a
  // build with -DRT_DIAG_PHASES, whose timings are for shares, not for speed.
  unsigned long long phase[24];   // 16-23: see RT_DIAG_TRIGGERS, RT_DIAG_RING, RT_DIAG_DRAIN in rt_kernels.hip
};

struct TraceParams {
  SceneDev scene;
  rt_camera cam;
  float origin[3];        // view_inv * (0,0,0,1), a per-frame constant (src/main.rs:71-72)
  // Terms of Camera::get_ray's two Mat4 * Vec4 products that do not depend on the sample
  // (src/main.rs:74-78: the point's z = 0 and w = 1, the direction's w = 0), formed on the host:
  // proj_inv's column 2 times 0.0, its column 3 times 1.0, view_inv's column 3 times 0.0
  float proj_z0[3], proj_w1[3], view_w0[3];
  uint32_t width, height;
  // `width as f32`, `height as f32` and their reciprocals 1.0f / x (correctly rounded, formed on
  // the host), for the divisions of src/cpu/job.rs:53-54: see div_by in rt_kernels.hip
  float fwidth, fheight, inv_width, inv_height;
  uint32_t max_depth, spp;
  uint32_t seed_a;          // rt_rng_seed_a(seed)
  uint32_t slices;          // work items per pixel; item = pixel * slices + slice
  Divider slices_div;       // division by slices
  uint32_t grab_pixels;     // pixels a wave reserves at a time from its workgroup's chunk
  uint32_t chunk_max;       // pixels a workgroup reserves at a time from the frame, at most ...
  uint32_t chunk_min;       // ... and at least; in between, what is left >> guide_shift
  uint32_t guide_shift;
  uint32_t ring_slots;      // slices > 1: pixels a wave can hold open (<= kRingSlots)
  uint32_t slot_table_offset;   // RT_FLAG_ACCEL: where the waves' slot tables start in LDS
  float* ring;              // slices > 1: [wave][ring_slots][slices][3] parked slice sums
  uint32_t slice_begin[kMaxSlices + 1];   // slice k runs samples [slice_begin[k], slice_begin[k+1])
  uint32_t n_tiles;
  uint32_t total_pixels;
  uint32_t total_items;     // total_pixels * slices
  uint64_t out_pixels;      // pixels the output spans (its last covered pixel + 1)
  TileDev tile0;            // the tile list when n_tiles == 1
  const TileDev* tiles;     // otherwise
  // A list of equal row blocks at a constant pitch (one rank's share of a frame dealt out in
  // blocks of rows) is passed as ONE tile, tile0 = the first block with all the list's rows:
  // row ty of the list is row ty % row_block of block ty / row_block, which starts row_period
  // image rows and out_period output pixels after the block before it.  0: an ordinary tile.
  uint32_t row_block, row_period;
  uint64_t out_period;
  Divider row_block_div;
  float* out;               // the call's output
  int resolve;              // apply x*(1/spp), sqrt, clamp on the way out
  uint32_t bvh_stack;       // RT_FLAG_ACCEL: rows of the traversal stack (levels of the tree)
  int bvh_in_lds;           // RT_FLAG_ACCEL: the workgroup keeps a copy of the hierarchy in LDS
  int accumulate;           // progressive rendering: a pixel's sum is added to what `out` holds
                            // (one item per pixel: its running sum starts from that value)
  float inv_spp;
  Counters* counters;
};

// zero_kernel: the pixels a call's tiles cover, set to (0, 0, 0) (a frame at max_depth 0).
struct ZeroParams {
  float* out;             // [out pixels][3]
  uint32_t n_tiles;
  uint32_t total_pixels;  // pixels the tiles cover; only these are written
  TileDev tile0;
  const TileDev* tiles;
};

// compose_final_framebuffer's arithmetic over a buffer of sums (progressive rendering).
struct ResolveParams {
  const float* sums;      // [n_pixels][3]
  float* out;             // [n_pixels][3]; may alias sums
  uint64_t n_pixels;
  float inv_spp;
};

struct HookParams {
  SceneDev scene;
  const rt_ray_in* rays;
  rt_ray_out* out;
  uint32_t n_rays;
  uint32_t seed;
  int accel;              // World::hit through the hierarchy instead of the scan
  int mfma;               // the scan's filter on the matrix pipe
};

// Launchers implemented in rt_kernels.hip.  `strict` selects the non-contracted discriminant.
size_t trace_lds_bytes(uint32_t n_spheres, bool accel, uint32_t bvh_nodes, uint32_t bvh_depth,
                       bool* bvh_in_lds, uint32_t* slot_table_offset = nullptr);
uint32_t bvh_stack_depth();
size_t trace_lds_bytes_mfma(int threads);
hipError_t trace_max_blocks_per_cu(bool strict, bool multi, bool accel, size_t lds_bytes, int* out);
hipError_t trace_max_blocks_per_cu_mfma(bool multi, size_t lds_bytes, int* out);
hipError_t launch_trace(const TraceParams& p, bool strict, bool accel, bool mfma, int blocks,
                        hipStream_t stream);
hipError_t launch_zero(const ZeroParams& p, hipStream_t stream);
hipError_t launch_clear_counters(Counters* c, hipStream_t stream);
hipError_t launch_resolve(const ResolveParams& p, hipStream_t stream);
hipError_t launch_hook(const HookParams& p, bool strict, hipStream_t stream);
hipError_t launch_sqrt_check(unsigned long long* out4, hipStream_t stream);   // out4 zeroed by the caller

}  // namespace rt
