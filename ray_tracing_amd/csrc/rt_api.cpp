// rt_api.cpp - host side of librt_hip.so: the C ABI declared in include/rt_abi.h.
//
// Everything here is plumbing around the kernels of rt_kernels.hip: argument checks, the
// device copy of the World, the tile table, launch geometry, and the two host-pointer entry
// points that stand in for the reference's dispatch_work_cpu_* (src/cpu/multi.rs:131-172,
// src/cpu/single.rs:10-50).  There is no CPU rendering path in this library: without a GPU
// every compute entry point fails with RT_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_rng.h"
#include "rt_internal.h"

struct rt_scene {
  // `dev` and the hierarchy's buffers change when the hierarchy is built on first use, which
  // happens behind a `const rt_scene*` (the scene the caller sees does not change)
  mutable rt::SceneDev dev;
  void* trig = nullptr;
  void* geom = nullptr;
  void* surf = nullptr;
  mutable void* bvh = nullptr;  // RT_FLAG_ACCEL: nodes, permuted geometry, original indices
  mutable void* leaf_geom = nullptr;
  mutable void* leaf_index = nullptr;
  mutable uint32_t bvh_depth = 0;
  mutable bool bvh_built = false;
  void* mfma_rows = nullptr;    // the matrix scan's copy of the scene (build_mfma_scene)
  void* mfma_geom = nullptr;
  void* mfma_index = nullptr;
  int device = -1;
  // host copies: what the hierarchy is built from, and what rt_render compares a request's
  // world with to decide whether the device copy it kept from the last call still serves
  std::vector<rt_sphere> host_spheres;
  std::vector<rt_material> host_materials;
};

namespace {

// ---- memory for rt_render_rows_device_noalloc ------------------------------------------------
// Work counters and partial-sum planes of the entry point that may not allocate.  Every GPU the
// library runs on holds its own copy of this array, at its own address, from the moment the
// code object is loaded until the process ends.  In host code the NAME below is only a handle
// for hipGetSymbolAddress (init_context asks once per device and keeps the answer in
// State::scratch); it is never passed to a kernel, a memset or a copy.
constexpr size_t kScratchCounterBytes = 256;                  // rt::Counters, padded
constexpr size_t kScratchPlaneBytes = RT_NOALLOC_PLANE_BYTES; // partial-sum planes
static __device__ __align__(256) unsigned char scratch[kScratchCounterBytes + kScratchPlaneBytes];
static_assert(sizeof(rt::Counters) <= kScratchCounterBytes, "counters must fit their slot");

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define RT_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE,       \
                  "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

constexpr int kEventRing = 64;
constexpr int kLaunchSlots = 4;

// What one in-flight render owns on the device.  Launches on the same stream run one after the
// other and share a slot; launches on different streams may overlap on the GPU, so each stream
// gets a slot of its own.  With more streams than slots the least recently used slot is
// recycled after waiting for its stream.
struct LaunchSlot {
  hipStream_t stream = nullptr;
  bool bound = false;
  rt::Counters* counters = nullptr;       // device
  float* ring = nullptr;                  // device: the waves' parked slice sums (grow-only;
  size_t ring_cap = 0;                    // fixed for the no-alloc slot), capacity in floats
  size_t ring_stride = 0;                 // floats per wave in the slot's last launch
  uint64_t last_use = 0;
  rt_stats last = {};
  uint64_t launch_no = 0;                 // 1-based number of the slot's last launch; 0 = none
};

// What the host-pointer entry points keep between calls on one device: the device copy of the
// last World they were given, a frame buffer in device memory (tile sums) and a page-locked host
// buffer the kernels write resolved frames into.  A host
// that renders frame after frame of one scene (the reference's main() renders the same World
// twice, src/main.rs:905-937) then pays for upload and allocation once.
struct FrameCache {
  rt_scene* scene = nullptr;
  float* frame = nullptr;                 // device
  size_t frame_cap = 0;                   // floats
  float* pinned = nullptr;                // host, page-locked
  size_t pinned_cap = 0;                  // floats
  hipEvent_t done = nullptr;              // behind the last kernel of the share in flight
};

struct State {
  bool ready = false;
  int device = -1;
  int cus = 0;
  hipStream_t stream = nullptr;
  LaunchSlot slots[kLaunchSlots];
  LaunchSlot noalloc_slot;                // counters and planes inside `scratch`
  void* scratch = nullptr;                // this device's address of `scratch`
  LaunchSlot* last_slot = nullptr;
  rt::TileDev* tiles = nullptr;           // device, grow-only
  size_t tiles_cap = 0;
  std::vector<rt::TileDev> tiles_host;    // what ctx().tiles currently holds
  hipEvent_t ev_begin[kEventRing], ev_end[kEventRing];
  uint64_t launches = 0;
  FrameCache cache;
};
// One State per GPU the process drives.  Context 0 is the one rt_init() makes; rt_render_multi
// adds one per listed device and puts it in charge (ctx()) around the calls it makes for that
// device.  Everything else in this file just says ctx().
State g_primary;
std::vector<State*> g_extra;     // contexts of rt_render_multi, by position in its device list
State* g_cur = &g_primary;
inline State& ctx() { return *g_cur; }

// The contexts are process-wide and unsynchronised; every entry point takes this lock, so calls
// from several host threads are serialised rather than racing (entry points call each other,
// hence recursive).
std::recursive_mutex g_api_mutex;

// An entry point runs with the library's device current and gives the calling thread its own
// current device back when it leaves (the caller's, and its framework's, notion of the current
// device is not ours to change).
class ApiScope {
 public:
  ApiScope() : lock_(g_api_mutex) {
    if (hipGetDevice(&saved_) != hipSuccess) {
      (void)hipGetLastError();
      saved_ = -1;
    }
  }
  ~ApiScope() {
    if (saved_ >= 0) (void)hipSetDevice(saved_);
  }

 private:
  std::lock_guard<std::recursive_mutex> lock_;
  int saved_ = -1;
};

int acquire_slot(hipStream_t stream, LaunchSlot** out) {
  int pick = -1;
  for (int i = 0; i < kLaunchSlots; ++i)
    if (ctx().slots[i].bound && ctx().slots[i].stream == stream) pick = i;
  if (pick < 0) {
    for (int i = 0; i < kLaunchSlots && pick < 0; ++i)
      if (!ctx().slots[i].bound) pick = i;
    if (pick < 0) {
      pick = 0;
      for (int i = 1; i < kLaunchSlots; ++i)
        if (ctx().slots[i].last_use < ctx().slots[pick].last_use) pick = i;
      RT_HIP(hipStreamSynchronize(ctx().slots[pick].stream));   // its buffers may still be in use
    }
    ctx().slots[pick].stream = stream;
    ctx().slots[pick].bound = true;
  }
  ctx().slots[pick].last_use = ctx().launches + 1;
  *out = &ctx().slots[pick];
  return RT_OK;
}

// Per-sphere constant of the kernel's conservative discriminant scan.
//
// The scan decides which spheres get the reference's exact Sphere::hit arithmetic.  It evaluates
//     disc' = hb'^2 - c',   hb' = o.dn - c.dn,   c' = k' - 2 o.c + |o|^2 (1 - 2^-17)
// with dn = d/|d| (slightly lengthened), i.e. the reference's discriminant divided by |d|^2 and
// expanded so that no o - c subtraction is needed.  The kernel works on disc'/2, as
//     (hb'/sqrt 2)^2 + (c.o - k'/2)   compared with   |o|^2 (1 - 2^-17) / 2
// (the record stores -k'/2, the per-ray direction carries the 1/sqrt 2; the halvings are exact
// and scale every term and every rounding error alike): 7 fused instructions and a compare
// instead of the 11 of the literal form, and one rounding fewer than disc' itself.  The
// expansion rounds differently from src/main.rs:131-135, so k' carries a slack that makes the
// scan CONSERVATIVE: with u = 2^-24, the reference's computed discriminant is within
// 22u |d|^2 (|o-c|^2 + r^2) of the exact one and the scan's within 24.3u (|o|+|c|)^2 + 4u r^2, so
// any slack >= 93u (|o|^2 + |c|^2) + 26u r^2 guarantees that a sphere whose reference discriminant
// is not negative is never skipped.  The kernel takes 2^-17 |o|^2 off per ray; here
// 2^-17 |c|^2 + 2^-19 r^2 comes off per sphere.  False positives only cost a replay.
// Computed in double and rounded toward -infinity.
float trigger_constant(const rt_sphere& s) {
  const double c2 = (double)s.x * s.x + (double)s.y * s.y + (double)s.z * s.z;
  const double r2 = (double)s.radius * (double)s.radius;
  const double k = c2 - r2 - std::ldexp(c2, -17) - std::ldexp(r2, -19);
  float kf = (float)k;
  if ((double)kf > k) kf = std::nextafterf(kf, -INFINITY);
  return -0.5f * kf;
}

// ---- the scene as the matrix scan reads it (scan_chunk_mfma in rt_kernels.hip) -----------------
// Per sphere one row of 32 f16: the ten numbers of the bilinear form in two halves each (k in
// three), laid out to meet the ray's column slot by slot, by quarters of 8 slots, in chunks of
// kMfmaChunk spheres padded by one block of rows that no ray passes.
// Order: the spheres inside the filter's range sorted along a Morton curve through their centres
// (neighbours in space share a block of 16 rows, so the rays of a wave, which travel together,
// find their candidates in few blocks), then the spheres outside it, which have no row and are
// listed for every ray.  A sphere is outside when it is one of at most kMfmaMaxAlways that stand
// a factor 4 above the rest in max(|c|, r), like the ground of the reference's scenes: the scale
// (a power of two, exact on every coordinate) is chosen so that the rest sits below 128.
struct MfmaScene {
  bool ok = false;
  float scale = 1.0f;
  uint32_t n_always = 0;
  std::vector<uint16_t> rows;      // f16 bit patterns, [chunk][4][kMfmaRows][8]
  std::vector<float4> geom;        // [position]
  std::vector<uint32_t> index;     // [position]
};

uint16_t f16_bits(_Float16 h) {
  uint16_t b;
  memcpy(&b, &h, sizeof b);
  return b;
}

void mfma_row_threshold(uint16_t* row) {   // what meets the three parts of the ray's -threshold
  row[29] = f16_bits((_Float16)32768.0f);
  row[30] = f16_bits((_Float16)32.0f);
  row[31] = f16_bits((_Float16)0.03125f);
}

// A row no ray passes.  Everything but k is zero, so a ray's entry is -k/2 minus its threshold,
// and the threshold can be as low as minus the slack taken off it: (2^-16 + 2^-23) |o|^2 + 1.5 *
// 2^-22 with scaled |o|^2 below 2^31 (kMfmaOriginLimit2 in rt_kernels.hip), i.e. up to 33 024.
// k is therefore the largest f16 in both its parts: -65 504 stays negative against that.  (A
// single part of 60 000 was enough for a slack of 2^-17 |o|^2, and not for 2^-16: rays from
// beyond 0.96 of the origin limit listed the padding rows of a chunk.)
void mfma_row_never(uint16_t* row) {
  for (int i = 0; i < 32; ++i) row[i] = 0;
  row[27] = f16_bits((_Float16)65504.0f);
  row[28] = f16_bits((_Float16)65504.0f);
  mfma_row_threshold(row);
}

void mfma_row(const rt_sphere& sp, double scale, uint16_t* row) {
  const double x = sp.x * scale, y = sp.y * scale, z = sp.z * scale, r = sp.radius * scale;
  const double c2 = x * x + y * y + z * z, r2 = r * r, c1 = std::sqrt(c2);
  const double S[9] = {x * x, y * y, z * z, x * y, x * z, y * z, x, y, z};
  for (int t = 0; t < 9; ++t) {
    const _Float16 hi = (_Float16)S[t];
    const _Float16 lo = (_Float16)(S[t] - (double)hi);
    row[t] = f16_bits(hi);
    row[9 + t] = f16_bits(hi);
    row[18 + t] = f16_bits(lo);
  }
  // k with the sphere side's slack taken off, in two parts that do not add up to more than k
  const double k = c2 - r2 - std::ldexp(c2, -16) - std::ldexp(r2, -17) - std::ldexp(c2 + c1, -21);
  _Float16 kh = (_Float16)k;
  _Float16 kl = (_Float16)(k - (double)kh);
  while ((double)kh + (double)kl > k) {
    uint16_t b = f16_bits(kl);
    // one f16 step towards -infinity
    if (b == 0x0000) b = 0x8001; else if (b & 0x8000) b++; else b--;
    memcpy(&kl, &b, sizeof kl);
  }
  row[27] = f16_bits(kh);
  row[28] = f16_bits(kl);
  mfma_row_threshold(row);
}

uint32_t morton10(uint32_t v) {   // spread the low 10 bits of v three apart
  v &= 0x3FF;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

void build_mfma_scene(const rt_sphere* spheres, uint32_t n, MfmaScene* out) {
  MfmaScene& m = *out;
  m = MfmaScene();
  if (n == 0) return;
  std::vector<double> size(n);
  std::vector<uint32_t> by_size(n);
  for (uint32_t i = 0; i < n; ++i) {
    const rt_sphere& s = spheres[i];
    const double c1 = std::sqrt((double)s.x * s.x + (double)s.y * s.y + (double)s.z * s.z);
    size[i] = std::max(c1, std::fabs((double)s.radius));
    if (!std::isfinite(size[i])) return;
    by_size[i] = i;
  }
  std::stable_sort(by_size.begin(), by_size.end(),
                   [&](uint32_t a, uint32_t b) { return size[a] > size[b]; });
  // the last gap of a factor 4 among the largest few separates the spheres that get no row
  uint32_t n_always = 0;
  for (uint32_t k = 1; k <= rt::kMfmaMaxAlways && k < n; ++k)
    if (size[by_size[k - 1]] >= 4.0 * size[by_size[k]]) n_always = k;
  const double bulk = size[by_size[n_always]];
  int e = 0;
  if (bulk > 0.0) {
    e = (int)std::floor(std::log2(128.0 / bulk));
    while (std::ldexp(bulk, e) > 128.0) --e;
    if (e > 60) e = 60;
    if (e < -60) return;
  }
  const double scale = std::ldexp(1.0, e);
  std::vector<char> always(n, 0);
  for (uint32_t k = 0; k < n_always; ++k) always[by_size[k]] = 1;

  // Morton order of the others
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t i = 0; i < n; ++i) {
    if (always[i]) continue;
    const double c[3] = {spheres[i].x, spheres[i].y, spheres[i].z};
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
  }
  std::vector<std::pair<uint32_t, uint32_t>> order;   // (code, sphere)
  order.reserve(n);
  for (uint32_t i = 0; i < n; ++i) {
    if (always[i]) continue;
    const double c[3] = {spheres[i].x, spheres[i].y, spheres[i].z};
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
      const double span = hi[a] - lo[a];
      const uint32_t q = span > 0.0 ? (uint32_t)std::min(1023.0, (c[a] - lo[a]) / span * 1024.0) : 0u;
      code |= morton10(q) << a;
    }
    order.push_back(std::make_pair(code, i));
  }
  std::sort(order.begin(), order.end());
  m.index.reserve(n);
  for (const auto& o : order) m.index.push_back(o.second);
  for (uint32_t i = 0; i < n; ++i)
    if (always[i]) m.index.push_back(i);

  const uint32_t n_in = n - n_always;
  const uint32_t chunks = (n + rt::kMfmaChunk - 1) / rt::kMfmaChunk;
  m.rows.assign((size_t)chunks * 4 * rt::kMfmaRows * 8, 0);
  m.geom.resize(n);
  uint16_t row[32];
  for (uint32_t ch = 0; ch < chunks; ++ch) {
    for (uint32_t i = 0; i < rt::kMfmaRows; ++i) {
      const uint32_t pos = ch * rt::kMfmaChunk + i;
      if (i < rt::kMfmaChunk && pos < n_in) mfma_row(spheres[m.index[pos]], scale, row);
      else mfma_row_never(row);
      for (int q = 0; q < 4; ++q)
        memcpy(&m.rows[(((size_t)ch * 4 + q) * rt::kMfmaRows + i) * 8], row + 8 * q, 16);
    }
  }
  for (uint32_t pos = 0; pos < n; ++pos) {
    const rt_sphere& s = spheres[m.index[pos]];
    volatile float r2 = s.radius * s.radius;   // `self.radius * self.radius`, src/main.rs:134
    m.geom[pos] = make_float4(s.x, s.y, s.z, r2);
  }
  m.scale = (float)scale;
  m.n_always = n_always;
  m.ok = true;
}

int check_ready() {
  if (!ctx().ready) return fail(RT_ERR_NOT_INITIALIZED, "rt_init() has not been called");
  // every call below allocates, copies or launches on the library's device, whatever device
  // the caller made current in between (ApiScope puts the caller's back afterwards)
  RT_HIP(hipSetDevice(ctx().device));
  return RT_OK;
}

// A device-resident output must live on the library's device (or be host memory the device can
// write, which HIP reports as such); a pointer HIP does not know cannot be written by a kernel.
int check_device_pointer(const void* p, const char* what) {
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof attr);
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(RT_ERR_INVALID_ARGUMENT, "%s is not memory the device can reach", what);
  }
  if (attr.type == hipMemoryTypeDevice && attr.device != ctx().device)
    return fail(RT_ERR_INVALID_ARGUMENT, "%s lives on device %d, the library on device %d", what,
                attr.device, ctx().device);
  if (attr.type == hipMemoryTypeUnregistered)
    return fail(RT_ERR_INVALID_ARGUMENT, "%s is pageable host memory, not device memory", what);
  return RT_OK;
}

int check_frame(const rt_frame* f) {
  if (!f) return fail(RT_ERR_INVALID_ARGUMENT, "frame is NULL");
  if (f->image_width == 0 || f->image_height == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "image is %ux%u", f->image_width, f->image_height);
  if (f->samples_per_pixel == 0) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel is 0");
  if ((uint64_t)f->image_width * f->image_height > 0x7FFFFFFFull)
    return fail(RT_ERR_LIMIT, "more than 2^31 pixels");
  if (f->flags & ~(RT_FLAG_STRICT_FP | RT_FLAG_SEQUENTIAL_SUM | RT_FLAG_ACCEL | RT_FLAG_SCAN_VALU))
    return fail(RT_ERR_INVALID_ARGUMENT, "unknown flag bits 0x%x", f->flags);
  return RT_OK;
}

// How one pixel's samples are cut into work items.  Slice k runs samples
// [begin[k], begin[k+1]) in one lane; the wave that runs a pixel's slices adds their sums in slice
// order and writes the pixel (merge_ready in rt_kernels.hip).  Two reasons to cut:
//  * granularity: a path at max_depth 50 inside glass can cost 20x the average, and a lane
//    runs its item to the end while the frame's last items drain, so items must be small where
//    it matters.  The last eight slices of a long plan are tapered down to single samples
//    (.., B, B, 4, 4, 2, 2, 1, 1, 1, 1).
//  * small images (and a rank's share of a frame): fewer pixels than lanes.
// The cut is a function of the FULL image and spp only, so a frame gives the same sums
// whatever part of it a call renders and however many GPUs share it.
// Writes begin[0..n] and returns n, the number of slices (at most rt::kMaxSlices).
uint32_t plan_slices(const rt_frame& f, uint32_t* begin) {
  const uint32_t spp = f.samples_per_pixel;
  uint32_t sizes[rt::kMaxSlices];
  uint32_t n = 0;
  if (f.flags & RT_FLAG_SEQUENTIAL_SUM) {
    sizes[n++] = spp;
  } else {
    // Full-size items run `big` samples: at most kSamplesPerItem, and no more than ~2 % of what
    // one lane-slot gets when the frame is split over 8 GPUs (a lane runs an item to its end,
    // so an item of deep glass paths must stay a small part of a GPU's share of the frame).
    const uint64_t total = (uint64_t)f.image_width * f.image_height * spp;
    const uint64_t per_slot_of_8 = total / (8ull * rt::kLaneSlots);
    uint32_t big = (uint32_t)(per_slot_of_8 / 48);
    if (big > rt::kSamplesPerItem) big = rt::kSamplesPerItem;
    if (big < 1) big = 1;
    static const uint32_t taper[8] = {4, 4, 2, 2, 1, 1, 1, 1};   // 16 samples
    const bool use_taper = big > 2 && spp >= 32;
    const uint32_t body = use_taper ? spp - 16 : spp;
    const uint32_t room = rt::kMaxSlices - (use_taper ? 8 : 0);
    if ((body + big - 1) / big > room) big = (body + room - 1) / room;
    for (uint32_t left = body; left > 0;) {
      const uint32_t k = left < big ? left : big;
      sizes[n++] = k;
      left -= k;
    }
    if (use_taper) {
      for (uint32_t t : taper) sizes[n++] = t;
    }
    // small image: more, equal slices so that every lane has an item
    const uint64_t pixels = (uint64_t)f.image_width * f.image_height;
    uint64_t want = pixels >= rt::kTargetItems ? 1 : (rt::kTargetItems + pixels - 1) / pixels;
    if (want > spp) want = spp;
    if (want > rt::kMaxSlices) want = rt::kMaxSlices;
    if (want > n) {
      n = 0;
      for (uint64_t k = 0; k < want; ++k)
        sizes[n++] = (uint32_t)(((k + 1) * spp) / want - (k * spp) / want);
    }
  }
  begin[0] = 0;
  for (uint32_t k = 0; k < n; ++k) begin[k + 1] = begin[k] + sizes[k];
  return n;
}

// The optional hierarchy (RT_FLAG_ACCEL) is built the first time a scene is rendered with it:
// a scene that only ever takes the linear scan pays nothing for it.
int ensure_bvh(const rt_scene* s) {
  if (s->bvh_built) return RT_OK;
  rt::BvhHost bvh;
  rt::build_bvh(s->host_spheres.data(), (uint32_t)s->host_spheres.size(), &bvh);
  const size_t nn = bvh.nodes.empty() ? 1 : bvh.nodes.size();
  const size_t nl = bvh.leaf_geom.empty() ? 1 : bvh.leaf_geom.size();
  RT_HIP(hipMalloc(&s->bvh, nn * sizeof(rt::BvhNode)));
  RT_HIP(hipMalloc(&s->leaf_geom, nl * sizeof(float4)));
  RT_HIP(hipMalloc(&s->leaf_index, nl * sizeof(uint32_t)));
  if (!bvh.nodes.empty())
    RT_HIP(hipMemcpy(s->bvh, bvh.nodes.data(), bvh.nodes.size() * sizeof(rt::BvhNode),
                     hipMemcpyHostToDevice));
  if (!bvh.leaf_geom.empty()) {
    RT_HIP(hipMemcpy(s->leaf_geom, bvh.leaf_geom.data(), bvh.leaf_geom.size() * sizeof(float4),
                     hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(s->leaf_index, bvh.leaf_index.data(),
                     bvh.leaf_index.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  s->bvh_depth = bvh.depth;
  s->dev.bvh_info = bvh.info;
  s->dev.bvh = (const rt::BvhNode*)s->bvh;
  s->dev.leaf_geom = (const float4*)s->leaf_geom;
  s->dev.leaf_index = (const uint32_t*)s->leaf_index;
  s->bvh_built = true;
  return RT_OK;
}

// Launch-geometry queries do not change for the life of the process; asking once per device and
// kernel variant keeps the call path of a frame down to its launches.  (Per device: the matrix
// kernel's query also raises that kernel's dynamic-LDS limit, which is a per-device attribute.)
int blocks_per_cu(bool strict, bool multi, bool accel, bool mfma, size_t lds, int* out,
                  hipStream_t capturing = nullptr) {
  struct Entry { bool valid; size_t lds; int per_cu; };
  static std::vector<Entry> cache;
  const size_t at = (size_t)ctx().device * 16 +
                    ((strict ? 1 : 0) | (multi ? 2 : 0) | (accel ? 4 : 0) | (mfma ? 8 : 0));
  if (cache.size() <= at) cache.resize(at + 1, Entry{false, 0, 0});
  Entry& e = cache[at];
  if (!e.valid || e.lds != lds) {
    // (`capturing`: a stream that may be under HIP graph capture, where a query that also sets a
    // kernel attribute has no place; the caller is told to make the first call outside)
    if (capturing) {
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(capturing, &st) != hipSuccess) (void)hipGetLastError();
      if (st != hipStreamCaptureStatusNone)
        return fail(RT_ERR_INVALID_ARGUMENT,
                    "first use of this kernel variant and scene size on this device: its launch "
                    "geometry has to be queried once outside stream capture (render once through "
                    "any entry point before capturing)");
    }
    int per_cu = 0;
    if (mfma) RT_HIP(rt::trace_max_blocks_per_cu_mfma(multi, lds, &per_cu));
    else RT_HIP(rt::trace_max_blocks_per_cu(strict, multi, accel, lds, &per_cu));
    e.valid = true;
    e.lds = lds;
    e.per_cu = per_cu;
  }
  *out = e.per_cu;
  return RT_OK;
}

// `sample_begin` / `accumulate`: the call adds samples [sample_begin, sample_begin + spp) of
// every pixel to what d_out already holds (progressive rendering; resolve must be 0).
// `no_alloc`: the call may not allocate, free or wait; its counters and partial planes are the
// fixed ones in `scratch`, and a request they cannot hold is refused.
int render_tiles_device(const rt_scene* scene, const rt_frame* frame, const rt::TileDev* tiles,
                        size_t n_tiles, uint64_t out_pixels, int resolve, float* d_out,
                        hipStream_t stream, uint32_t sample_begin = 0, bool accumulate = false,
                        bool no_alloc = false) {
  rt::TraceParams p;
  memset(&p, 0, sizeof p);
  p.cam = frame->camera;
  // origin = view_inv * (0,0,0,1) in glam's order ((c0*x + c1*y) + c2*z) + c3*w
  for (int i = 0; i < 3; ++i) {
    const float* V = frame->camera.view_inv;
    volatile float r = V[i] * 0.0f;
    r = r + V[4 + i] * 0.0f;
    r = r + V[8 + i] * 0.0f;
    r = r + V[12 + i] * 1.0f;
    p.origin[i] = r;
    const float* PI = frame->camera.proj_inv;
    volatile float z0 = PI[8 + i] * 0.0f, w1 = PI[12 + i] * 1.0f, w0 = V[12 + i] * 0.0f;
    p.proj_z0[i] = z0;
    p.proj_w1[i] = w1;
    p.view_w0[i] = w0;
  }
  p.width = frame->image_width;
  p.height = frame->image_height;
  p.fwidth = (float)frame->image_width;
  p.fheight = (float)frame->image_height;
  p.inv_width = 1.0f / p.fwidth;
  p.inv_height = 1.0f / p.fheight;
  p.max_depth = frame->max_depth;
  p.spp = frame->samples_per_pixel;
  p.seed_a = rt_rng_seed_a(frame->seed);
  uint32_t slice_begin[rt::kMaxSlices + 1];
  p.slices = plan_slices(*frame, slice_begin);
  for (uint32_t k = 0; k <= p.slices; ++k) p.slice_begin[k] = sample_begin + slice_begin[k];
  p.slices_div = rt::make_divider(p.slices);
  // a wave reserves whole pixels: 64 when a pixel is one item, otherwise as many as make up to
  // kGrabItems items (1 to kGrabPixelsMax)
  p.grab_pixels = p.slices == 1 ? rt::kItemsPerGrab
                                : std::min(rt::kGrabPixelsMax, std::max(1u, rt::kGrabItems / p.slices));
  p.n_tiles = (uint32_t)n_tiles;
  uint64_t total_pixels = 0;
  for (size_t i = 0; i < n_tiles; ++i) total_pixels += (uint64_t)tiles[i].w * tiles[i].h;
  if (total_pixels == 0) return RT_OK;
  if (total_pixels * p.slices >= 0xFFFFFF00ull || out_pixels * p.slices >= 0xFFFFFF00ull)
    return fail(RT_ERR_LIMIT, "more than 2^32 work items in one call");
  p.total_pixels = (uint32_t)total_pixels;
  p.total_items = (uint32_t)(total_pixels * p.slices);
  p.out_pixels = out_pixels;
  p.inv_spp = 1.0f / (float)frame->samples_per_pixel;   // `1.0 / spp as f32`, src/cpu/mod.rs:67
  if (scene->device != ctx().device)
    return fail(RT_ERR_INVALID_ARGUMENT, "scene lives on device %d, the library on device %d",
                scene->device, ctx().device);

  const bool strict = (frame->flags & RT_FLAG_STRICT_FP) != 0;
  const bool accel = (frame->flags & RT_FLAG_ACCEL) != 0;
  if (accel) {
    if (no_alloc && !scene->bvh_built)
      return fail(RT_ERR_INVALID_ARGUMENT,
                  "RT_FLAG_ACCEL: the scene's hierarchy has not been built yet, and this entry "
                  "point may not allocate it (render once through rt_render_rows_device_async)");
    if (int rc = ensure_bvh(scene)) return rc;
    if (scene->bvh_depth > rt::bvh_stack_depth())
      return fail(RT_ERR_LIMIT, "hierarchy of %u levels exceeds the traversal stack (%u)",
                  scene->bvh_depth, rt::bvh_stack_depth());
  }
  p.scene = scene->dev;

  LaunchSlot* slot = nullptr;
  if (no_alloc) {
    slot = &ctx().noalloc_slot;
    if (n_tiles != 1) return fail(RT_ERR_INVALID_ARGUMENT, "one tile per call");
    slot->stream = stream;
    slot->last_use = ctx().launches + 1;
  } else if (int rc = acquire_slot(stream, &slot)) {
    return rc;
  }
  p.counters = slot->counters;

  // Equal row blocks at a constant pitch (a rank's share of a frame): the trace kernel takes
  // them as one tile with a rule for its rows, and looks nothing up per work item.
  bool row_blocks = n_tiles > 1 && tiles[0].h > 0;
  for (size_t i = 1; row_blocks && i < n_tiles; ++i) {
    const rt::TileDev &a = tiles[0], &b = tiles[i];
    row_blocks = b.x == a.x && b.w == a.w && b.out_stride == a.out_stride &&
                 (b.h == a.h || (i + 1 == n_tiles && b.h <= a.h && b.h > 0)) &&
                 b.first_pixel == (uint64_t)i * a.w * a.h && tiles[1].y > a.y &&
                 b.y == a.y + i * (uint64_t)(tiles[1].y - a.y) && tiles[1].out_offset > a.out_offset &&
                 b.out_offset == a.out_offset + i * (tiles[1].out_offset - a.out_offset);
  }
  if (row_blocks) {
    p.row_block = tiles[0].h;
    p.row_period = tiles[1].y - tiles[0].y;
    p.out_period = tiles[1].out_offset - tiles[0].out_offset;
    p.row_block_div = rt::make_divider(p.row_block);
    if (p.out_period >= 0xFFFFFFFFull) row_blocks = false, p.row_block = 0;
  }
  if (n_tiles == 1) {
    p.tile0 = tiles[0];
    p.tiles = nullptr;
  } else {
    // The table lives in one grow-only device buffer.  A caller that renders the same tile
    // list frame after frame (a rank's row blocks) pays for the upload once.
    const bool same = ctx().tiles_host.size() == n_tiles &&
                      memcmp(ctx().tiles_host.data(), tiles, n_tiles * sizeof(rt::TileDev)) == 0;
    if (!same) {
      RT_HIP(hipDeviceSynchronize());   // an earlier launch, on any stream, may still read it
      if (n_tiles > ctx().tiles_cap) {
        if (ctx().tiles) RT_HIP(hipFree(ctx().tiles));
        ctx().tiles = nullptr;
        ctx().tiles_cap = 0;
        ctx().tiles_host.clear();
        RT_HIP(hipMalloc((void**)&ctx().tiles, n_tiles * sizeof(rt::TileDev)));
        ctx().tiles_cap = n_tiles;
      }
      RT_HIP(hipMemcpy(ctx().tiles, tiles, n_tiles * sizeof(rt::TileDev), hipMemcpyHostToDevice));
      ctx().tiles_host.assign(tiles, tiles + n_tiles);
    }
    p.tiles = ctx().tiles;
  }

  rt::ZeroParams zp;
  memset(&zp, 0, sizeof zp);
  zp.out = d_out;
  zp.n_tiles = p.n_tiles;
  zp.total_pixels = p.total_pixels;
  zp.tile0 = p.tile0;
  zp.tiles = p.tiles;
  if (row_blocks) {   // the trace kernel takes the rule instead of the list
    p.n_tiles = 1;
    p.tile0 = tiles[0];
  }

  if (frame->max_depth == 0) {
    // shoot_ray(ray, 0) is black (src/main.rs:192-194): every sum is 0 and nothing is traced.
    // Only the pixels the tiles cover are written, like in every other case (the gaps of a
    // strided output may belong to another call); adding zero to a running sum changes nothing.
    if (!accumulate) RT_HIP(rt::launch_zero(zp, stream));
    slot->last = {};
    slot->last.samples = total_pixels * frame->samples_per_pixel;
    slot->launch_no = 0;
    ctx().last_slot = slot;
    return RT_OK;
  }

  p.resolve = resolve;
  p.accumulate = accumulate ? 1 : 0;
  p.out = d_out;

  bool bvh_in_lds = false;
  p.bvh_stack = scene->bvh_depth ? scene->bvh_depth : 1;
  // the scan's filter runs on the matrix pipe unless the request or the scene says otherwise
  // (a camera farther out than the matrix filter's origins reach would send every primary ray
  // through the literal loop: such a frame keeps to the vector scan as well)
  const double cam2 = ((double)p.origin[0] * p.origin[0] + (double)p.origin[1] * p.origin[1] +
                       (double)p.origin[2] * p.origin[2]) * scene->dev.mfma_scale * scene->dev.mfma_scale;
  const bool mfma = !strict && !accel && !(frame->flags & RT_FLAG_SCAN_VALU) && scene->dev.mfma_ok &&
                    cam2 < 536870912.0;   // 2^29: a quarter of the filter's limit on |o|^2
  const size_t lds = mfma ? rt::trace_lds_bytes_mfma(rt::kThreadsMfma)
                          : rt::trace_lds_bytes(scene->dev.n_spheres, accel, scene->dev.bvh_info.n_nodes,
                                                p.bvh_stack, &bvh_in_lds, &p.slot_table_offset);
  p.bvh_in_lds = bvh_in_lds ? 1 : 0;
  const bool multi = scene->dev.n_spheres > (mfma ? rt::kMfmaChunk : rt::kChunkSpheres);
  int per_cu = 0;
  if (int rc = blocks_per_cu(strict, multi, accel, mfma, lds, &per_cu, no_alloc ? stream : nullptr))
    return rc;
  if (per_cu < 1) return fail(RT_ERR_DEVICE, "trace kernel does not fit on a CU (LDS %zu B)", lds);
  int blocks = per_cu * ctx().cus;
  const uint32_t paths_per_lane = accel ? 1 : rt::kRaysPerLane;
  const uint64_t lanes_needed = ((uint64_t)p.total_items + paths_per_lane - 1) / paths_per_lane;
  const uint32_t threads = (uint32_t)rt::trace_threads(multi, accel, mfma);
  const uint64_t blocks_needed = (lanes_needed + threads - 1) / threads;
  if ((uint64_t)blocks > blocks_needed) blocks = (int)blocks_needed;
  if (blocks < 1) blocks = 1;
  // A workgroup reserves chunks of the frame for its waves: kChunkGrabs reservations for each of
  // them while there is plenty of work, shrinking once fewer than RT_GUIDE_FACTOR such chunks per
  // workgroup are left (see wg_reserve in rt_kernels.hip).  Measured on the benchmark frame and
  // on one rank's eighth of it: the whole frame does not care (29.7-30.0 ms for chunks of 16 to
  // 128 pixels), the eighth wants small chunks early (4.77 ms against 5.1-5.25 for 128-pixel
  // chunks kept to the end).
  p.chunk_max = p.grab_pixels * (threads / rt::kWave) * rt::kChunkGrabs;
  p.chunk_min = p.grab_pixels;
  p.guide_shift = 0;
  while ((1ull << p.guide_shift) < (uint64_t)RT_GUIDE_FACTOR * (uint64_t)blocks) ++p.guide_shift;

  // Sliced samples: every wave of the launch parks the slice sums of its open pixels in a ring
  // of its own, ring_slots pixels long (all of kRingSlots when the memory is the library's to
  // size; as many as fit for the entry point that may not allocate).
  if (p.slices > 1) {
    const size_t waves = (size_t)blocks * (threads / rt::kWave);
    const size_t per_slot = (size_t)3 * p.slices;               // floats
    p.ring_slots = rt::kRingSlots;
    if (no_alloc) {
      const size_t fit = slot->ring_cap / (waves * per_slot);
      if (fit < p.ring_slots) p.ring_slots = (uint32_t)fit;
      if (p.ring_slots < 1)
        return fail(RT_ERR_LIMIT, "%zu waves x %u sample slices do not fit the fixed buffer", waves,
                    p.slices);
    }
    if (p.grab_pixels > p.ring_slots) p.grab_pixels = p.ring_slots;
    const size_t need = waves * p.ring_slots * per_slot;
    if (need > slot->ring_cap) {              // never true for the fixed slot
      RT_HIP(hipStreamSynchronize(stream));   // the slot's earlier launches use the old buffer
      if (slot->ring) RT_HIP(hipFree(slot->ring));
      slot->ring = nullptr;
      slot->ring_cap = 0;
      RT_HIP(hipMalloc((void**)&slot->ring, need * sizeof(float)));
      slot->ring_cap = need;
    }
    p.ring = slot->ring;
    slot->ring_stride = (size_t)p.ring_slots * per_slot;
  }

  // the counters must start at zero; the rings need no clearing (a parked sum is read only
  // after it was written)
  // (for the entry point that may run under graph capture the clearing is a kernel: two
  // hipMemsetAsync nodes on the same address in one captured graph were seen to take effect once
  // per replay, not twice - the second launch of a replay then found the first one's cursor)
  if (no_alloc) RT_HIP(rt::launch_clear_counters(slot->counters, stream));
  else RT_HIP(hipMemsetAsync(slot->counters, 0, sizeof(rt::Counters), stream));
  if (no_alloc) {
    // nothing but the memset and the kernel goes onto the caller's stream: under graph capture
    // both become nodes of the graph, and an event pair would not be one this library can read
    // afterwards (rt_get_stats reports kernel_ms as NaN for such a launch)
    RT_HIP(rt::launch_trace(p, strict, accel, mfma, blocks, stream));
    slot->launch_no = 0;
  } else {
    const int ev = (int)(ctx().launches % kEventRing);
    RT_HIP(hipEventRecord(ctx().ev_begin[ev], stream));
    RT_HIP(rt::launch_trace(p, strict, accel, mfma, blocks, stream));
    RT_HIP(hipEventRecord(ctx().ev_end[ev], stream));
    ctx().launches++;
    slot->launch_no = ctx().launches;
  }

  slot->last = {};
  slot->last.rays_per_lane = paths_per_lane;
  slot->last.slices = p.slices;
  slot->last.workgroups = (uint32_t)blocks;
  slot->last.threads_per_workgroup = threads;
  // multiplied by rays in rt_get_stats; a traversal does not count its tests
  slot->last.sphere_tests = accel ? 0 : scene->dev.n_spheres;
  ctx().last_slot = slot;
  return RT_OK;
}

int check_request(const rt_request* req, bool need_tiles) {
  if (!req) return fail(RT_ERR_INVALID_ARGUMENT, "request is NULL");
  if (req->n_spheres && !req->spheres) return fail(RT_ERR_INVALID_ARGUMENT, "spheres is NULL");
  if (req->n_materials && !req->materials) return fail(RT_ERR_INVALID_ARGUMENT, "materials is NULL");
  if (need_tiles && req->n_tiles && !req->tiles) return fail(RT_ERR_INVALID_ARGUMENT, "tiles is NULL");
  return RT_OK;
}

rt_frame frame_of(const rt_request* req) {
  rt_frame f;
  f.image_width = req->image_width;
  f.image_height = req->image_height;
  f.max_depth = req->max_depth;
  f.samples_per_pixel = req->samples_per_pixel;
  f.seed = req->seed;
  f.flags = req->flags;
  f.camera = req->camera;
  return f;
}

int check_tiles(const rt_tile* tiles, uint32_t n, uint32_t w, uint32_t h) {
  for (uint32_t i = 0; i < n; ++i) {
    const rt_tile& t = tiles[i];
    if ((uint64_t)t.x + t.width > w || (uint64_t)t.y + t.height > h)
      return fail(RT_ERR_INVALID_ARGUMENT, "tile %u (%u,%u %ux%u) leaves the %ux%u image", i, t.x,
                  t.y, t.width, t.height, w, h);
  }
  return RT_OK;
}

struct ScopedScene {
  rt_scene* s = nullptr;
  ~ScopedScene() { if (s) rt_scene_destroy(s); }
};
struct ScopedDev {
  void* p = nullptr;
  ~ScopedDev() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_last_error(void) { return g_error.c_str(); }

namespace {
// Brings the context in charge up on `device`.
int init_context(int device) {
  int count = 0;
  RT_HIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count)
    return fail(RT_ERR_INVALID_ARGUMENT, "device %d of %d", device, count);
  RT_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RT_ERR_DEVICE, "device %d is %s; this library holds gfx950 code only", device,
                prop.gcnArchName);
  ctx().cus = prop.multiProcessorCount;
  RT_HIP(hipStreamCreateWithFlags(&ctx().stream, hipStreamNonBlocking));
  for (LaunchSlot& sl : ctx().slots) {
    RT_HIP(hipMalloc((void**)&sl.counters, sizeof(rt::Counters)));
    RT_HIP(hipMemset(sl.counters, 0, sizeof(rt::Counters)));
  }
  // This device's copy of `scratch`: asked for once, with the device current, and used through
  // the returned address from here on.
  void* base = nullptr;
  RT_HIP(hipGetSymbolAddress(&base, HIP_SYMBOL(scratch)));
  ctx().scratch = base;
  LaunchSlot& na = ctx().noalloc_slot;
  na = LaunchSlot();
  na.bound = true;
  na.counters = reinterpret_cast<rt::Counters*>(base);
  na.ring = reinterpret_cast<float*>(static_cast<unsigned char*>(base) + kScratchCounterBytes);
  na.ring_cap = kScratchPlaneBytes / sizeof(float);
  for (int i = 0; i < kEventRing; ++i) {
    RT_HIP(hipEventCreate(&ctx().ev_begin[i]));
    RT_HIP(hipEventCreate(&ctx().ev_end[i]));
  }
  RT_HIP(hipEventCreateWithFlags(&ctx().cache.done, hipEventDisableTiming));
  ctx().device = device;
  ctx().launches = 0;
  ctx().ready = true;
  // The launch geometry of the two matrix-filter kernels does not depend on the scene: asked
  // for here, so that a first frame - possibly one being captured into a graph - finds it.
  for (int multi = 0; multi < 2; ++multi) {
    int per_cu = 0;
    if (int rc = blocks_per_cu(false, multi != 0, false, true, rt::trace_lds_bytes_mfma(rt::kThreadsMfma),
                               &per_cu))
      return rc;
  }
  return RT_OK;
}

void destroy_context() {
  if (!ctx().ready) return;
  (void)hipSetDevice(ctx().device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < kEventRing; ++i) {
    (void)hipEventDestroy(ctx().ev_begin[i]);
    (void)hipEventDestroy(ctx().ev_end[i]);
  }
  for (LaunchSlot& sl : ctx().slots) {
    if (sl.counters) (void)hipFree(sl.counters);
    if (sl.ring) (void)hipFree(sl.ring);
  }
  FrameCache& fc = ctx().cache;
  if (fc.done) (void)hipEventDestroy(fc.done);
  if (fc.scene) (void)rt_scene_destroy(fc.scene);
  if (fc.frame) (void)hipFree(fc.frame);
  if (fc.pinned) (void)hipHostFree(fc.pinned);
  if (ctx().tiles) (void)hipFree(ctx().tiles);
  if (ctx().stream) (void)hipStreamDestroy(ctx().stream);
  ctx() = State();
}
}  // namespace

int rt_init(int device) {
  ApiScope scope;
  g_cur = &g_primary;
  if (ctx().ready) {
    if (ctx().device == device) return RT_OK;
    return fail(RT_ERR_INVALID_ARGUMENT, "already initialised on device %d", ctx().device);
  }
  return init_context(device);
}

int rt_shutdown(void) {
  ApiScope scope;
  for (State* ctx : g_extra) {
    g_cur = ctx;
    destroy_context();
    delete ctx;
  }
  g_extra.clear();
  g_cur = &g_primary;
  destroy_context();
  return RT_OK;
}

int rt_scene_create(const rt_sphere* spheres, uint32_t n_spheres, const rt_material* materials,
                    uint32_t n_materials, rt_scene** out_scene) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!out_scene) return fail(RT_ERR_INVALID_ARGUMENT, "out_scene is NULL");
  *out_scene = nullptr;
  if ((n_spheres && !spheres) || (n_materials && !materials))
    return fail(RT_ERR_INVALID_ARGUMENT, "NULL scene arrays");
  if (n_spheres > (1u << 24)) return fail(RT_ERR_LIMIT, "more than 2^24 spheres");
  for (uint32_t i = 0; i < n_spheres; ++i) {
    if (spheres[i].material_id >= n_materials)
      return fail(RT_ERR_INVALID_ARGUMENT, "sphere %u names material %u of %u", i,
                  spheres[i].material_id, n_materials);
    // |c|^2 and r^2 must be finite f32 values for the discriminant scan
    const rt_sphere& sp = spheres[i];
    const double mag = (double)sp.x * sp.x + (double)sp.y * sp.y + (double)sp.z * sp.z +
                       (double)sp.radius * sp.radius;
    if (!(mag < 1e37))
      return fail(RT_ERR_LIMIT, "sphere %u: centre or radius not finite or beyond 1e18", i);
  }

  std::vector<float4> trig(n_spheres ? n_spheres : 1);
  std::vector<float4> geom(n_spheres ? n_spheres : 1);
  std::vector<float4> surf(n_spheres ? 2 * (size_t)n_spheres : 2);
  for (uint32_t i = 0; i < n_spheres; ++i) {
    const rt_sphere& s = spheres[i];
    volatile float r2 = s.radius * s.radius;   // `self.radius * self.radius`, src/main.rs:134
    geom[i] = make_float4(s.x, s.y, s.z, r2);
    trig[i] = make_float4(s.x, s.y, s.z, trigger_constant(s));
    const rt_material& m = materials[s.material_id];
    float idbits, flagbits;
    memcpy(&idbits, &s.material_id, sizeof idbits);
    memcpy(&flagbits, &m.flags, sizeof flagbits);
    surf[2 * (size_t)i] = make_float4(m.r, m.g, m.b, flagbits);
    if (m.flags & RT_MAT_DIELECTRIC) {
      // A dielectric scatters white (src/main.rs:201-227) and nothing reads its colour: the three
      // words hold what the branch computes from ir alone, formed here with the same operations:
      // `1.0 / ir` (src/main.rs:207) and reflectance's r0 (src/main.rs:163-164) for ir and 1 / ir.
      const auto r0_of = [](float ref_index) {
        volatile float q = (1.0f - ref_index) / (1.0f + ref_index);
        volatile float r0 = q * q;
        return (float)r0;
      };
      volatile float inv_ir = 1.0f / m.ir;
      surf[2 * (size_t)i] = make_float4(inv_ir, r0_of(inv_ir), r0_of(m.ir), flagbits);
    }
    surf[2 * (size_t)i + 1] = make_float4(m.metallic_strength, m.ir, s.radius, idbits);
    if ((m.flags & RT_MAT_METALLIC) && !(m.flags & RT_MAT_DIELECTRIC)) {
      // A metal's attenuation is its colour applied twice and its fuzz 1 - metallic_strength
      // (src/main.rs:199, 230, 236-242): the products and the difference, formed here with the
      // same operations, take the colour's and the strength's places.
      volatile float rr = m.r * m.r, gg = m.g * m.g, bb = m.b * m.b, fuzz = 1.0f - m.metallic_strength;
      surf[2 * (size_t)i] = make_float4(rr, gg, bb, flagbits);
      surf[2 * (size_t)i + 1] = make_float4(fuzz, m.ir, s.radius, idbits);
    }
  }
  rt_scene* s = new rt_scene();
  s->device = ctx().device;
  if (n_spheres) s->host_spheres.assign(spheres, spheres + n_spheres);
  if (n_materials) s->host_materials.assign(materials, materials + n_materials);
  ScopedScene guard;
  guard.s = s;
  RT_HIP(hipMalloc(&s->trig, trig.size() * sizeof(float4)));
  RT_HIP(hipMalloc(&s->geom, geom.size() * sizeof(float4)));
  RT_HIP(hipMalloc(&s->surf, surf.size() * sizeof(float4)));
  RT_HIP(hipMemcpy(s->trig, trig.data(), trig.size() * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(s->geom, geom.data(), geom.size() * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(s->surf, surf.data(), surf.size() * sizeof(float4), hipMemcpyHostToDevice));
  memset(&s->dev, 0, sizeof s->dev);
  s->dev.trig = (const float4*)s->trig;
  s->dev.geom = (const float4*)s->geom;
  s->dev.surf = (const float4*)s->surf;
  s->dev.n_spheres = n_spheres;
  s->dev.n_materials = n_materials;
  s->dev.bvh_info = rt::BvhInfo();   // the hierarchy is built on first use (ensure_bvh)
  MfmaScene ms;
  build_mfma_scene(spheres, n_spheres, &ms);
  if (ms.ok) {
    RT_HIP(hipMalloc(&s->mfma_rows, ms.rows.size() * sizeof(uint16_t)));
    RT_HIP(hipMalloc(&s->mfma_geom, ms.geom.size() * sizeof(float4)));
    RT_HIP(hipMalloc(&s->mfma_index, ms.index.size() * sizeof(uint32_t)));
    RT_HIP(hipMemcpy(s->mfma_rows, ms.rows.data(), ms.rows.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(s->mfma_geom, ms.geom.data(), ms.geom.size() * sizeof(float4), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(s->mfma_index, ms.index.data(), ms.index.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    s->dev.mfma_rows = (const uint4*)s->mfma_rows;
    s->dev.mfma_geom = (const float4*)s->mfma_geom;
    s->dev.mfma_index = (const uint32_t*)s->mfma_index;
    s->dev.mfma_ok = 1;
    s->dev.mfma_always = ms.n_always;
    s->dev.mfma_scale = ms.scale;
  }
  guard.s = nullptr;
  *out_scene = s;
  return RT_OK;
}

int rt_scene_destroy(rt_scene* scene) {
  if (!scene) return RT_OK;
  ApiScope scope;
  if (scene->device >= 0) (void)hipSetDevice(scene->device);
  if (scene->trig) (void)hipFree(scene->trig);
  if (scene->geom) (void)hipFree(scene->geom);
  if (scene->surf) (void)hipFree(scene->surf);
  if (scene->mfma_rows) (void)hipFree(scene->mfma_rows);
  if (scene->mfma_geom) (void)hipFree(scene->mfma_geom);
  if (scene->mfma_index) (void)hipFree(scene->mfma_index);
  if (scene->bvh) (void)hipFree(scene->bvh);
  if (scene->leaf_geom) (void)hipFree(scene->leaf_geom);
  if (scene->leaf_index) (void)hipFree(scene->leaf_index);
  delete scene;
  return RT_OK;
}

namespace {
int tile_table(const rt_frame* frame, const rt_tile* tiles, const uint64_t* out_offset,
               const uint32_t* out_stride, uint32_t n_tiles, std::vector<rt::TileDev>* out_td,
               uint64_t* out_pixels_total) {
  if (int rc = check_tiles(tiles, n_tiles, frame->image_width, frame->image_height)) return rc;
  std::vector<rt::TileDev>& td = *out_td;
  td.reserve(n_tiles);
  uint64_t first = 0, out_pixels = 0;
  for (uint32_t i = 0; i < n_tiles; ++i) {
    if (tiles[i].width == 0 || tiles[i].height == 0) continue;
    if (out_stride[i] < tiles[i].width)
      return fail(RT_ERR_INVALID_ARGUMENT, "tile %u: out_stride %u < width %u", i, out_stride[i],
                  tiles[i].width);
    rt::TileDev t;
    memset(&t, 0, sizeof t);
    t.x = tiles[i].x; t.y = tiles[i].y; t.h = tiles[i].height;
    rt::set_tile_width(t, tiles[i].width);
    t.first_pixel = (uint32_t)first;
    t.out_stride = out_stride[i];
    t.out_offset = out_offset[i];
    first += (uint64_t)t.w * t.h;
    const uint64_t end = t.out_offset + (uint64_t)(t.h - 1) * t.out_stride + t.w;
    if (end > out_pixels) out_pixels = end;
    td.push_back(t);
  }
  *out_pixels_total = out_pixels;
  return RT_OK;
}

void rows_tile(const rt_frame* frame, uint32_t row_begin, uint32_t row_end, rt::TileDev* out) {
  rt::TileDev& t = *out;
  memset(&t, 0, sizeof t);
  t.x = 0; t.y = row_begin; t.h = row_end - row_begin;
  rt::set_tile_width(t, frame->image_width);
  t.first_pixel = 0;
  t.out_stride = frame->image_width;
  t.out_offset = 0;
}

int check_rows(const rt_scene* scene, const rt_frame* frame, uint32_t row_begin, uint32_t row_end) {
  if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if (int rc = check_frame(frame)) return rc;
  if (row_begin > row_end || row_end > frame->image_height)
    return fail(RT_ERR_INVALID_ARGUMENT, "rows [%u,%u) of %u", row_begin, row_end,
                frame->image_height);
  return RT_OK;
}
}  // namespace

int rt_render_device_async(const rt_scene* scene, const rt_frame* frame, const rt_tile* tiles,
                           const uint64_t* out_offset, const uint32_t* out_stride,
                           uint32_t n_tiles, int resolve, float* d_out, void* stream) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if (int rc = check_frame(frame)) return rc;
  if (n_tiles == 0) return RT_OK;
  if (!tiles || !out_offset || !out_stride || !d_out)
    return fail(RT_ERR_INVALID_ARGUMENT, "NULL tile table or output");
  if (int rc = check_device_pointer(d_out, "d_out")) return rc;
  std::vector<rt::TileDev> td;
  uint64_t out_pixels = 0;
  if (int rc = tile_table(frame, tiles, out_offset, out_stride, n_tiles, &td, &out_pixels)) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : ctx().stream;
  return render_tiles_device(scene, frame, td.data(), td.size(), out_pixels, resolve, d_out, s);
}

int rt_accumulate_device_async(const rt_scene* scene, const rt_frame* frame, uint32_t sample_begin,
                               const rt_tile* tiles, const uint64_t* out_offset,
                               const uint32_t* out_stride, uint32_t n_tiles, float* d_sums,
                               void* stream) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if (int rc = check_frame(frame)) return rc;
  if ((uint64_t)sample_begin + frame->samples_per_pixel > 0xFFFFFFFFull)
    return fail(RT_ERR_LIMIT, "sample index beyond 2^32");
  if (n_tiles == 0) return RT_OK;
  if (!tiles || !out_offset || !out_stride || !d_sums)
    return fail(RT_ERR_INVALID_ARGUMENT, "NULL tile table or output");
  if (int rc = check_device_pointer(d_sums, "d_sums")) return rc;
  std::vector<rt::TileDev> td;
  uint64_t out_pixels = 0;
  if (int rc = tile_table(frame, tiles, out_offset, out_stride, n_tiles, &td, &out_pixels)) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : ctx().stream;
  return render_tiles_device(scene, frame, td.data(), td.size(), out_pixels, 0, d_sums, s,
                             sample_begin, true);
}

int rt_resolve_device_async(const float* d_sums, uint64_t n_pixels, uint32_t samples_per_pixel,
                            float* d_rgb, void* stream) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (n_pixels == 0) return RT_OK;
  if (!d_sums || !d_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (samples_per_pixel == 0) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel is 0");
  rt::ResolveParams rp;
  rp.sums = d_sums;
  rp.out = d_rgb;
  rp.n_pixels = n_pixels;
  rp.inv_spp = 1.0f / (float)samples_per_pixel;   // `1.0 / spp as f32`, src/cpu/mod.rs:67
  RT_HIP(rt::launch_resolve(rp, stream ? (hipStream_t)stream : ctx().stream));
  return RT_OK;
}

int rt_render_rows_device_async(const rt_scene* scene, const rt_frame* frame, uint32_t row_begin,
                                uint32_t row_end, int resolve, float* d_out, void* stream) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (int rc = check_rows(scene, frame, row_begin, row_end)) return rc;
  if (row_begin == row_end) return RT_OK;
  if (!d_out) return fail(RT_ERR_INVALID_ARGUMENT, "d_out is NULL");
  if (int rc = check_device_pointer(d_out, "d_out")) return rc;
  rt::TileDev t;
  rows_tile(frame, row_begin, row_end, &t);
  hipStream_t s = stream ? (hipStream_t)stream : ctx().stream;
  return render_tiles_device(scene, frame, &t, 1, (uint64_t)t.w * t.h, resolve, d_out, s);
}

int rt_render_rows_device_noalloc(const rt_scene* scene, const rt_frame* frame, uint32_t row_begin,
                                  uint32_t row_end, int resolve, float* d_out, void* stream) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (int rc = check_rows(scene, frame, row_begin, row_end)) return rc;
  if (row_begin == row_end) return RT_OK;
  if (!d_out) return fail(RT_ERR_INVALID_ARGUMENT, "d_out is NULL");
  if (!ctx().scratch) return fail(RT_ERR_NOT_INITIALIZED, "the fixed buffer has no address");
  rt::TileDev t;
  rows_tile(frame, row_begin, row_end, &t);
  hipStream_t s = stream ? (hipStream_t)stream : ctx().stream;
  return render_tiles_device(scene, frame, &t, 1, (uint64_t)t.w * t.h, resolve, d_out, s, 0, false,
                             /*no_alloc=*/true);
}

int rt_synchronize(void* stream) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  RT_HIP(hipStreamSynchronize(stream ? (hipStream_t)stream : ctx().stream));
  return RT_OK;
}

int rt_get_stats(rt_stats* out) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
  if (!ctx().last_slot) {
    *out = rt_stats{};
    return RT_OK;
  }
  const LaunchSlot& sl = *ctx().last_slot;
  RT_HIP(hipStreamSynchronize(sl.stream));
  rt_stats s = sl.last;
  if (s.workgroups != 0) {
    rt::Counters c;
    RT_HIP(hipMemcpy(&c, sl.counters, sizeof c, hipMemcpyDeviceToHost));
    s.rays = c.rays;
    s.samples = c.samples;
    s.sphere_tests = c.rays * sl.last.sphere_tests;
    // the event pair of a launch is reused by the launch kEventRing launches later
    s.kernel_ms = std::nanf("");
    if (sl.launch_no != 0 && ctx().launches - sl.launch_no < (uint64_t)kEventRing) {
      const int ev = (int)((sl.launch_no - 1) % kEventRing);
      float ms = 0.0f;
      RT_HIP(hipEventElapsedTime(&ms, ctx().ev_begin[ev], ctx().ev_end[ev]));
      s.kernel_ms = ms;
    }
  }
  *out = s;
  return RT_OK;
}

#ifdef RT_DIAG_DRAIN
// Developer build only (not part of the ABI): the record every wave of the last launch left at
// the start of its ring (see the epilogue of trace_kernel), 16 words per wave.
extern "C" int rt_debug_wave_log(uint64_t* out, uint32_t max_waves, uint32_t* n_waves) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!ctx().last_slot || !ctx().last_slot->ring) return fail(RT_ERR_INVALID_ARGUMENT, "no sliced launch yet");
  const LaunchSlot& sl = *ctx().last_slot;
  RT_HIP(hipStreamSynchronize(sl.stream));
  const uint32_t waves = sl.last.workgroups * (sl.last.threads_per_workgroup / rt::kWave);
  const size_t stride = sl.ring_stride;   // floats per wave
  *n_waves = waves < max_waves ? waves : max_waves;
  for (uint32_t w = 0; w < *n_waves; ++w)
    RT_HIP(hipMemcpy(out + 16 * (size_t)w, sl.ring + stride * w, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RT_OK;
}
#endif

int rt_get_phase_cycles(uint64_t* out16) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!out16) return fail(RT_ERR_INVALID_ARGUMENT, "out16 is NULL");
  rt::Counters c;
  memset(&c, 0, sizeof c);
  if (ctx().last_slot) {
    const LaunchSlot& sl = *ctx().last_slot;
    RT_HIP(hipStreamSynchronize(sl.stream));
    RT_HIP(hipMemcpy(&c, sl.counters, sizeof c, hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < 16; ++i) out16[i] = c.phase[i];
#ifdef RT_DIAG_DRAIN
  // developer build: slots 0-3 carry the phases' wave-cycles after the waves ran out of work
  for (int i = 0; i < 4; ++i) out16[i] = c.phase[16 + i];
  // slots 8-11: first and last wave out of work, last wave's end, counted from the first wave's start
  {
    const uint64_t start = ~c.phase[20];
    out16[8] = ~c.phase[21] - start;
    out16[9] = c.phase[22] - start;
    out16[10] = c.phase[23] - start;
    out16[11] = c.phase[15];   // most steps a wave took after it ran out of work
  }
#endif
#ifdef RT_DIAG_RING
  // developer build: slots 8-11 carry the waves' open-pixel statistics instead of drain steps
  // (steps starved of slots, path slots idle in them, merges, pixels merged)
  for (int i = 0; i < 4; ++i) out16[8 + i] = c.phase[16 + i];
#endif
#ifdef RT_DIAG_TRIGGERS
  // developer build: slots 8-10 carry the scan's branch statistics instead of drain steps
  for (int i = 0; i < 3; ++i) out16[8 + i] = c.phase[16 + i];
  out16[11] = c.phase[19];   // wave-cycles inside the matrix scan's candidate code
#endif
  return RT_OK;
}

int rt_get_launch_ms(uint32_t n_last, float* out_ms) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (n_last && !out_ms) return fail(RT_ERR_INVALID_ARGUMENT, "out_ms is NULL");
  uint64_t n = n_last;
  if (n > ctx().launches) n = ctx().launches;
  if (n > (uint64_t)kEventRing) n = kEventRing;
  for (uint64_t k = 0; k < n; ++k) {
    const int slot = (int)((ctx().launches - n + k) % kEventRing);
    RT_HIP(hipEventSynchronize(ctx().ev_end[slot]));
    float ms = 0.0f;
    RT_HIP(hipEventElapsedTime(&ms, ctx().ev_begin[slot], ctx().ev_end[slot]));
    out_ms[k] = ms;
  }
  return (int)n;
}

namespace {

// The device copy of the request's World on the context in charge: the one kept from the last
// call when the request names the same spheres and materials, a fresh upload otherwise.
int cached_scene(const rt_request* req, const rt_scene** out) {
  FrameCache& fc = ctx().cache;
  if (fc.scene && fc.scene->host_spheres.size() == req->n_spheres &&
      fc.scene->host_materials.size() == req->n_materials &&
      (req->n_spheres == 0 ||
       memcmp(fc.scene->host_spheres.data(), req->spheres, req->n_spheres * sizeof(rt_sphere)) == 0) &&
      (req->n_materials == 0 ||
       memcmp(fc.scene->host_materials.data(), req->materials,
              req->n_materials * sizeof(rt_material)) == 0)) {
    *out = fc.scene;
    return RT_OK;
  }
  if (fc.scene) {
    RT_HIP(hipDeviceSynchronize());   // nothing may still be reading the old copy
    (void)rt_scene_destroy(fc.scene);
    fc.scene = nullptr;
  }
  if (int rc = rt_scene_create(req->spheres, req->n_spheres, req->materials, req->n_materials,
                               &fc.scene))
    return rc;
  *out = fc.scene;
  return RT_OK;
}

int cached_frame(size_t floats) {
  FrameCache& fc = ctx().cache;
  if (floats > fc.frame_cap) {
    RT_HIP(hipStreamSynchronize(ctx().stream));
    if (fc.frame) RT_HIP(hipFree(fc.frame));
    fc.frame = nullptr;
    fc.frame_cap = 0;
    RT_HIP(hipMalloc((void**)&fc.frame, floats * sizeof(float)));
    fc.frame_cap = floats;
  }
  return RT_OK;
}

int cached_pinned(size_t floats) {
  FrameCache& fc = ctx().cache;
  if (floats > fc.pinned_cap) {
    RT_HIP(hipStreamSynchronize(ctx().stream));
    if (fc.pinned) RT_HIP(hipHostFree(fc.pinned));
    fc.pinned = nullptr;
    fc.pinned_cap = 0;
    // mapped and coherent: the kernels store into it directly (the device sees it at this address)
    RT_HIP(hipHostMalloc((void**)&fc.pinned, floats * sizeof(float), hipHostMallocDefault));
    fc.pinned_cap = floats;
  }
  return RT_OK;
}

// One device's part of a host-pointer render.  `tiles` address a compact buffer of `pixels`
// pixels (out_offset / out_stride); `dst[i]` is where tile i's rows go in the caller's frame
// (pixel index of its first pixel; rows there are image_width apart).
struct Share {
  std::vector<rt::TileDev> tiles;
  std::vector<uint64_t> dst;
  uint64_t pixels = 0;
  void add(uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t image_width) {
    rt::TileDev t;
    memset(&t, 0, sizeof t);
    t.x = x; t.y = y; t.h = h;
    rt::set_tile_width(t, w);
    t.first_pixel = (uint32_t)pixels;
    t.out_stride = w;
    t.out_offset = pixels;
    pixels += (uint64_t)w * h;
    tiles.push_back(t);
    dst.push_back((uint64_t)y * image_width + x);
  }
};

// Enqueue a share on the context in charge: the kernels write the resolved pixels straight into
// the page-locked host buffer (which the device addresses like its own memory), so the frame
// crosses PCIe once, pushed by the kernel that produces it, and no copy-engine pass follows.
// Returns without waiting, so several devices can be given their shares one after the other.
int enqueue_share(const rt_request* req, const rt_frame& f, const Share& sh) {
  if (sh.pixels == 0) return RT_OK;
  const rt_scene* scene = nullptr;
  if (int rc = cached_scene(req, &scene)) return rc;
  if (int rc = cached_pinned((size_t)sh.pixels * 3)) return rc;
  FrameCache& fc = ctx().cache;
  if (int rc = render_tiles_device(scene, &f, sh.tiles.data(), sh.tiles.size(), sh.pixels, 1,
                                   fc.pinned, ctx().stream))
    return rc;
  RT_HIP(hipEventRecord(fc.done, ctx().stream));
  return RT_OK;
}

// A large copy out of the pinned buffer, shared by a few threads: one core moves about 10 GB/s
// into pageable memory, and a 1080p frame is 25 MB.
void copy_out(float* dst, const float* src, size_t floats) {
  constexpr size_t kPerThread = 1u << 19;   // floats (2 MiB) a thread should at least have
  size_t parts = floats / kPerThread;
  if (parts > 6) parts = 6;
  if (parts < 2) {
    memcpy(dst, src, floats * sizeof(float));
    return;
  }
  std::thread helpers[5];
  const size_t each = floats / parts;
  for (size_t k = 1; k < parts; ++k) {
    const size_t b = k * each, e = k + 1 == parts ? floats : b + each;
    helpers[k - 1] = std::thread([=] { memcpy(dst + b, src + b, (e - b) * sizeof(float)); });
  }
  memcpy(dst, src, each * sizeof(float));
  for (size_t k = 1; k < parts; ++k) helpers[k - 1].join();
}

// Wait for the share and scatter its tile rows from the pinned buffer into the caller's frame.
int collect_share(const Share& sh, uint32_t image_width, float* out_rgb) {
  if (sh.pixels == 0) return RT_OK;
  FrameCache& fc = ctx().cache;
  RT_HIP(hipEventSynchronize(fc.done));
  size_t ti = 0;
  while (ti < sh.tiles.size()) {
    const rt::TileDev& t = sh.tiles[ti];
    const size_t src = (size_t)t.out_offset * 3;
    if (t.w == image_width) {
      // full-width tiles that follow each other in the frame and in the buffer: one copy
      size_t tj = ti + 1;
      uint64_t rows = t.h;
      while (tj < sh.tiles.size() && sh.tiles[tj].w == image_width &&
             sh.dst[tj] == sh.dst[ti] + rows * image_width &&
             sh.tiles[tj].out_offset == t.out_offset + rows * image_width) {
        rows += sh.tiles[tj].h;
        ++tj;
      }
      copy_out(out_rgb + (size_t)sh.dst[ti] * 3, fc.pinned + src, (size_t)rows * image_width * 3);
      ti = tj;
    } else {
      for (uint32_t row = 0; row < t.h; ++row)
        memcpy(out_rgb + ((size_t)sh.dst[ti] + (size_t)row * image_width) * 3,
               fc.pinned + src + (size_t)row * t.w * 3, (size_t)t.w * 3 * sizeof(float));
      ++ti;
    }
  }
  return RT_OK;
}

// Does the request's tile list cover every pixel of the image?  (Overlaps do not matter: a
// pixel's value does not depend on the tile that renders it.)
bool covers_frame(const rt_tile* tiles, uint32_t n, uint32_t w, uint32_t h) {
  uint64_t area = 0;
  for (uint32_t i = 0; i < n; ++i) area += (uint64_t)tiles[i].width * tiles[i].height;
  if (area < (uint64_t)w * h) return false;
  std::vector<uint8_t> seen((size_t)w * h, 0);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t y = 0; y < tiles[i].height; ++y)
      if (tiles[i].width)
        memset(&seen[(size_t)(tiles[i].y + y) * w + tiles[i].x], 1, tiles[i].width);
  return memchr(seen.data(), 0, seen.size()) == nullptr;
}

// Puts the primary context back in charge when a multi-device call leaves, however it leaves.
struct ContextGuard {
  ~ContextGuard() { g_cur = &g_primary; }
};

// rt_render and rt_render_multi.  devices == nullptr: the primary context renders everything.
int render_host(const rt_request* req, const int* devices, int n_devices, float* out_rgb) {
  if (int rc = check_ready()) return rc;
  if (int rc = check_request(req, true)) return rc;
  if (!out_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "out_rgb is NULL");
  rt_frame f = frame_of(req);
  if (int rc = check_frame(&f)) return rc;
  const uint32_t W = f.image_width, H = f.image_height;
  // The reference renders exactly the queued tiles and leaves every other pixel (0,0,0)
  // (src/cpu/mod.rs:57-80).  No tile list, or one that covers the image, is the whole frame.
  const bool listed = req->tiles != nullptr && req->n_tiles != 0;
  if (listed)
    if (int rc = check_tiles(req->tiles, req->n_tiles, W, H)) return rc;
  const bool whole = !listed || covers_frame(req->tiles, req->n_tiles, W, H);
  if (!whole) memset(out_rgb, 0, (size_t)W * H * 3 * sizeof(float));

  constexpr uint32_t kBlockRows = 8;   // rows are dealt to the devices in blocks of 8, round-robin
  const int n = devices ? n_devices : 1;
  std::vector<Share> shares(n);
  if (!devices) {
    if (whole) {
      shares[0].add(0, 0, W, H, W);
    } else {
      for (uint32_t i = 0; i < req->n_tiles; ++i)
        if (req->tiles[i].width && req->tiles[i].height)
          shares[0].add(req->tiles[i].x, req->tiles[i].y, req->tiles[i].width, req->tiles[i].height, W);
    }
  } else if (whole) {
    const uint32_t n_blocks = (H + kBlockRows - 1) / kBlockRows;
    for (uint32_t b = 0; b < n_blocks; ++b) {
      const uint32_t y = b * kBlockRows;
      shares[b % n].add(0, y, W, H - y < kBlockRows ? H - y : kBlockRows, W);
    }
  } else {
    // a partial queue: every tile is cut along the same 8-row blocks, each piece going to the
    // device that owns its block
    for (uint32_t i = 0; i < req->n_tiles; ++i) {
      const rt_tile& t = req->tiles[i];
      if (!t.width || !t.height) continue;
      for (uint32_t y = t.y; y < t.y + t.height;) {
        const uint32_t b = y / kBlockRows;
        uint32_t end = (b + 1) * kBlockRows;
        if (end > t.y + t.height) end = t.y + t.height;
        shares[b % n].add(t.x, y, t.width, end - y, W);
        y = end;
      }
    }
  }

  ContextGuard guard;
  // enqueue every device's share; nothing in this loop waits, so the GPUs run side by side
  for (int k = 0; k < n; ++k) {
    if (devices) {
      while ((int)g_extra.size() <= k) g_extra.push_back(new State());
      g_cur = g_extra[k];
      if (ctx().ready && ctx().device != devices[k]) destroy_context();
      if (!ctx().ready)
        if (int rc = init_context(devices[k])) return rc;
    }
    RT_HIP(hipSetDevice(ctx().device));
    if (int rc = enqueue_share(req, f, shares[k])) return rc;
  }
  for (int k = 0; k < n; ++k) {
    if (devices) g_cur = g_extra[k];
    RT_HIP(hipSetDevice(ctx().device));
    if (int rc = collect_share(shares[k], W, out_rgb)) return rc;
  }
  return RT_OK;
}

}  // namespace

int rt_render(const rt_request* req, float* out_rgb) {
  ApiScope scope;
  return render_host(req, nullptr, 0, out_rgb);
}

int rt_render_multi(const rt_request* req, const int* devices, int n_devices, float* out_rgb) {
  ApiScope scope;
  if (!devices || n_devices < 1 || n_devices > 64)
    return fail(RT_ERR_INVALID_ARGUMENT, "device list of %d entries", n_devices);
  return render_host(req, devices, n_devices, out_rgb);
}

namespace {
// rt_render_tiles, and rt_accumulate_tiles when `inout` (the sums are uploaded first and the
// call adds samples [sample_begin, sample_begin + spp) to them).
int render_tiles_host(const rt_request* req, float* out_sums, bool inout, uint32_t sample_begin) {
  if (int rc = check_ready()) return rc;
  if (int rc = check_request(req, true)) return rc;
  rt_frame f = frame_of(req);
  if (int rc = check_frame(&f)) return rc;
  if (req->n_tiles == 0) return RT_OK;
  if (!out_sums) return fail(RT_ERR_INVALID_ARGUMENT, "out_sums is NULL");
  std::vector<uint64_t> off(req->n_tiles);
  std::vector<uint32_t> stride(req->n_tiles);
  uint64_t total = 0;
  for (uint32_t i = 0; i < req->n_tiles; ++i) {
    off[i] = total;
    stride[i] = req->tiles[i].width;
    total += (uint64_t)req->tiles[i].width * req->tiles[i].height;
  }
  if (total == 0) return RT_OK;
  const rt_scene* scene = nullptr;
  if (int rc = cached_scene(req, &scene)) return rc;
  const size_t bytes = (size_t)total * 3 * sizeof(float);
  if (int rc = cached_frame((size_t)total * 3)) return rc;
  float* d = ctx().cache.frame;
  if (inout) {
    RT_HIP(hipMemcpyAsync(d, out_sums, bytes, hipMemcpyHostToDevice, ctx().stream));
    if (int rc = rt_accumulate_device_async(scene, &f, sample_begin, req->tiles, off.data(),
                                            stride.data(), req->n_tiles, d, ctx().stream))
      return rc;
  } else if (int rc = rt_render_device_async(scene, &f, req->tiles, off.data(), stride.data(),
                                             req->n_tiles, 0, d, ctx().stream)) {
    return rc;
  }
  RT_HIP(hipMemcpyAsync(out_sums, d, bytes, hipMemcpyDeviceToHost, ctx().stream));
  RT_HIP(hipStreamSynchronize(ctx().stream));
  return RT_OK;
}
}  // namespace

int rt_render_tiles(const rt_request* req, float* out_sums) {
  ApiScope scope;
  return render_tiles_host(req, out_sums, false, 0);
}

int rt_accumulate_tiles(const rt_request* req, uint32_t sample_begin, float* inout_sums) {
  ApiScope scope;
  return render_tiles_host(req, inout_sums, true, sample_begin);
}

int rt_compose(const rt_tile* tiles, uint32_t n_tiles, const float* tile_sums, uint32_t image_width,
               uint32_t image_height, uint32_t samples_per_pixel, float* out_rgb) {
  if (!out_rgb || (n_tiles && (!tiles || !tile_sums)))
    return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (samples_per_pixel == 0) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel is 0");
  if (int rc = check_tiles(tiles, n_tiles, image_width, image_height)) return rc;
  // compose_final_framebuffer, src/cpu/mod.rs:51-83 (host arithmetic; a 25 MB scatter is not
  // worth a round trip to the device when the sums are already on the host)
  memset(out_rgb, 0, (size_t)image_width * image_height * 3 * sizeof(float));
  const float scale = 1.0f / (float)samples_per_pixel;
  size_t base = 0;
  for (uint32_t j = 0; j < n_tiles; ++j) {
    const rt_tile& t = tiles[j];
    for (uint32_t yoff = 0; yoff < t.height; ++yoff) {
      for (uint32_t xoff = 0; xoff < t.width; ++xoff) {
        const float* c = tile_sums + 3 * (base + xoff + (size_t)yoff * t.width);
        float* o = out_rgb + 3 * ((size_t)(t.x + xoff) + (size_t)(t.y + yoff) * image_width);
        for (int k = 0; k < 3; ++k) {
          float v = std::sqrt(c[k] * scale);
          if (v < 0.0f) v = 0.0f; else if (v > 0.999f) v = 0.999f;   // NaN stays NaN
          o[k] = v;
        }
      }
    }
    base += (size_t)t.width * t.height;
  }
  return RT_OK;
}

int rt_scene_scan_info(const rt_scene* scene, rt_scan_info* out) {
  if (!scene || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL scene or out");
  out->matrix_filter = scene->dev.mfma_ok;
  out->listed_spheres = scene->dev.mfma_always;
  out->scale = scene->dev.mfma_ok ? scene->dev.mfma_scale : 1.0f;
  out->origin_limit = scene->dev.mfma_ok ? 46340.0f / scene->dev.mfma_scale : 0.0f;   // sqrt(2^31)
  return RT_OK;
}

int rt_trace_rays(const rt_scene* scene, const rt_ray_in* rays, uint32_t n_rays, uint32_t seed,
                  uint32_t flags, rt_ray_out* out) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if (n_rays == 0) return RT_OK;
  if (!rays || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL rays or out");
  if (scene->device != ctx().device)
    return fail(RT_ERR_INVALID_ARGUMENT, "scene lives on device %d, the library on device %d",
                scene->device, ctx().device);
  const bool accel = (flags & RT_FLAG_ACCEL) != 0;
  if (accel) {
    if (int rc = ensure_bvh(scene)) return rc;
    if (scene->bvh_depth > rt::bvh_stack_depth())
      return fail(RT_ERR_LIMIT, "hierarchy of %u levels exceeds the traversal stack (%u)",
                  scene->bvh_depth, rt::bvh_stack_depth());
  }
  ScopedDev din, dout;
  RT_HIP(hipMalloc(&din.p, (size_t)n_rays * sizeof(rt_ray_in)));
  RT_HIP(hipMalloc(&dout.p, (size_t)n_rays * sizeof(rt_ray_out)));
  RT_HIP(hipMemcpy(din.p, rays, (size_t)n_rays * sizeof(rt_ray_in), hipMemcpyHostToDevice));
  rt::HookParams hp;
  hp.scene = scene->dev;
  hp.rays = (const rt_ray_in*)din.p;
  hp.out = (rt_ray_out*)dout.p;
  hp.n_rays = n_rays;
  hp.seed = seed;
  hp.accel = accel ? 1 : 0;
  hp.mfma = !accel && !(flags & (RT_FLAG_STRICT_FP | RT_FLAG_SCAN_VALU)) && scene->dev.mfma_ok ? 1 : 0;
  RT_HIP(rt::launch_hook(hp, (flags & RT_FLAG_STRICT_FP) != 0, ctx().stream));
  RT_HIP(hipStreamSynchronize(ctx().stream));
  RT_HIP(hipMemcpy(out, dout.p, (size_t)n_rays * sizeof(rt_ray_out), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_selftest_sqrt(uint64_t* out4) {
  ApiScope scope;
  if (int rc = check_ready()) return rc;
  if (!out4) return fail(RT_ERR_INVALID_ARGUMENT, "out4 is NULL");
  ScopedDev d;
  RT_HIP(hipMalloc(&d.p, 4 * sizeof(unsigned long long)));
  RT_HIP(hipMemsetAsync(d.p, 0, 4 * sizeof(unsigned long long), ctx().stream));
  RT_HIP(rt::launch_sqrt_check((unsigned long long*)d.p, ctx().stream));
  RT_HIP(hipStreamSynchronize(ctx().stream));
  unsigned long long h[4];
  RT_HIP(hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; ++i) out4[i] = h[i];
  return RT_OK;
}

}  // extern "C"
