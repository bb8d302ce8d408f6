// rt_bvh.cpp - host-side build of the optional bounding-volume hierarchy over a World's spheres.
//
// The reference's World::hit is a linear scan (src/main.rs:175-189) and the default kernel keeps
// it that way.  RT_FLAG_ACCEL swaps the scan for a traversal of this hierarchy; every sphere the
// traversal reaches goes through the same exact Sphere::hit arithmetic and the same
// minimum-t / later-sphere-wins rule as the scan's replay, so the result is the same bits -
// provided the traversal never prunes a sphere the reference's arithmetic would accept.  The
// boxes stored here are the spheres' exact bounds; the kernel widens them per ray by a margin
// that covers the rounding of the reference's own discriminant and roots (bvh_margin in
// rt_kernels.hip).
//
// Layout: spheres much larger than the typical one (the ground, the three big spheres) would
// bloat every box above them, so they are kept out of the tree and tested for every ray
// (leaf slots [0, n_big)).  The rest is split top-down at the median centre along the longest
// axis of the centres' bounds until a node holds at most kLeafSpheres (8: measured against 1, 2 and 4); leaves index a
// permuted copy of (cx, cy, cz, r*r) with the original sphere numbers beside it.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "rt_internal.h"

namespace rt {

namespace {

#ifndef RT_BVH_LEAF
#define RT_BVH_LEAF 8
#endif
constexpr uint32_t kLeafSpheres = RT_BVH_LEAF;

struct Builder {
  const rt_sphere* s;
  std::vector<uint32_t> order;        // permutation under construction (tree part only)
  std::vector<BvhNode> nodes;
  uint32_t depth = 0;                 // deepest level reached (root = 1)

  void bounds(uint32_t first, uint32_t count, float lo[3], float hi[3]) const {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (uint32_t i = first; i < first + count; ++i) {
      const rt_sphere& sp = s[order[i]];
      const float c[3] = {sp.x, sp.y, sp.z};
      const float r = std::fabs(sp.radius);
      for (int k = 0; k < 3; ++k) {
        // one ulp outward: c - r and c + r round to nearest
        lo[k] = std::min(lo[k], std::nextafterf(c[k] - r, -INFINITY));
        hi[k] = std::max(hi[k], std::nextafterf(c[k] + r, INFINITY));
      }
    }
  }

  // Builds the subtree over order[first, first+count) into nodes[index].
  void build(uint32_t index, uint32_t first, uint32_t count, uint32_t leaf_base,
             uint32_t level = 1) {
    depth = std::max(depth, level);
    BvhNode n;
    memset(&n, 0, sizeof n);
    bounds(first, count, n.lo, n.hi);
    if (count <= kLeafSpheres) {
      n.a = leaf_base + first;
      n.b = count | kBvhLeaf;
      nodes[index] = n;
      return;
    }
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = first; i < first + count; ++i) {
      const rt_sphere& sp = s[order[i]];
      const float c[3] = {sp.x, sp.y, sp.z};
      for (int k = 0; k < 3; ++k) { clo[k] = std::min(clo[k], c[k]); chi[k] = std::max(chi[k], c[k]); }
    }
    int axis = 0;
    for (int k = 1; k < 3; ++k)
      if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    const uint32_t half = count / 2;
    auto key = [&](uint32_t id) { return axis == 0 ? s[id].x : axis == 1 ? s[id].y : s[id].z; };
    std::nth_element(order.begin() + first, order.begin() + first + half,
                     order.begin() + first + count,
                     [&](uint32_t p, uint32_t q) { return key(p) < key(q) || (key(p) == key(q) && p < q); });
    const uint32_t left = (uint32_t)nodes.size();
    nodes.push_back(BvhNode());
    nodes.push_back(BvhNode());
    n.a = left;
    n.b = left + 1;
    nodes[index] = n;
    build(left, first, half, leaf_base, level + 1);
    build(left + 1, first + half, count - half, leaf_base, level + 1);
  }
};

}  // namespace

void build_bvh(const rt_sphere* spheres, uint32_t n, BvhHost* out) {
  out->nodes.clear();
  out->leaf_geom.clear();
  out->leaf_index.clear();
  out->depth = 0;
  out->info = BvhInfo();
  if (n == 0) return;

  // "big" = radius above 4x the median radius
  std::vector<float> radii(n);
  for (uint32_t i = 0; i < n; ++i) radii[i] = std::fabs(spheres[i].radius);
  std::vector<float> sorted = radii;
  std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
  const float median = sorted[n / 2];
  std::vector<uint32_t> big, rest;
  for (uint32_t i = 0; i < n; ++i) (radii[i] > 4.0f * median ? big : rest).push_back(i);

  Builder b;
  b.s = spheres;
  b.order = rest;
  if (!rest.empty()) {
    b.nodes.push_back(BvhNode());
    b.build(0, 0, (uint32_t)rest.size(), (uint32_t)big.size());
  }

  // leaf arrays: big spheres first (storage order), then the tree's permutation
  std::vector<uint32_t> all = big;
  all.insert(all.end(), b.order.begin(), b.order.end());
  out->leaf_geom.resize(all.size());
  out->leaf_index = all;
  for (size_t k = 0; k < all.size(); ++k) {
    const rt_sphere& sp = spheres[all[k]];
    volatile float r2 = sp.radius * sp.radius;   // `self.radius * self.radius`, src/main.rs:134
    out->leaf_geom[k] = make_float4(sp.x, sp.y, sp.z, r2);
  }
  out->nodes = b.nodes;
  out->depth = b.depth;

  // what the per-ray margin needs: a ball that holds every tree sphere's centre, and the
  // smallest and largest radius in the tree (double arithmetic, rounded outward)
  BvhInfo info;
  info.n_big = (uint32_t)big.size();
  info.n_nodes = (uint32_t)b.nodes.size();
  if (!rest.empty()) {
    double c[3] = {0, 0, 0};
    for (uint32_t id : rest) { c[0] += spheres[id].x; c[1] += spheres[id].y; c[2] += spheres[id].z; }
    for (double& v : c) v /= (double)rest.size();
    double ext = 0, rmin = INFINITY, rmax = 0;
    for (uint32_t id : rest) {
      const double dx = spheres[id].x - c[0], dy = spheres[id].y - c[1], dz = spheres[id].z - c[2];
      ext = std::max(ext, std::sqrt(dx * dx + dy * dy + dz * dz));
      rmin = std::min(rmin, (double)radii[id]);
      rmax = std::max(rmax, (double)radii[id]);
    }
    info.centre[0] = (float)c[0]; info.centre[1] = (float)c[1]; info.centre[2] = (float)c[2];
    info.extent = std::nextafterf((float)(ext * 1.000001 + 1e-30), INFINITY);
    info.r_min = std::nextafterf((float)rmin, 0.0f);
    info.r_max = std::nextafterf((float)rmax, INFINITY);
  }
  out->info = info;
}

}  // namespace rt
