// Host-only check of the hierarchy builder (ray_tracing_amd/csrc/rt_bvh.cpp): reads spheres
// (x y z r per line) from stdin, builds, and verifies the invariants the traversal relies on.
// Prints "ok <nodes> <depth> <n_big>" or a message and exits 1.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../ray_tracing_amd/csrc/rt_internal.h"

static int fail(const char* what, long a = 0, long b = 0) {
  std::printf("FAIL %s %ld %ld\n", what, a, b);
  return 1;
}

int main() {
  std::vector<rt_sphere> s;
  float x, y, z, r;
  while (std::scanf("%f %f %f %f", &x, &y, &z, &r) == 4) {
    rt_sphere sp = {x, y, z, r, 0};
    s.push_back(sp);
  }
  rt::BvhHost h;
  rt::build_bvh(s.data(), (uint32_t)s.size(), &h);
  const uint32_t n = (uint32_t)s.size();
  if (h.leaf_geom.size() != n || h.leaf_index.size() != n) return fail("leaf arrays", h.leaf_geom.size(), n);
  // the leaf arrays are a permutation of the spheres
  std::vector<int> seen(n, 0);
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t id = h.leaf_index[k];
    if (id >= n || seen[id]++) return fail("permutation", k, id);
    const rt_sphere& sp = s[id];
    if (h.leaf_geom[k].x != sp.x || h.leaf_geom[k].y != sp.y || h.leaf_geom[k].z != sp.z ||
        h.leaf_geom[k].w != sp.radius * sp.radius)
      return fail("leaf record", k, id);
  }
  if (h.info.n_nodes != h.nodes.size()) return fail("n_nodes", h.info.n_nodes, h.nodes.size());
  if (h.info.n_big > n) return fail("n_big", h.info.n_big, n);
  if (n - h.info.n_big > 0 && h.nodes.empty()) return fail("no tree for small spheres");
  // walk the tree: every slot beyond the big ones is reached exactly once, boxes hold their
  // spheres and their children, depth is what the builder reports
  std::vector<int> slot_seen(n, 0);
  uint32_t max_depth = 0;
  struct Item { uint32_t node, depth; };
  std::vector<Item> stack;
  if (!h.nodes.empty()) stack.push_back({0u, 1u});
  size_t visited = 0;
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    visited++;
    if (it.node >= h.nodes.size()) return fail("child index", it.node);
    const rt::BvhNode& nd = h.nodes[it.node];
    if (it.depth > max_depth) max_depth = it.depth;
    if (nd.b & rt::kBvhLeaf) {
      const uint32_t count = nd.b & ~rt::kBvhLeaf;
      if (count == 0 || count > 8) return fail("leaf size", count);
      for (uint32_t k = nd.a; k < nd.a + count; ++k) {
        if (k < h.info.n_big || k >= n || slot_seen[k]++) return fail("leaf slot", k);
        const rt_sphere& sp = s[h.leaf_index[k]];
        const float c[3] = {sp.x, sp.y, sp.z};
        for (int a = 0; a < 3; ++a)
          if (!(nd.lo[a] <= c[a] - std::fabs(sp.radius)) || !(nd.hi[a] >= c[a] + std::fabs(sp.radius)))
            return fail("sphere outside its leaf box", k, a);
      }
    } else {
      for (uint32_t child : {nd.a, nd.b}) {
        if (child >= h.nodes.size()) return fail("child index", child);
        for (int a = 0; a < 3; ++a)
          if (!(nd.lo[a] <= h.nodes[child].lo[a]) || !(nd.hi[a] >= h.nodes[child].hi[a]))
            return fail("child box outside parent", child, a);
        stack.push_back({child, it.depth + 1});
      }
    }
  }
  if (visited != h.nodes.size()) return fail("unreachable nodes", visited, h.nodes.size());
  for (uint32_t k = h.info.n_big; k < n; ++k)
    if (!slot_seen[k]) return fail("slot not in any leaf", k);
  if (max_depth != h.depth) return fail("depth", max_depth, h.depth);
  // the ball (centre, extent) holds every tree sphere's centre; r_min / r_max bracket the radii
  for (uint32_t k = h.info.n_big; k < n; ++k) {
    const rt_sphere& sp = s[h.leaf_index[k]];
    const double dx = sp.x - h.info.centre[0], dy = sp.y - h.info.centre[1], dz = sp.z - h.info.centre[2];
    if (std::sqrt(dx * dx + dy * dy + dz * dz) > h.info.extent) return fail("centre outside the ball", k);
    if (std::fabs(sp.radius) < h.info.r_min || std::fabs(sp.radius) > h.info.r_max) return fail("radius range", k);
  }
  std::printf("ok %zu %u %u\n", h.nodes.size(), h.depth, h.info.n_big);
  return 0;
}
