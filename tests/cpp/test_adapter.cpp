// Drives the C++ host adapter (schwarzwald_amd/host/swz_tiling.hpp) the way the reference's (disabled)
// tiler integration tests intended (test/TestTiler.cpp:113-161, 361-421): a MemoryPersistence-like sink
// collects every node; afterwards every point must be stored exactly once (ACCURATE), lie inside its
// node's bounds, and the per-node lists must equal the oracle's tiling.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <map>
#include <vector>

#include "../../oracle/oracle.h"
#include "../../schwarzwald_amd/host/swz_tiling.hpp"

using namespace swz_host;

struct MemorySink : PointsSink {  // cf. core/io/MemoryPersistence.h:14-52
  std::map<std::string, std::vector<uint32_t>> nodes;
  std::map<std::string, AABB> bounds;
  void persist_points(const uint32_t* b, const uint32_t* e, const AABB& nb, const std::string& name) override {
    nodes[name].assign(b, e);
    bounds[name] = nb;
  }
};

static int fail(const char* msg) {
  std::fprintf(stderr, "FAIL: %s\n", msg);
  return 1;
}

int main() {
  const size_t n = 120000;
  std::vector<double> xyz(n * 3);
  orc_generate_uniform(42, 0, n, xyz.data());
  const AABB bounds{{0, 0, 0}, {1, 1, 1}};
  const float spacing = (float)(std::sqrt(3.0) / 250.0);
  const char* names[] = {"RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED"};
  for (int s = 0; s < 4; ++s) {
    MemorySink sink;
    TilerMetaParameters meta;
    meta.spacing_at_root = spacing;
    meta.max_points_per_node = 1000;
    TilingAlgorithmGPU tiler(make_sampling_strategy_from_name(names[s], 1000), sink, meta);
    std::vector<double> pos = xyz;
    const auto res = tiler.tile_batch(pos.data(), n, bounds);
    if (res.nodes_persisted != sink.nodes.size() || sink.nodes.size() != res.stats.num_nodes) return fail("node count");
    // every point exactly once, inside its node
    std::vector<int> seen(n, 0);
    for (const auto& kv : sink.nodes) {
      const AABB& b = sink.bounds[kv.first];
      for (uint32_t i : kv.second) {
        ++seen[i];
        const double x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        if (!(x >= b.min.x && x <= b.max.x && y >= b.min.y && y <= b.max.y && z >= b.min.z && z <= b.max.z))
          return fail("point outside node bounds");
      }
    }
    for (size_t i = 0; i < n; ++i)
      if (seen[i] != 1) return fail("point not stored exactly once");
    // oracle: same per-node membership and order
    std::vector<double> opos = xyz;
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> perm(n);
    std::vector<int8_t> level(n);
    orc_tile_params p{s, 1000, spacing, 100, ORC_ACCURATE, 8};
    const double mn[3] = {0, 0, 0}, mx[3] = {1, 1, 1};
    if (orc_tile(opos.data(), n, mn, mx, &p, keys.data(), perm.data(), level.data(), nullptr, nullptr) != 0)
      return fail("oracle status");
    std::map<std::string, std::vector<uint32_t>> expect;
    for (size_t i = 0; i < n; ++i) {
      std::string name = "r";
      for (int l = 0; l <= level[i]; ++l) name.push_back((char)('0' + get_octant_at_level(keys[i], (uint32_t)l)));
      expect[name].push_back(perm[i]);
    }
    if (expect != sink.nodes) return fail("node lists differ from the oracle");
    std::printf("%-12s ok: %zu nodes\n", names[s], sink.nodes.size());
  }
  return 0;
}
