// Drives the C++ host adapter (schwarzwald_amd/host/swz_tiling.hpp) the way the reference's (disabled)
// tiler integration tests intended (test/TestTiler.cpp:85-161, 361-421: internal_cache_size smaller than the
// data set, i.e. SEVERAL batches through one tiling algorithm object): a MemoryPersistence-like sink collects
// every node; afterwards every point must be stored exactly once (ACCURATE), lie inside its node's bounds, and
// the node files must equal the multi-batch oracle's.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <map>
#include <vector>

#include "../../oracle/oracle.h"
#include "../../schwarzwald_amd/host/swz_tiling.hpp"

using namespace swz_host;

struct MemorySink : PointsSink {  // cf. core/io/MemoryPersistence.h:14-52 (but replacing, like BinaryPersistence)
  std::map<std::string, std::vector<uint32_t>> nodes;
  std::map<std::string, std::vector<double>> positions;
  std::map<std::string, AABB> bounds;
  void persist_points(const uint32_t* b, const uint32_t* e, const double* xyz, const AABB& nb,
                      const std::string& name) override {
    nodes[name].assign(b, e);
    positions[name].assign(xyz, xyz + 3 * (e - b));
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
  const double mn[3] = {0, 0, 0}, mx[3] = {1, 1, 1};
  const float spacing = (float)(std::sqrt(3.0) / 250.0);
  const char* names[] = {"RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED"};
  for (int s = 0; s < 4; ++s) {
    for (int batches = 1; batches <= 3; batches += 2) {
      MemorySink sink;
      TilerMetaParameters meta;
      meta.spacing_at_root = spacing;
      meta.max_points_per_node = 1000;
      TilingAlgorithmGPU tiler(make_sampling_strategy_from_name(names[s], 1000), sink, meta);
      if (batches == 3) tiler.set_export_chunk_points(2500);  // finalize() hands the files over in many small chunks of whole nodes
      orc_tile_params p{s, 1000, spacing, 100, ORC_ACCURATE, 8};
      orc_tiler* oracle = orc_tiler_create(mn, mx, &p);
      for (int b = 0; b < batches; ++b) {
        const size_t lo = n * b / batches, hi = n * (b + 1) / batches;
        tiler.tile_batch(xyz.data() + 3 * lo, hi - lo, bounds);
        std::vector<double> copy(xyz.begin() + 3 * lo, xyz.begin() + 3 * hi);
        if (orc_tiler_add_batch(oracle, copy.data(), hi - lo) != 0) return fail("oracle status");
      }
      const size_t persisted = tiler.finalize(bounds);
      if (orc_tiler_finalize(oracle) != 0) return fail("oracle finalize");
      if (persisted != sink.nodes.size()) return fail("node count");
      // every point exactly once, inside its node, with its own position
      std::vector<int> seen(n, 0);
      for (const auto& kv : sink.nodes) {
        const AABB& b = sink.bounds[kv.first];
        const std::vector<double>& pos = sink.positions[kv.first];
        for (size_t q = 0; q < kv.second.size(); ++q) {
          const uint32_t i = kv.second[q];
          ++seen[i];
          const double x = pos[3 * q], y = pos[3 * q + 1], z = pos[3 * q + 2];
          if (x != xyz[3 * i] || y != xyz[3 * i + 1] || z != xyz[3 * i + 2]) return fail("gathered position differs");
          if (!(x >= b.min.x && x <= b.max.x && y >= b.min.y && y <= b.max.y && z >= b.min.z && z <= b.max.z))
            return fail("point outside node bounds");
        }
      }
      for (size_t i = 0; i < n; ++i)
        if (seen[i] != 1) return fail("point not stored exactly once");
      // oracle: same node files, same order
      uint64_t nn = 0, ns = 0;
      orc_tiler_counts(oracle, &nn, &ns, nullptr, nullptr);
      std::vector<int8_t> nl(nn);
      std::vector<uint64_t> nk(nn), no(nn), nc(nn);
      std::vector<uint32_t> ids(ns);
      orc_tiler_export(oracle, nl.data(), nk.data(), no.data(), nc.data(), ids.data(), nullptr);
      orc_tiler_destroy(oracle);
      std::map<std::string, std::vector<uint32_t>> expect;
      for (uint64_t j = 0; j < nn; ++j) {
        std::string name = "r";
        for (int l = 0; l <= nl[j]; ++l) name.push_back((char)('0' + get_octant_at_level(nk[j], (uint32_t)l)));
        expect[name].assign(ids.begin() + no[j], ids.begin() + no[j] + nc[j]);
      }
      if (expect != sink.nodes) return fail("node files differ from the oracle");
      std::printf("%-12s %d batch(es) ok: %zu nodes\n", names[s], batches, sink.nodes.size());
    }
  }
  return 0;
}
