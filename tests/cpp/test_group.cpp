// One batch sharded over several contexts from ONE C++ process (swz_group_*, include/swz_gpu.h): the points start
// scattered over the shards in uneven pieces (one piece empty), the library exchanges them by level-0 octant and
// tiles; the union of the shards' results must equal the single-process oracle's result point for point
// (Morton key and node level of every position), every shard's keys ascend and lie in its own octants.
// Usage: test_group [transport]   (0 = peer copies, default; 1 = RCCL, one shard per visible GPU)
// Exit code 0 = pass.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/swz_gpu.h"
#include "../../oracle/oracle.h"
#include "../../schwarzwald_amd/host/swz_tiling.hpp"

static int fail(const char* msg, const char* detail = "") {
  std::fprintf(stderr, "FAIL: %s %s\n", msg, detail);
  return 1;
}

// collects the node files a ShardedTilingAlgorithmGPU hands over: name -> the input index of every row (GPS time column)
struct RowSink : swz_host::PointsSink {
  std::map<std::string, std::vector<uint32_t>> nodes;
  const std::vector<double>* xyz = nullptr;
  bool rows_match = true;
  void persist_points(const uint32_t*, const uint32_t*, const double*, const swz_host::AABB&, const std::string&) override {}
  void persist_rows(size_t count, const double* positions, const swz_attribute_columns& attrs, const swz_host::AABB& b,
                    const std::string& name) override {
    const double* gps = static_cast<const double*>(attrs.column[SWZ_ATTR_GPS_TIME]);
    std::vector<uint32_t>& ids = nodes[name];
    for (size_t i = 0; i < count; ++i) {
      const uint32_t src = (uint32_t)gps[i];
      ids.push_back(src);
      for (int d = 0; d < 3; ++d) rows_match &= positions[3 * i + d] == (*xyz)[3 * (size_t)src + d];
      rows_match &= positions[3 * i] >= b.min.x && positions[3 * i] <= b.max.x && positions[3 * i + 1] >= b.min.y &&
                    positions[3 * i + 1] <= b.max.y && positions[3 * i + 2] >= b.min.z && positions[3 * i + 2] <= b.max.z;
    }
  }
};

using Row = std::tuple<uint64_t, double, double, double, int>;  // key, x, y, z, level

// what the multi-batch oracle stores, by (sampler, strategy, batches): it does not depend on the number of shards
static std::map<std::tuple<int, int, int>, std::map<std::string, std::vector<uint32_t>>> g_expect;

int main(int argc, char** argv) {
  const int transport = argc > 1 ? std::atoi(argv[1]) : 0;
  const size_t n = 300000;
  const double mn[3] = {0, 0, 0}, mx[3] = {1, 1, 1};
  const float spacing = (float)(std::sqrt(3.0) / 250.0);
  const char* names[] = {"RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED"};
  std::vector<double> xyz(n * 3);
  orc_generate_uniform(7, 0, n, xyz.data());
  // two attribute columns that identify the point they belong to: GPS time = its index, RGB = the index's low bytes
  std::vector<double> gps(n);
  std::vector<uint8_t> rgb(n * 3);
  for (size_t i = 0; i < n; ++i) {
    gps[i] = (double)i;
    rgb[3 * i] = (uint8_t)i, rgb[3 * i + 1] = (uint8_t)(i >> 8), rgb[3 * i + 2] = (uint8_t)(i >> 16);
  }

  for (int shards : {1, 2, 4, 8}) {
    if (transport == 1 && shards > 1) {
      // RCCL wants one GPU per rank; run what the box offers
      swz_group* probe = nullptr;
      std::vector<int> dev(shards);
      for (int i = 0; i < shards; ++i) dev[i] = i;
      if (swz_group_create(shards, dev.data(), 1, &probe) != SWZ_OK) {
        std::printf("RCCL with %d shards skipped: not that many GPUs\n", shards);
        continue;
      }
      swz_group_destroy(probe);
    }
    std::vector<int> devices(shards, 0);
    if (transport == 1)
      for (int i = 0; i < shards; ++i) devices[i] = i;
    swz_group* g = nullptr;
    if (swz_group_create(shards, devices.data(), transport, &g) != SWZ_OK) return fail("swz_group_create", swz_group_last_error(nullptr));
    // uneven pieces in input order; the second piece is empty
    std::vector<size_t> cut(shards + 1, 0);
    for (int s = 1; s <= shards; ++s) cut[s] = (s == 2 && shards > 2) ? cut[1] : std::min(n, (size_t)((double)n * s * s / ((double)shards * shards)));
    cut[shards] = n;
    // (FAST: TilingAlgorithmV3 through the same call -- start level from the shards' summed prefix counts, the skipped levels
    // rebuilt per shard, the root from all shards' level-0 nodes on shard 0; the rows then carry the dup mask as well)
    for (int strategy : {SWZ_ACCURATE, SWZ_FAST})
    for (int sampler = 0; sampler < 4; ++sampler) {
      std::vector<double*> d_xyz(shards, nullptr);
      std::vector<swz_attribute_columns> d_attrs(shards);
      std::vector<uint64_t> cnt(shards, 0);
      for (int s = 0; s < shards; ++s) {
        cnt[s] = cut[s + 1] - cut[s];
        d_attrs[s] = swz_attribute_columns{};
        const uint64_t rows = std::max<uint64_t>(cnt[s], 1);
        swz_ctx* c = swz_group_ctx(g, s);
        if (swz_device_alloc_on(c, rows * 24, (void**)&d_xyz[s]) != SWZ_OK || swz_device_alloc_on(c, rows * 8, &d_attrs[s].column[SWZ_ATTR_GPS_TIME]) != SWZ_OK ||
            swz_device_alloc_on(c, rows * 3, &d_attrs[s].column[SWZ_ATTR_RGB]) != SWZ_OK)
          return fail("device alloc");
        if (!cnt[s]) continue;
        if (swz_copy_to_device(c, d_xyz[s], xyz.data() + 3 * cut[s], cnt[s] * 24) != SWZ_OK ||
            swz_copy_to_device(c, d_attrs[s].column[SWZ_ATTR_GPS_TIME], gps.data() + cut[s], cnt[s] * 8) != SWZ_OK ||
            swz_copy_to_device(c, d_attrs[s].column[SWZ_ATTR_RGB], rgb.data() + 3 * cut[s], cnt[s] * 3) != SWZ_OK)
          return fail("upload");
      }
      swz_tile_params p{};
      p.sampler = sampler;
      p.max_points_per_node = 2000;
      p.spacing_at_root = spacing;
      p.max_depth = 100;
      p.strategy = strategy;
      p.fast_concurrency = 8;
      std::vector<swz_group_result> res(shards);
      if (swz_group_tile(g, d_xyz.data(), d_attrs.data(), cnt.data(), mn, mx, &p, res.data()) != SWZ_OK) return fail("swz_group_tile", swz_group_last_error(g));

      std::vector<Row> got;
      for (int s = 0; s < shards; ++s) {
        const uint64_t m = res[s].num_points;
        std::vector<double> px(m * 3);
        std::vector<uint64_t> k(m);
        std::vector<uint32_t> perm(m);
        std::vector<int8_t> lv(m);
        std::vector<uint32_t> dup(m, 0u);
        std::vector<double> got_gps(m);
        std::vector<uint8_t> got_rgb(m * 3);
        swz_ctx* c = swz_group_ctx(g, s);
        if (m) {
          if (!res[s].attrs.column[SWZ_ATTR_GPS_TIME] || !res[s].attrs.column[SWZ_ATTR_RGB]) return fail("attribute columns missing in the result");
          if (swz_copy_to_host(c, got_gps.data(), res[s].attrs.column[SWZ_ATTR_GPS_TIME], m * 8) ||
              swz_copy_to_host(c, got_rgb.data(), res[s].attrs.column[SWZ_ATTR_RGB], m * 3))
            return fail("download of the attributes");
          if (swz_copy_to_host(c, px.data(), res[s].d_xyz, m * 24) || swz_copy_to_host(c, k.data(), res[s].d_keys, m * 8) ||
              swz_copy_to_host(c, perm.data(), res[s].d_perm, m * 4) || swz_copy_to_host(c, lv.data(), res[s].d_level, m))
            return fail("download");
          if ((strategy == SWZ_FAST) != (res[s].d_dup != nullptr)) return fail("d_dup must be set for FAST and only then");
          if (res[s].d_dup && swz_copy_to_host(c, dup.data(), res[s].d_dup, m * 4)) return fail("download of the dup mask");
        }
        for (uint64_t i = 0; i < m; ++i) {
          if (i && k[i] < k[i - 1]) return fail("keys of a shard do not ascend");
          const int octant = (int)(k[i] >> 60);
          if (octant * shards / 8 != s) return fail("point on a shard that does not own its octant");
          const uint32_t q = perm[i];
          if (q >= m) return fail("perm out of range");
          const size_t src = (size_t)got_gps[q];  // the attributes name the input point this row came from
          if (src >= n || xyz[3 * src] != px[3 * q] || xyz[3 * src + 1] != px[3 * q + 1] || xyz[3 * src + 2] != px[3 * q + 2])
            return fail("GPS time column did not travel with its point");
          if (got_rgb[3 * q] != (uint8_t)src || got_rgb[3 * q + 1] != (uint8_t)(src >> 8) || got_rgb[3 * q + 2] != (uint8_t)(src >> 16))
            return fail("RGB column did not travel with its point");
          got.emplace_back(k[i], px[3 * q], px[3 * q + 1], px[3 * q + 2], (int)(lv[i] & 0xFF) + (int)(dup[i] << 8));
        }
      }
      if (got.size() != n) return fail("points lost or duplicated in the exchange");

      std::vector<double> copy(xyz);
      std::vector<uint64_t> ok(n);
      std::vector<uint32_t> operm(n);
      std::vector<int8_t> olv(n);
      std::vector<uint32_t> odup(n, 0u);
      orc_tile_params op{sampler, 2000, spacing, 100, strategy == SWZ_FAST ? ORC_FAST : ORC_ACCURATE, 8};
      orc_tile_stats ost;
      if (orc_tile(copy.data(), n, mn, mx, &op, ok.data(), operm.data(), olv.data(), strategy == SWZ_FAST ? odup.data() : nullptr, &ost) != 0)
        return fail("oracle");
      std::vector<Row> want;
      want.reserve(n);
      for (size_t i = 0; i < n; ++i) {
        const uint32_t q = operm[i];
        want.emplace_back(ok[i], copy[3 * q], copy[3 * q + 1], copy[3 * q + 2], (int)(olv[i] & 0xFF) + (int)(odup[i] << 8));
      }
      std::sort(got.begin(), got.end());
      std::sort(want.begin(), want.end());
      if (got != want) {
        size_t bad = 0, first = n;
        for (size_t i = 0; i < n; ++i)
          if (got[i] != want[i]) {
            ++bad;
            if (first == n) first = i;
          }
        std::fprintf(stderr, "%s%s, %d shards: %zu rows differ, first at %zu (level | dup << 8: %d vs %d)\n", strategy == SWZ_FAST ? "FAST " : "",
                     names[sampler], shards, bad, first, std::get<4>(got[first]), std::get<4>(want[first]));
        if (strategy == SWZ_FAST) {  // which octants' points the two roots hold
          size_t g8[8] = {0}, w8[8] = {0};
          for (size_t i = 0; i < n; ++i) {
            if ((std::get<4>(got[i]) >> 8) & 1) ++g8[std::get<0>(got[i]) >> 60];
            if ((std::get<4>(want[i]) >> 8) & 1) ++w8[std::get<0>(want[i]) >> 60];
          }
          for (int o = 0; o < 8; ++o) std::fprintf(stderr, "  root points of octant %d: %zu, oracle %zu\n", o, g8[o], w8[o]);
        }
        return fail("sharded result differs from the oracle");
      }
      std::printf("%s%-12s %d shard(s) ok\n", strategy == SWZ_FAST ? "FAST " : "", names[sampler], shards);
      for (int s = 0; s < shards; ++s) {
        swz_device_free(d_xyz[s]);
        swz_device_free(d_attrs[s].column[SWZ_ATTR_GPS_TIME]);
        swz_device_free(d_attrs[s].column[SWZ_ATTR_RGB]);
      }
    }

    // ---- a lopsided cloud: nearly all points in the lowest shard's octants, a handful in the others.  In the
    // joint MIN_DISTANCE root the nearly empty shards' face cells look at the busy shard every round for as long as its
    // sweep takes: their round limit has to cover that, and the result must still be the oracle's.
    if (shards > 1) {
      std::vector<double> lop(xyz);
      for (size_t i = 0; i < n; ++i)
        if (i % 6000 != 0) {  // (50 points stay where they are, anywhere in the cube)
          lop[3 * i] *= 0.49;
          if (shards > 2) lop[3 * i + 1] *= 0.49;
          if (shards > 4) lop[3 * i + 2] *= 0.49;
        }
      std::vector<double*> d_xyz(shards, nullptr);
      std::vector<swz_attribute_columns> d_attrs(shards);
      std::vector<uint64_t> cnt(shards, 0);
      for (int s = 0; s < shards; ++s) {
        cnt[s] = cut[s + 1] - cut[s];
        d_attrs[s] = swz_attribute_columns{};
        swz_ctx* c = swz_group_ctx(g, s);
        if (swz_device_alloc_on(c, std::max<uint64_t>(cnt[s], 1) * 24, (void**)&d_xyz[s]) != SWZ_OK) return fail("device alloc");
        if (cnt[s] && swz_copy_to_device(c, d_xyz[s], lop.data() + 3 * cut[s], cnt[s] * 24) != SWZ_OK) return fail("upload");
      }
      swz_tile_params p{};
      p.sampler = SWZ_MIN_DISTANCE;
      p.max_points_per_node = 2000;
      p.spacing_at_root = spacing;
      p.max_depth = 100;
      p.strategy = SWZ_ACCURATE;
      p.fast_concurrency = 8;
      std::vector<swz_group_result> res(shards);
      if (swz_group_tile(g, d_xyz.data(), d_attrs.data(), cnt.data(), mn, mx, &p, res.data()) != SWZ_OK) return fail("swz_group_tile (lopsided cloud)", swz_group_last_error(g));
      std::vector<std::pair<uint64_t, int>> got, want;
      uint64_t smallest = n;
      for (int s = 0; s < shards; ++s) {
        const uint64_t m = res[s].num_points;
        if (m) smallest = std::min(smallest, m);
        std::vector<uint64_t> k(m);
        std::vector<int8_t> lv(m);
        swz_ctx* c = swz_group_ctx(g, s);
        if (m && (swz_copy_to_host(c, k.data(), res[s].d_keys, m * 8) || swz_copy_to_host(c, lv.data(), res[s].d_level, m))) return fail("download");
        for (uint64_t i = 0; i < m; ++i) got.emplace_back(k[i], (int)lv[i]);
      }
      std::vector<double> copy(lop);
      std::vector<uint64_t> ok(n);
      std::vector<uint32_t> operm(n);
      std::vector<int8_t> olv(n);
      orc_tile_params op{SWZ_MIN_DISTANCE, 2000, spacing, 100, ORC_ACCURATE, 8};
      orc_tile_stats ost;
      if (orc_tile(copy.data(), n, mn, mx, &op, ok.data(), operm.data(), olv.data(), nullptr, &ost) != 0) return fail("oracle");
      for (size_t i = 0; i < n; ++i) want.emplace_back(ok[i], (int)olv[i]);
      std::sort(got.begin(), got.end());
      std::sort(want.begin(), want.end());
      if (got != want) return fail("lopsided cloud: sharded MIN_DISTANCE differs from the oracle");
      std::printf("MIN_DISTANCE %d shard(s), lopsided cloud (smallest shard with points: %llu) ok\n", shards, (unsigned long long)smallest);
      // the same batch with SWZ_FLAG_MIN_DISTANCE_PROPERTY: the root spans the shards and stays the exact one, the levels
      // below run in property mode (checked on one context in tests/test_min_distance_property.py): every point gets a level
      p.flags = SWZ_FLAG_MIN_DISTANCE_PROPERTY;
      if (swz_group_tile(g, d_xyz.data(), d_attrs.data(), cnt.data(), mn, mx, &p, res.data()) != SWZ_OK) return fail("swz_group_tile (property mode)", swz_group_last_error(g));
      std::vector<uint64_t> root_got, root_want;
      uint64_t total_p = 0;
      for (int s = 0; s < shards; ++s) {
        const uint64_t m = res[s].num_points;
        std::vector<uint64_t> k(m);
        std::vector<int8_t> lv(m);
        swz_ctx* c = swz_group_ctx(g, s);
        if (m && (swz_copy_to_host(c, k.data(), res[s].d_keys, m * 8) || swz_copy_to_host(c, lv.data(), res[s].d_level, m))) return fail("download");
        for (uint64_t i = 0; i < m; ++i) {
          if (lv[i] < -1 || lv[i] > 20) return fail("property mode: a point without a level");
          if (lv[i] == -1) root_got.push_back(k[i]);
        }
        total_p += m;
      }
      for (size_t i = 0; i < n; ++i)
        if (olv[i] == -1) root_want.push_back(ok[i]);
      std::sort(root_got.begin(), root_got.end());
      std::sort(root_want.begin(), root_want.end());
      if (total_p != n || root_got != root_want) return fail("property mode: the root of a sharded batch must be the exact one");
      std::printf("MIN_DISTANCE %d shard(s), property mode below the exact root: fine\n", shards);
      for (int s = 0; s < shards; ++s) swz_device_free(d_xyz[s]);
    }

    // ---- the same group, a data set in several batches (swz_group_add_batch / _stage_batch, one swz_tiler per shard):
    // the union of the shards' node files = the multi-batch oracle's, file by file and in file order; the root's file
    // is the concatenation of the shards' parts in shard order.  k = 1 from device buffers, k = 3 staged from pinned memory.
    if (transport == 0 || shards == 1)
      for (int sampler = 0; sampler < 4; ++sampler)
       for (int strategy : {SWZ_ACCURATE, SWZ_FAST})  // FAST: start level from the summed histograms, root rebuilt on shard 0
        for (int k : {1, 3}) {
          swz_tile_params p{};
          p.sampler = sampler;
          p.max_points_per_node = 2000;
          p.spacing_at_root = spacing;
          p.max_depth = 100;
          p.strategy = strategy;
          p.fast_concurrency = 8;
          if (swz_group_tiler_open(g, mn, mx, &p, 0) != SWZ_OK) return fail("swz_group_tiler_open", swz_group_last_error(g));
          orc_tile_params op{sampler, 2000, spacing, 100, strategy == SWZ_FAST ? ORC_FAST : ORC_ACCURATE, 8};
          const bool have_expect = g_expect.count(std::make_tuple(sampler, strategy, k)) != 0;
          orc_tiler* oracle = have_expect ? nullptr : orc_tiler_create(mn, mx, &op);
          std::vector<std::vector<void*>> pinned;  // freed after the data set
          for (int b = 0; b < k; ++b) {
            const size_t b0 = n * b / k, b1 = n * (b + 1) / k, bn = b1 - b0;
            std::vector<double> copy(xyz.begin() + 3 * b0, xyz.begin() + 3 * b1);
            if (oracle && orc_tiler_add_batch(oracle, copy.data(), bn) != 0) return fail("oracle tiler batch");
            // the batch's points lie on the shards in uneven pieces (the last shard takes the rest, the second is empty)
            std::vector<size_t> bc(shards + 1, 0);
            for (int s = 1; s <= shards; ++s) bc[s] = (s == 2 && shards > 2) ? bc[1] : std::min(bn, (size_t)((double)bn * s * s / ((double)shards * shards)));
            bc[shards] = bn;
            std::vector<uint64_t> cnt(shards);
            std::vector<swz_tile_stats> st(shards);
            if (k == 1) {
              std::vector<double*> d_xyz(shards, nullptr);
              std::vector<swz_attribute_columns> d_attrs(shards);
              for (int s = 0; s < shards; ++s) {
                cnt[s] = bc[s + 1] - bc[s];
                d_attrs[s] = swz_attribute_columns{};
                const uint64_t rows = std::max<uint64_t>(cnt[s], 1);
                swz_ctx* c = swz_group_ctx(g, s);
                if (swz_device_alloc_on(c, rows * 24, (void**)&d_xyz[s]) != SWZ_OK || swz_device_alloc_on(c, rows * 8, &d_attrs[s].column[SWZ_ATTR_GPS_TIME]) != SWZ_OK)
                  return fail("device alloc");
                if (cnt[s] && (swz_copy_to_device(c, d_xyz[s], xyz.data() + 3 * (b0 + bc[s]), cnt[s] * 24) != SWZ_OK ||
                               swz_copy_to_device(c, d_attrs[s].column[SWZ_ATTR_GPS_TIME], gps.data() + b0 + bc[s], cnt[s] * 8) != SWZ_OK))
                  return fail("upload");
              }
              if (swz_group_add_batch(g, d_xyz.data(), d_attrs.data(), cnt.data(), st.data()) != SWZ_OK) return fail("swz_group_add_batch", swz_group_last_error(g));
              for (int s = 0; s < shards; ++s) {
                swz_device_free(d_xyz[s]);
                swz_device_free(d_attrs[s].column[SWZ_ATTR_GPS_TIME]);
              }
            } else {
              std::vector<const double*> h_xyz(shards, nullptr);
              std::vector<swz_attribute_columns> h_attrs(shards);
              pinned.emplace_back();
              for (int s = 0; s < shards; ++s) {
                cnt[s] = bc[s + 1] - bc[s];
                h_attrs[s] = swz_attribute_columns{};
                void *px = nullptr, *pg = nullptr;
                if (swz_host_alloc_pinned(std::max<uint64_t>(cnt[s], 1) * 24, &px) != SWZ_OK || swz_host_alloc_pinned(std::max<uint64_t>(cnt[s], 1) * 8, &pg) != SWZ_OK)
                  return fail("pinned alloc");
                pinned.back().push_back(px);
                pinned.back().push_back(pg);
                std::copy(xyz.begin() + 3 * (b0 + bc[s]), xyz.begin() + 3 * (b0 + bc[s + 1]), (double*)px);
                std::copy(gps.begin() + b0 + bc[s], gps.begin() + b0 + bc[s + 1], (double*)pg);
                h_xyz[s] = (const double*)px;
                h_attrs[s].column[SWZ_ATTR_GPS_TIME] = pg;
              }
              // batch b + 1 is staged before batch b is tiled whenever there is one: its copy runs beside the kernels
              if (b == 0 && swz_group_stage_batch(g, h_xyz.data(), h_attrs.data(), cnt.data()) != SWZ_OK) return fail("swz_group_stage_batch", swz_group_last_error(g));
              if (b + 1 < k) {
                const size_t c0 = n * (b + 1) / k, c1 = n * (b + 2) / k, cn = c1 - c0;
                std::vector<size_t> cc(shards + 1, 0);
                for (int s = 1; s <= shards; ++s) cc[s] = (s == 2 && shards > 2) ? cc[1] : std::min(cn, (size_t)((double)cn * s * s / ((double)shards * shards)));
                cc[shards] = cn;
                std::vector<const double*> n_xyz(shards, nullptr);
                std::vector<swz_attribute_columns> n_attrs(shards);
                std::vector<uint64_t> ncnt(shards);
                pinned.emplace_back();
                for (int s = 0; s < shards; ++s) {
                  ncnt[s] = cc[s + 1] - cc[s];
                  n_attrs[s] = swz_attribute_columns{};
                  void *px = nullptr, *pg = nullptr;
                  if (swz_host_alloc_pinned(std::max<uint64_t>(ncnt[s], 1) * 24, &px) != SWZ_OK || swz_host_alloc_pinned(std::max<uint64_t>(ncnt[s], 1) * 8, &pg) != SWZ_OK)
                    return fail("pinned alloc");
                  pinned.back().push_back(px);
                  pinned.back().push_back(pg);
                  std::copy(xyz.begin() + 3 * (c0 + cc[s]), xyz.begin() + 3 * (c0 + cc[s + 1]), (double*)px);
                  std::copy(gps.begin() + c0 + cc[s], gps.begin() + c0 + cc[s + 1], (double*)pg);
                  n_xyz[s] = (const double*)px;
                  n_attrs[s].column[SWZ_ATTR_GPS_TIME] = pg;
                }
                if (swz_group_stage_batch(g, n_xyz.data(), n_attrs.data(), ncnt.data()) != SWZ_OK) return fail("swz_group_stage_batch (next)", swz_group_last_error(g));
              }
              if (swz_group_tile_staged(g, st.data()) != SWZ_OK) return fail("swz_group_tile_staged", swz_group_last_error(g));
            }
          }
          if (swz_group_finalize(g, nullptr) != SWZ_OK) return fail("swz_group_finalize", swz_group_last_error(g));
          if (oracle && orc_tiler_finalize(oracle) != 0) return fail("oracle tiler finalize");
          // the shards' node files, ids translated to input indices through the GPS-time column of the shard's pools
          std::map<std::string, std::vector<uint32_t>> got_files;
          uint64_t total_points = 0;
          for (int s = 0; s < shards; ++s) {
            swz_tiler* t = swz_group_tiler(g, s);
            swz_ctx* c = swz_group_ctx(g, s);
            swz_tiler_info info{};
            if (swz_tiler_get_info(t, &info) != SWZ_OK) return fail("swz_tiler_get_info");
            total_points += info.num_points;
            if (!info.num_stored) continue;
            std::vector<int8_t> nl(info.num_nodes);
            std::vector<uint64_t> nk(info.num_nodes), no(info.num_nodes), nc(info.num_nodes);
            uint64_t nn = 0;
            if (swz_tiler_node_table(t, info.num_nodes, nl.data(), nk.data(), no.data(), nc.data(), &nn) != SWZ_OK) return fail("swz_tiler_node_table");
            uint32_t* d_ids = nullptr;
            if (swz_device_alloc_on(c, info.num_stored * 4, (void**)&d_ids) != SWZ_OK) return fail("device alloc");
            if (swz_tiler_export_device(t, nullptr, d_ids, nullptr) != SWZ_OK) return fail("swz_tiler_export_device");
            std::vector<uint32_t> ids(info.num_stored);
            if (swz_copy_to_host(c, ids.data(), d_ids, info.num_stored * 4) != SWZ_OK) return fail("download ids");
            swz_device_free(d_ids);
            const double* d_pool = nullptr;
            swz_attribute_columns pool_cols{};
            if (swz_tiler_pools_device(t, &d_pool, &pool_cols) != SWZ_OK || !pool_cols.column[SWZ_ATTR_GPS_TIME]) return fail("swz_tiler_pools_device");
            std::vector<double> pool_gps(info.num_points);
            if (swz_copy_to_host(c, pool_gps.data(), pool_cols.column[SWZ_ATTR_GPS_TIME], info.num_points * 8) != SWZ_OK) return fail("download pool column");
            for (uint64_t j = 0; j < nn; ++j) {
              std::string name = "r";
              for (int l = 0; l <= nl[j]; ++l) name.push_back((char)('0' + swz_host::get_octant_at_level(nk[j], (uint32_t)l)));
              if (nl[j] >= 0 && got_files.count(name)) return fail("a node below the root on two shards", name.c_str());
              std::vector<uint32_t>& f = got_files[name];  // (the root: appended in shard order)
              for (uint64_t i = no[j]; i < no[j] + nc[j]; ++i) {
                if (ids[i] >= info.num_points) return fail("point id out of range");
                f.push_back((uint32_t)pool_gps[ids[i]]);
              }
            }
          }
          if (total_points != n) return fail("points lost or duplicated over the batches");
          if (oracle) {
            uint64_t onn = 0, ons = 0;
            orc_tiler_counts(oracle, &onn, &ons, nullptr, nullptr);
            std::vector<int8_t> onl(onn);
            std::vector<uint64_t> onk(onn), ono(onn), onc(onn);
            std::vector<uint32_t> oids(ons);
            orc_tiler_export(oracle, onl.data(), onk.data(), ono.data(), onc.data(), oids.data(), nullptr);
            orc_tiler_destroy(oracle);
            std::map<std::string, std::vector<uint32_t>>& e = g_expect[std::make_tuple(sampler, strategy, k)];
            for (uint64_t j = 0; j < onn; ++j) {
              std::string name = "r";
              for (int l = 0; l <= onl[j]; ++l) name.push_back((char)('0' + swz_host::get_octant_at_level(onk[j], (uint32_t)l)));
              e[name].assign(oids.begin() + ono[j], oids.begin() + ono[j] + onc[j]);
            }
          }
          const std::map<std::string, std::vector<uint32_t>>& expect = g_expect[std::make_tuple(sampler, strategy, k)];
          if (got_files != expect) {
            size_t bad = 0;
            for (const auto& kv : expect) {
              const bool differs = !got_files.count(kv.first) || got_files.at(kv.first) != kv.second;
              if (differs && bad < 4)
                std::fprintf(stderr, "  node %s: %zu points expected, %zu here\n", kv.first.c_str(), kv.second.size(),
                             got_files.count(kv.first) ? got_files.at(kv.first).size() : (size_t)0);
              bad += differs;
            }
            std::fprintf(stderr, "%s %s, %d shards, %d batches: %zu of %zu node files differ (%zu files here)\n", names[sampler],
                         strategy == SWZ_FAST ? "FAST" : "ACCURATE", shards, k, bad, expect.size(), got_files.size());
            return fail("node files of the sharded multi-batch tilers differ from the oracle");
          }
          if (swz_group_tiler_close(g) != SWZ_OK) return fail("swz_group_tiler_close");
          for (auto& v : pinned)
            for (void* q : v) swz_host_free_pinned(q);
          std::printf("%-12s %-8s %d shard(s) %d batch(es) files ok: %zu nodes\n", names[sampler], strategy == SWZ_FAST ? "FAST" : "ACCURATE", shards, k,
                      expect.size());
        }
    swz_group_destroy(g);

    // the reference-shaped C++ class on top: node files (names, contents, order) equal the oracle's
    if (transport == 0)
      for (int sampler = 0; sampler < 4; ++sampler) {
        RowSink sink;
        sink.xyz = &xyz;
        swz_host::TilerMetaParameters meta;
        meta.spacing_at_root = spacing;
        meta.max_points_per_node = 2000;
        swz_host::ShardedTilingAlgorithmGPU tiler(swz_host::make_sampling_strategy_from_name(names[sampler], 2000), sink, meta, devices, 0);
        swz_attribute_columns cols{};
        cols.column[SWZ_ATTR_GPS_TIME] = gps.data();
        cols.column[SWZ_ATTR_RGB] = rgb.data();
        const swz_host::AABB bounds{{0, 0, 0}, {1, 1, 1}};
        const size_t files = tiler.tile_batch(xyz.data(), &cols, n, bounds);
        if (!sink.rows_match) return fail("a row of a node file is not the point its attributes name, or lies outside the node");
        orc_tile_params op{sampler, 2000, spacing, 100, ORC_ACCURATE, 8};
        orc_tiler* oracle = orc_tiler_create(mn, mx, &op);
        std::vector<double> copy(xyz);
        if (orc_tiler_add_batch(oracle, copy.data(), n) != 0 || orc_tiler_finalize(oracle) != 0) return fail("oracle tiler");
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
          for (int l = 0; l <= nl[j]; ++l) name.push_back((char)('0' + swz_host::get_octant_at_level(nk[j], (uint32_t)l)));
          expect[name].assign(ids.begin() + no[j], ids.begin() + no[j] + nc[j]);
        }
        if (files != expect.size() || expect != sink.nodes) return fail("node files of the sharded C++ tiler differ from the oracle");
        std::printf("%-12s %d shard(s) files ok: %zu nodes\n", names[sampler], shards, files);
      }
  }
  return 0;
}
