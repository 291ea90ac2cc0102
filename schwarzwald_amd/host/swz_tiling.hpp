// swz_tiling.hpp -- host-side C++17 mirror of the reference's tiler/sampling interface on top of the
// C ABI (include/swz_gpu.h).  Header only, no dependency besides the C++ standard library and
// libswz_gpu.so.  Names, argument meaning and error behaviour follow the reference seams so that the
// adapter a maintainer writes inside Schwarzwald is a thin forwarding layer:
//
//   reference (schwarzwald/core/...)                         here (namespace swz_host)
//   ------------------------------------------------------   -----------------------------------------
//   Vector3<double>, AABB            math/Vector3.h, AABB.h   Vector3d, AABB
//   MortonIndex64                    MortonIndex.h:80-169     MortonIndex64 (uint64_t) + helpers
//   IndexedPoint64                   tiling/Sampling.h:147    IndexedPoint64 {point_index, morton_index}
//   OutlierPointsBehaviour           OctreeAlgorithms.h:20    OutlierPointsBehaviour
//   index_points<21>                 OctreeAlgorithms.h:181   index_points
//   Range::sort                      containers/Range.h:62    sort_indexed_points
//   SamplingStrategy + factory       Sampling.h:761-791       SamplingStrategy, make_sampling_strategy_from_name
//   SamplingBehaviour                Sampling.h:170-181       SamplingBehaviour
//   sample_points                    Sampling.h:799-821       sample_points (returns the partition point)
//   TilerMetaParameters              process/Tiler.h:64-75    TilerMetaParameters
//   PointsPersistence::persist_points io/PointsPersistence.h  PointsSink::persist_points
//   TilingAlgorithmBase              TilingAlgorithms.h:70    TilingAlgorithmGPU::tile_batch / finalize
//   get_octant_bounds                OctreeAlgorithms.cpp:3   get_octant_bounds
//
// Errors: every failing ABI call becomes std::runtime_error carrying swz_last_error(), the way the
// reference reports tiler failures (executable/main.cpp:599-602).
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/swz_gpu.h"

namespace swz_host {

struct Vector3d {
  double x = 0, y = 0, z = 0;
};
struct AABB {
  Vector3d min, max;
  Vector3d extent() const { return {max.x - min.x, max.y - min.y, max.z - min.z}; }
};

using MortonIndex64 = uint64_t;
constexpr unsigned MortonIndex64Levels = 21;

struct IndexedPoint64 {
  uint32_t point_index;  // stands for PointBuffer::PointReference: row of the position array
  MortonIndex64 morton_index;
};

enum class OutlierPointsBehaviour { ClampToBounds, Abort };
enum class SamplingBehaviour { TakeAllWhenCountBelowMaxPoints, AlwaysAdhereToMinSpacing };
enum class TilingStrategy { Accurate, Fast };

struct SamplingStrategy {
  int kind;  // SWZ_RANDOM_GRID ...
  size_t max_points_per_node;
};
// make_sampling_strategy_from_name -- core/tiling/Sampling.h:774-791 (MIN_DISTANCE_FAST is out of scope)
inline SamplingStrategy make_sampling_strategy_from_name(const std::string& name, size_t max_points_per_node) {
  if (name == "RANDOM_GRID") return {SWZ_RANDOM_GRID, max_points_per_node};
  if (name == "GRID_CENTER") return {SWZ_GRID_CENTER, max_points_per_node};
  if (name == "MIN_DISTANCE") return {SWZ_MIN_DISTANCE, max_points_per_node};
  if (name == "JITTERED") return {SWZ_JITTERED, max_points_per_node};
  throw std::runtime_error{"Unrecognized sampling strategy name \"" + name + "\""};
}

struct TilerMetaParameters {  // core/process/Tiler.h:64-75
  float spacing_at_root = 0.f;
  uint32_t max_depth = 100;
  size_t max_points_per_node = 20000;
  TilingStrategy tiling_strategy = TilingStrategy::Accurate;
  uint32_t num_indexing_threads = 8;  // FAST: decides the start level
};

// get_octant_bounds -- core/tiling/OctreeAlgorithms.cpp:3-18
inline AABB get_octant_bounds(uint8_t octant, const AABB& parent) {
  const Vector3d e = parent.extent();
  const double min_z = (octant & 1) ? (parent.min.z + e.z / 2) : parent.min.z;
  const double min_y = ((octant >> 1) & 1) ? (parent.min.y + e.y / 2) : parent.min.y;
  const double min_x = ((octant >> 2) & 1) ? (parent.min.x + e.x / 2) : parent.min.x;
  return {{min_x, min_y, min_z}, {min_x + e.x / 2, min_y + e.y / 2, min_z + e.z / 2}};
}
inline uint8_t get_octant_at_level(MortonIndex64 key, uint32_t level) {
  return static_cast<uint8_t>((key >> ((MortonIndex64Levels - level - 1) * 3)) & 0b111);
}

// One swz_ctx, owned.  Not thread-safe (one call in flight per context).
class Context {
public:
  explicit Context(int device = 0) {
    if (swz_create(&_ctx, device) != SWZ_OK) throw std::runtime_error{swz_last_error(nullptr)};
  }
  ~Context() { swz_destroy(_ctx); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  swz_ctx* get() const { return _ctx; }
  void check(int status) const {
    if (status != SWZ_OK) throw std::runtime_error{swz_last_error(_ctx)};
  }

private:
  swz_ctx* _ctx = nullptr;
};

// index_points<21> -- OctreeAlgorithms.h:181-197.  positions: n x 3 doubles (PointBuffer::positions().data());
// outliers are clamped in place, like index_point does.
inline void index_points(Context& ctx, double* positions, size_t n, IndexedPoint64* indexed_points_begin,
                         const AABB& bounds, OutlierPointsBehaviour behaviour) {
  if (behaviour != OutlierPointsBehaviour::ClampToBounds)
    throw std::runtime_error{"only OutlierPointsBehaviour::ClampToBounds is implemented (the tiler never uses Abort)"};
  std::vector<uint64_t> keys(n);
  const double mn[3] = {bounds.min.x, bounds.min.y, bounds.min.z}, mx[3] = {bounds.max.x, bounds.max.y, bounds.max.z};
  ctx.check(swz_morton_encode(ctx.get(), positions, n, mn, mx, keys.data()));
  for (size_t i = 0; i < n; ++i) indexed_points_begin[i] = {static_cast<uint32_t>(i), keys[i]};
}

// Range<IndexedPointsIter>::sort -- containers/Range.h:62-66; ties keep their input order
inline void sort_indexed_points(Context& ctx, IndexedPoint64* begin, IndexedPoint64* end) {
  const size_t n = static_cast<size_t>(end - begin);
  std::vector<uint64_t> keys(n);
  for (size_t i = 0; i < n; ++i) keys[i] = begin[i].morton_index;
  std::vector<uint32_t> perm(n);
  ctx.check(swz_sort_by_key(ctx.get(), keys.data(), n, perm.data(), nullptr));
  std::vector<IndexedPoint64> tmp(begin, end);
  for (size_t i = 0; i < n; ++i) begin[i] = tmp[perm[i]];
}

// sample_points -- Sampling.h:799-821: stable partition of [begin, end) into [taken | rest], returns the
// partition point.  positions: the PointBuffer's position array the point indices refer to.
inline IndexedPoint64* sample_points(Context& ctx, const SamplingStrategy& strategy, IndexedPoint64* begin,
                                     IndexedPoint64* end, MortonIndex64 node_key, int32_t node_level,
                                     const AABB& root_bounds, float spacing_at_root, SamplingBehaviour behaviour,
                                     const double* positions, size_t num_positions) {
  const size_t n = static_cast<size_t>(end - begin);
  std::vector<uint64_t> keys(n);
  std::vector<uint32_t> idx(n);
  for (size_t i = 0; i < n; ++i) {
    keys[i] = begin[i].morton_index;
    idx[i] = begin[i].point_index;
  }
  std::vector<uint8_t> taken(n);
  uint64_t num_taken = 0;
  const double mn[3] = {root_bounds.min.x, root_bounds.min.y, root_bounds.min.z};
  const double mx[3] = {root_bounds.max.x, root_bounds.max.y, root_bounds.max.z};
  ctx.check(swz_sample_points(ctx.get(), strategy.kind, strategy.max_points_per_node, keys.data(), idx.data(), n,
                              positions, num_positions, node_key, node_level, mn, mx, spacing_at_root,
                              behaviour == SamplingBehaviour::AlwaysAdhereToMinSpacing
                                ? SWZ_ALWAYS_ADHERE_TO_MIN_SPACING
                                : SWZ_TAKE_ALL_WHEN_COUNT_BELOW_MAX_POINTS,
                              taken.data(), &num_taken));
  std::vector<IndexedPoint64> tmp(begin, end);
  IndexedPoint64* front = begin;
  IndexedPoint64* back = begin + num_taken;
  for (size_t i = 0; i < n; ++i) *(taken[i] ? front++ : back++) = tmp[i];
  return begin + num_taken;
}

// What the adapter needs from PointsPersistence (core/io/PointsPersistence.h:23-53).
struct PointsSink {
  virtual ~PointsSink() = default;
  // The complete content of one node file, in file (Morton) order.  ids are point ids: the running index of the
  // point over ALL batches in input order (batch 0 holds ids 0..n0-1, batch 1 the next n1, ...); positions are
  // the node's count x 3 clamped positions, gathered on the device (what persist_points reads through the
  // PointReferences, core/tiling/TilingAlgorithms.cpp:232-236, 316-322).
  virtual void persist_points(const uint32_t* ids_begin, const uint32_t* ids_end, const double* positions,
                              const AABB& node_bounds, const std::string& node_name) = 0;
  // Sharded batches (ShardedTilingAlgorithmGPU): the attribute columns travelled with the points between the
  // GPUs, so a node's file arrives as rows instead of ids: count positions plus every attribute column present
  // (host memory, count rows each, in file order).
  virtual void persist_rows(size_t /*count*/, const double* /*positions*/, const swz_attribute_columns& /*attributes*/,
                            const AABB& /*node_bounds*/, const std::string& /*node_name*/) {
    throw std::runtime_error{"PointsSink::persist_rows is not implemented by this sink"};
  }
};

// The shape of TilingAlgorithmBase (core/tiling/TilingAlgorithms.h:70-116): one object per Tiler, fed one batch at
// a time (Tiler.cpp:509-510), finalize() once at the end.  tile_batch() is what the single task emitted by
// build_execution_graph() runs.  Unlike the reference, which rewrites the files of every node a batch reaches
// (and re-reads them for the next batch, TilingAlgorithms.cpp:50-109), the node files live in device memory
// between the batches (swz_tiler) and each is handed to the persistence ONCE, by finalize(): same final files,
// no intermediate ones.
class TilingAlgorithmGPU {
public:
  TilingAlgorithmGPU(SamplingStrategy sampling_strategy, PointsSink& persistence, TilerMetaParameters meta,
                     int device = 0)
    : _ctx(device), _sampling_strategy(sampling_strategy), _persistence(persistence), _meta(meta) {}
  ~TilingAlgorithmGPU() { swz_tiler_destroy(_tiler); }
  TilingAlgorithmGPU(const TilingAlgorithmGPU&) = delete;
  TilingAlgorithmGPU& operator=(const TilingAlgorithmGPU&) = delete;

  // positions: n x 3 doubles of this batch (host memory; pinned memory makes the copy asynchronous);
  // bounds: the octree's (cubic) root bounds, the same for every batch
  swz_tile_stats tile_batch(const double* positions, size_t n, const AABB& bounds) {
    if (_finalized) throw std::runtime_error{"TilingAlgorithmGPU: tile_batch after finalize"};
    if (!_tiler) {
      swz_tile_params p{};
      p.sampler = _sampling_strategy.kind;
      p.max_points_per_node = _sampling_strategy.max_points_per_node;
      p.spacing_at_root = _meta.spacing_at_root;
      p.max_depth = _meta.max_depth;
      p.strategy = _meta.tiling_strategy == TilingStrategy::Fast ? SWZ_FAST : SWZ_ACCURATE;
      p.fast_concurrency = _meta.num_indexing_threads;
      const double mn[3] = {bounds.min.x, bounds.min.y, bounds.min.z}, mx[3] = {bounds.max.x, bounds.max.y, bounds.max.z};
      _ctx.check(swz_tiler_create(_ctx.get(), mn, mx, &p, 0, &_tiler));
      _bounds = bounds;
    }
    swz_tile_stats stats{};
    _ctx.check(swz_tiler_add_batch(_tiler, positions, n, nullptr, &stats));
    return stats;
  }

  // TilingAlgorithmBase::finalize (FAST: reconstruct_left_out_nodes), then every node file goes to the persistence.
  // Returns the number of nodes persisted.
  size_t finalize(const AABB&) {
    if (!_tiler || _finalized) return 0;
    _finalized = true;
    swz_tile_stats stats{};
    _ctx.check(swz_tiler_finalize(_tiler, &stats));
    swz_tiler_info info{};
    _ctx.check(swz_tiler_get_info(_tiler, &info));
    const uint64_t ns = info.num_stored, nn = info.num_nodes;
    if (ns == 0) return 0;
    // The ids of all files in node order (4 bytes per stored point), then the files in CHUNKS of whole nodes: positions
    // gathered by id on the device, copied, handed to the persistence -- host and device hold ns * 4 + chunk * 24 bytes
    // instead of all ns * 28 at once, and the persistence starts writing while later chunks are still on the device.
    DeviceBuffer ids_buf(ns * 4);
    uint32_t* d_ids = static_cast<uint32_t*>(ids_buf.ptr);
    _ctx.check(swz_tiler_export_device(_tiler, nullptr, d_ids, nullptr));
    const double* pool = nullptr;
    _ctx.check(swz_tiler_pools_device(_tiler, &pool, nullptr));
    std::vector<uint32_t> ids(ns);
    _ctx.check(swz_copy_to_host(_ctx.get(), ids.data(), d_ids, ns * 4));
    std::vector<int8_t> nl(nn);
    std::vector<uint64_t> nk(nn), no(nn), nc(nn);
    uint64_t got = 0;
    _ctx.check(swz_tiler_node_table(_tiler, nn, nl.data(), nk.data(), no.data(), nc.data(), &got));
    uint64_t largest = 0;
    for (uint64_t j = 0; j < got; ++j) largest = std::max<uint64_t>(largest, nc[j]);
    const uint64_t chunk_cap = std::max<uint64_t>(std::max<uint64_t>(_export_chunk_points, 1), largest);
    DeviceBuffer xyz_buf(std::min(chunk_cap, ns) * 24);
    std::vector<double> xyz(std::min(chunk_cap, ns) * 3);
    for (uint64_t j0 = 0; j0 < got;) {
      uint64_t j1 = j0, cnt = 0;  // nodes [j0, j1): consecutive files, stored back to back from no[j0] on
      while (j1 < got && cnt + nc[j1] <= chunk_cap) cnt += nc[j1++];
      if (cnt) {
        _ctx.check(swz_gather_payload_device(_ctx.get(), d_ids + no[j0], nullptr, cnt, pool, nullptr, static_cast<double*>(xyz_buf.ptr),
                                             nullptr));
        _ctx.check(swz_copy_to_host(_ctx.get(), xyz.data(), xyz_buf.ptr, cnt * 24));
      }
      for (uint64_t j = j0; j < j1; ++j) {
        std::string name = "r";  // node names: "r" + octant digits, TilingAlgorithms.cpp:139
        AABB b = _bounds;
        for (int l = 0; l <= nl[j]; ++l) {
          const uint8_t o = get_octant_at_level(nk[j], static_cast<uint32_t>(l));
          name.push_back(static_cast<char>('0' + o));
          b = get_octant_bounds(o, b);
        }
        _persistence.persist_points(ids.data() + no[j], ids.data() + no[j] + nc[j], xyz.data() + 3 * (no[j] - no[j0]), b, name);
      }
      j0 = j1;
    }
    return static_cast<size_t>(got);
  }

  swz_tiler_info info() {
    swz_tiler_info i{};
    if (_tiler) _ctx.check(swz_tiler_get_info(_tiler, &i));
    return i;
  }
  // stored points per export chunk of finalize() (whole node files; at least the largest file)
  void set_export_chunk_points(uint64_t n) { _export_chunk_points = n; }

private:
  struct DeviceBuffer {  // device scratch through the ABI (no HIP headers needed by the host code)
    void* ptr = nullptr;
    explicit DeviceBuffer(uint64_t bytes) {
      if (swz_device_alloc(bytes, &ptr) != SWZ_OK) throw std::runtime_error{"swz_device_alloc failed"};
    }
    ~DeviceBuffer() { swz_device_free(ptr); }
  };

  Context _ctx;
  SamplingStrategy _sampling_strategy;
  PointsSink& _persistence;
  TilerMetaParameters _meta;
  swz_tiler* _tiler = nullptr;
  AABB _bounds;
  bool _finalized = false;
  uint64_t _export_chunk_points = 16u << 20;
};

// One batch over several GPUs from this one process (swz_group_*): the batch is cut into equal pieces in input
// order, one per GPU, the library exchanges the points (and their attribute columns) by level-0 octant, tiles, and
// every shard hands the files of its subtrees to the persistence; the root node's file is the concatenation of the
// shards' parts in shard order (= Morton order).  transport: 0 peer copies, 1 RCCL.
class ShardedTilingAlgorithmGPU {
public:
  ShardedTilingAlgorithmGPU(SamplingStrategy sampling_strategy, PointsSink& persistence, TilerMetaParameters meta,
                            const std::vector<int>& devices, int transport = 0)
    : _sampling_strategy(sampling_strategy), _persistence(persistence), _meta(meta), _shards(static_cast<int>(devices.size())) {
    if (swz_group_create(_shards, devices.data(), transport, &_group) != SWZ_OK)
      throw std::runtime_error{std::string("swz_group_create: ") + swz_group_last_error(nullptr)};
  }
  ~ShardedTilingAlgorithmGPU() { swz_group_destroy(_group); }
  ShardedTilingAlgorithmGPU(const ShardedTilingAlgorithmGPU&) = delete;
  ShardedTilingAlgorithmGPU& operator=(const ShardedTilingAlgorithmGPU&) = delete;

  // positions: n x 3 doubles, attributes: host columns of n rows (may be NULL).  Returns the number of node files.
  size_t tile_batch(const double* positions, const swz_attribute_columns* attributes, size_t n, const AABB& bounds) {
    swz_tile_params p{};
    p.sampler = _sampling_strategy.kind;
    p.max_points_per_node = _sampling_strategy.max_points_per_node;
    p.spacing_at_root = _meta.spacing_at_root;
    p.max_depth = _meta.max_depth;
    p.strategy = SWZ_ACCURATE;
    const double mn[3] = {bounds.min.x, bounds.min.y, bounds.min.z}, mx[3] = {bounds.max.x, bounds.max.y, bounds.max.z};
    std::vector<std::vector<Scratch>> owned(_shards);
    std::vector<double*> d_xyz(_shards, nullptr);
    std::vector<swz_attribute_columns> d_attrs(_shards);
    std::vector<uint64_t> cnt(_shards, 0);
    for (int s = 0; s < _shards; ++s) {
      const size_t lo = n * s / _shards, hi = n * (s + 1) / _shards;
      cnt[s] = hi - lo;
      swz_ctx* c = swz_group_ctx(_group, s);
      d_attrs[s] = swz_attribute_columns{};
      d_xyz[s] = static_cast<double*>(upload(owned[s], c, positions + 3 * lo, cnt[s] * 24));
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
        if (attributes && attributes->column[t]) {
          const uint32_t rb = swz_attribute_row_bytes(t);
          d_attrs[s].column[t] = upload(owned[s], c, static_cast<const char*>(attributes->column[t]) + lo * rb, cnt[s] * rb);
        }
    }
    std::vector<swz_group_result> res(_shards);
    if (swz_group_tile(_group, d_xyz.data(), attributes ? d_attrs.data() : nullptr, cnt.data(), mn, mx, &p, res.data()) != SWZ_OK)
      throw std::runtime_error{std::string("swz_group_tile: ") + swz_group_last_error(_group)};

    // node files: per shard on its device, the root's parts joined afterwards
    size_t files = 0;
    std::vector<double> root_xyz;
    std::vector<std::vector<char>> root_attr(SWZ_ATTR_COUNT);
    for (int s = 0; s < _shards; ++s) {
      const uint64_t m = res[s].num_points;
      if (!m) continue;
      swz_ctx* c = swz_group_ctx(_group, s);
      std::vector<Scratch> tmp;
      uint32_t* d_order = static_cast<uint32_t*>(alloc(tmp, c, m * 4));
      std::vector<int8_t> nl(m);
      std::vector<uint64_t> nk(m), no(m), nc(m);
      uint64_t nn = 0;
      check(c, swz_build_node_lists_device(c, res[s].d_keys, res[s].d_level, m, d_order, m, nl.data(), nk.data(), no.data(), nc.data(), &nn));
      double* d_rows = static_cast<double*>(alloc(tmp, c, m * 24));
      swz_attribute_columns d_out{}, h_out{};
      std::vector<std::vector<char>> host_cols(SWZ_ATTR_COUNT);
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
        if (res[s].attrs.column[t]) d_out.column[t] = alloc(tmp, c, m * swz_attribute_row_bytes(t));
      check(c, swz_gather_payload_device(c, res[s].d_perm, d_order, m, res[s].d_xyz, &res[s].attrs, d_rows, &d_out));
      std::vector<double> rows(m * 3);
      check(c, swz_copy_to_host(c, rows.data(), d_rows, m * 24));
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
        if (d_out.column[t]) {
          host_cols[t].resize(m * swz_attribute_row_bytes(t));
          check(c, swz_copy_to_host(c, host_cols[t].data(), d_out.column[t], host_cols[t].size()));
        }
      for (uint64_t j = 0; j < nn; ++j) {
        if (nl[j] < 0) {  // this shard's part of the root
          root_xyz.insert(root_xyz.end(), rows.begin() + 3 * no[j], rows.begin() + 3 * (no[j] + nc[j]));
          for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
            if (d_out.column[t]) {
              const uint32_t rb = swz_attribute_row_bytes(t);
              root_attr[t].insert(root_attr[t].end(), host_cols[t].begin() + no[j] * rb, host_cols[t].begin() + (no[j] + nc[j]) * rb);
            }
          continue;
        }
        std::string name = "r";
        AABB b = bounds;
        for (int l = 0; l <= nl[j]; ++l) {
          const uint8_t o = get_octant_at_level(nk[j], static_cast<uint32_t>(l));
          name.push_back(static_cast<char>('0' + o));
          b = get_octant_bounds(o, b);
        }
        for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
          h_out.column[t] = d_out.column[t] ? host_cols[t].data() + no[j] * swz_attribute_row_bytes(t) : nullptr;
        _persistence.persist_rows(nc[j], rows.data() + 3 * no[j], h_out, b, name);
        ++files;
      }
    }
    if (!root_xyz.empty()) {
      swz_attribute_columns h_out{};
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t) h_out.column[t] = root_attr[t].empty() ? nullptr : root_attr[t].data();
      _persistence.persist_rows(root_xyz.size() / 3, root_xyz.data(), h_out, bounds, "r");
      ++files;
    }
    return files;
  }

private:
  struct Scratch {
    void* ptr = nullptr;
    Scratch() = default;
    Scratch(Scratch&& o) noexcept : ptr(o.ptr) { o.ptr = nullptr; }
    Scratch(const Scratch&) = delete;
    ~Scratch() { swz_device_free(ptr); }
  };
  static void* alloc(std::vector<Scratch>& owner, swz_ctx* c, uint64_t bytes) {
    Scratch s;
    if (swz_device_alloc_on(c, bytes ? bytes : 1, &s.ptr) != SWZ_OK) throw std::runtime_error{"swz_device_alloc_on failed"};
    void* p = s.ptr;
    owner.push_back(std::move(s));
    return p;
  }
  static void check(swz_ctx* c, int status) {
    if (status != SWZ_OK) throw std::runtime_error{std::string("swz: ") + swz_last_error(c)};
  }
  static void* upload(std::vector<Scratch>& owner, swz_ctx* c, const void* src, uint64_t bytes) {
    void* d = alloc(owner, c, bytes);
    if (bytes) check(c, swz_copy_to_device(c, d, src, bytes));
    return d;
  }

  SamplingStrategy _sampling_strategy;
  PointsSink& _persistence;
  TilerMetaParameters _meta;
  int _shards;
  swz_group* _group = nullptr;
};

}  // namespace swz_host
