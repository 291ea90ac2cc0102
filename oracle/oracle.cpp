/*
 * oracle.cpp -- CPU parity oracle for the Schwarzwald tiler hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Every function restates one piece of
 * the reference (igd-geo/schwarzwald, paths relative to schwarzwald/) and cites
 * the file:line it follows.  Arithmetic forms are kept literally (order of
 * operations, float/double narrowing, truncating casts) because parity is
 * bit-exact.  Build with -ffp-contract=off and without -ffast-math / -march
 * (the reference is built for baseline x86-64, CMakeLists.txt:28, so it never
 * contracts a*b+c into an FMA).
 *
 * PARITY UNPINNED for GRID_CENTER / MIN_DISTANCE / JITTERED / tile_node control
 * flow / FAST: no reference test vector exists for them and their reference
 * sources cannot be compiled in this image (Boost, GSL, taskflow missing).
 */
#include "oracle.h"

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <functional>
#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

/* ------------------------------------------------------------------ math */
/* core/math/Vector3.h:10-165, core/math/AABB.h:10-87 */
struct V3 {
  double x = 0, y = 0, z = 0;
};
static inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
/* Vector3.h:55: x*x + y*y + z*z, left to right */
static inline double squared_length(const V3& v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
/* Vector3.h:59-62: ((*this) - p).squaredLength() */
static inline double squared_distance(const V3& a, const V3& b) { return squared_length(a - b); }

struct AABB {
  V3 min, max;
  V3 extent() const { return max - min; } /* AABB.h:25 */
  bool isInside(const V3& p) const {      /* AABB.h:27-31 */
    return (p.x >= min.x && p.x <= max.x && p.y >= min.y && p.y <= max.y && p.z >= min.z &&
            p.z <= max.z);
  }
  V3 getCenter() const { /* AABB.h:70: min + extent() / 2 */
    const V3 e = extent();
    return {min.x + e.x / 2, min.y + e.y / 2, min.z + e.z / 2};
  }
};

static inline AABB make_aabb(const double mn[3], const double mx[3]) {
  return {{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
}

struct Positions {
  const double* xyz;
  V3 at(uint32_t i) const { return {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]}; }
};

/* ------------------------------------------------------------ bit tricks */
/* core/util/stuff.h:207-221 (uint64_t overload; the narrower overloads :179-205 are the same
 * spread restricted to 2/5/10 input bits) */
static inline uint64_t expand_bits_by_3(uint64_t val) {
  val &= 0x1FFFFFull;
  val = (val | (val << 32)) & 0x00FF00000000FFFFull;
  val = (val | (val << 16)) & 0x00FF0000FF0000FFull;
  val = (val | (val << 8)) & 0xF00F00F00F00F00Full;
  val = (val | (val << 4)) & 0x30C30C30C30C30C3ull; /* octal 0303030303030303030303 */
  val = (val | (val << 2)) & 0x1249249249249249ull;
  return val;
}
/* core/util/stuff.h:223-234 */
static inline uint64_t contract_bits_by_3(uint64_t val) {
  val &= 0x1249249249249249ull;
  val = (val | (val >> 2)) & 0x30C30C30C30C30C3ull;
  val = (val | (val >> 4)) & 0xF00F00F00F00F00Full;
  val = (val | (val >> 8)) & 0x00FF0000FF0000FFull;
  val = (val | (val >> 16)) & 0x00FF00000000FFFFull;
  val = (val | (val >> 32)) & 0x00000000FFFFFFFFull;
  return val;
}
/* core/util/stuff.cpp:340-349 */
static inline uint32_t get_prev_power_of_two(uint32_t x) {
  x = x | (x >> 1);
  x = x | (x >> 2);
  x = x | (x >> 4);
  x = x | (x >> 8);
  x = x | (x >> 16);
  return x - (x >> 1);
}

/* ----------------------------------------------------------- MortonIndex */
/* core/datastructures/MortonIndex.h:80-169 with MaxLevels as a run-time value */
static inline uint64_t key_mask(uint32_t levels) {
  return (levels * 3 >= 64) ? ~0ull : ((1ull << (levels * 3)) - 1ull);
}
static inline uint64_t truncate_to_level(uint64_t key, uint32_t level, uint32_t levels) {
  const uint32_t shift = (levels - level - 1) * 3; /* MortonIndex.h:123-129 */
  return key >> shift;
}
static inline uint8_t get_octant_at_level(uint64_t key, uint32_t level, uint32_t levels) {
  const uint32_t shift = (levels - level - 1) * 3; /* MortonIndex.h:133-138 */
  return static_cast<uint8_t>((key >> shift) & 0b111);
}
static inline uint64_t set_octant_at_level(uint64_t key, uint32_t level, uint8_t octant,
                                           uint32_t levels) {
  const uint32_t shift = (levels - level - 1) * 3; /* MortonIndex.h:140-145 (ORs, never clears) */
  return key | (static_cast<uint64_t>(octant & 0b111) << shift);
}

/* -------------------------------------------------------- octree algorithms */
/* core/tiling/OctreeAlgorithms.cpp:3-18 */
static AABB get_octant_bounds(uint8_t octant, const AABB& parent) {
  const V3 e = parent.extent();
  const double min_z = (octant & 1) ? (parent.min.z + e.z / 2) : (parent.min.z);
  const double min_y = ((octant >> 1) & 1) ? (parent.min.y + e.y / 2) : (parent.min.y);
  const double min_x = ((octant >> 2) & 1) ? (parent.min.x + e.x / 2) : (parent.min.x);
  const double max_x = min_x + e.x / 2;
  const double max_y = min_y + e.y / 2;
  const double max_z = min_z + e.z / 2;
  return {{min_x, min_y, min_z}, {max_x, max_y, max_z}};
}

/* core/tiling/OctreeAlgorithms.h:104-116 */
static AABB get_bounds_from_morton_index(uint64_t key, uint32_t levels, const AABB& root,
                                         uint32_t depth) {
  AABB b = root;
  const uint32_t max_level = std::min(depth, levels);
  for (uint32_t level = 0; level < max_level; ++level)
    b = get_octant_bounds(get_octant_at_level(key, level, levels), b);
  return b;
}

/* core/tiling/OctreeAlgorithms.cpp:72-84 */
static uint8_t get_octant(const V3& p, const AABB& bounds) {
  const V3 e = bounds.extent();
  auto nx = (uint8_t)(2 * (p.x - bounds.min.x) / e.x);
  auto ny = (uint8_t)(2 * (p.y - bounds.min.y) / e.y);
  auto nz = (uint8_t)(2 * (p.z - bounds.min.z) / e.z);
  auto ix = std::min(nx, (uint8_t)1);
  auto iy = std::min(ny, (uint8_t)1);
  auto iz = std::min(nz, (uint8_t)1);
  return (uint8_t)((iz) | (iy << 1) | (ix << 2));
}

/* core/tiling/OctreeAlgorithms.h:64-87 */
static uint64_t calculate_morton_index(const V3& p, const AABB& b, uint32_t levels) {
  const V3 e = b.extent();
  const double two_pow = std::pow(2, (double)levels);
  /* operator/(T, Vector3): per-axis division, Vector3.h:131-134 */
  const V3 scale = {two_pow / e.x, two_pow / e.y, two_pow / e.z};
  const V3 d = p - b.min;
  const V3 n = {d.x * scale.x, d.y * scale.y, d.z * scale.z}; /* multiply_component_wise */
  const uint64_t lim = (1ull << levels) - 1ull;
  const uint64_t bx = std::min(static_cast<uint64_t>(n.x), lim);
  const uint64_t by = std::min(static_cast<uint64_t>(n.y), lim);
  const uint64_t bz = std::min(static_cast<uint64_t>(n.z), lim);
  const uint64_t key = expand_bits_by_3(bz) | (expand_bits_by_3(by) << 1) | (expand_bits_by_3(bx) << 2);
  return key & key_mask(levels); /* MortonIndex(Store_t) masks, MortonIndex.h:94-96 */
}

/* core/tiling/OctreeAlgorithms.h:89-102 */
static uint64_t calculate_morton_index_naive(const V3& p, const AABB& b, uint32_t levels) {
  uint64_t key = 0;
  AABB cur = b;
  for (uint32_t level = 0; level < levels; ++level) {
    const uint8_t o = get_octant(p, cur);
    key = set_octant_at_level(key, level, o, levels);
    cur = get_octant_bounds(o, cur);
  }
  return key;
}

/* IndexedPoint<MaxLevels> (Sampling.h:147-152): point_reference -> index into xyz */
struct IP {
  uint32_t idx;
  uint64_t key;
};

/* --------------------------------------------------------------- Algorithm.h */
/* util/algorithms/Algorithm.h:22-77.  pred(cur, end) -> (selected, next). */
template <typename T, typename Pred>
static T* stable_partition_with_jumps(T* begin, T* end, Pred pred) {
  const std::ptrdiff_t count = end - begin;
  if (!count) return end;
  std::vector<T> selected_buf, unselected_buf;
  selected_buf.reserve((size_t)count);
  unselected_buf.reserve((size_t)count);
  T* current = begin;
  while (current != end) {
    const std::pair<T*, T*> sn = pred(current, end);
    T* selected = sn.first;
    T* next = sn.second;
    assert(next != current);
    if (selected == next) {
      unselected_buf.insert(unselected_buf.end(), current, next);
    } else {
      unselected_buf.insert(unselected_buf.end(), current, selected);
      selected_buf.push_back(*selected);
      unselected_buf.insert(unselected_buf.end(), selected + 1, next);
    }
    current = next;
  }
  T* pivot = std::copy(selected_buf.begin(), selected_buf.end(), begin);
  std::copy(unselected_buf.begin(), unselected_buf.end(), pivot);
  return pivot;
}

/* -------------------------------------------------------------------- Node.cpp */
struct NodeStructure { /* core/tiling/Node.h:12-20 (name omitted) */
  uint64_t morton_index = 0;
  AABB bounds;
  int32_t level = -1;
  float max_spacing = 0;
  uint32_t max_depth = 0;
};

/* core/tiling/Node.cpp:37-46 */
static int32_t first_node_level_obeying_spacing(float target_spacing, const NodeStructure& root) {
  return std::max(-1, (int)std::floor(std::log2f(root.bounds.extent().x / target_spacing)) - 1);
}
/* core/tiling/Node.cpp:49-57 */
static int32_t get_node_level_to_sample_from(int32_t source_node_level, const NodeStructure& root) {
  const auto spacing_at_target_node = root.max_spacing / std::pow(2, source_node_level + 1);
  return first_node_level_obeying_spacing(spacing_at_target_node, root); /* double -> float */
}

/* core/tiling/Sampling.cpp:29-62 */
static int32_t required_morton_index_depth(int sampler, int32_t node_level, const NodeStructure& root) {
  switch (sampler) {
    case ORC_RANDOM_GRID:
    case ORC_GRID_CENTER:
      return get_node_level_to_sample_from(node_level, root);
    case ORC_MIN_DISTANCE:
      return node_level;
    case ORC_JITTERED: {
      const auto spacing_at_this_node = root.max_spacing / std::pow(2, node_level + 1);
      const auto perfect_cell_count =
        (root.bounds.extent().x / std::pow(2, node_level + 1)) / spacing_at_this_node;
      const auto actual_cell_count = get_prev_power_of_two(static_cast<uint32_t>(perfect_cell_count));
      const uint32_t levels = static_cast<uint32_t>(std::log2(actual_cell_count));
      return static_cast<int32_t>(static_cast<uint32_t>(node_level + levels));
    }
  }
  return node_level;
}

/* ------------------------------------------------------------------ SparseGrid */
/* core/datastructures/SparseGrid.{h,cpp}, GridCell.{h,cpp}: hash grid of cells of side
 * ~5*spacing; a point is accepted iff no previously accepted point in its cell or in an existing
 * neighbour cell is closer than spacing (float-squared, strict <). */
struct GridCell {
  std::vector<V3> points;
  std::vector<GridCell*> neighbours;
};
struct SparseGrid {
  int width, height, depth;
  AABB aabb;
  float squaredSpacing;
  std::unordered_map<long long, GridCell*> cells;

  SparseGrid(const AABB& box, float spacing) : aabb(box), squaredSpacing(spacing * spacing) {
    const double cellSizeFactor = 5.0; /* SparseGrid.cpp:9 */
    const V3 e = aabb.extent();
    width = (int)(e.x / (spacing * cellSizeFactor)); /* float * double -> double, :16-18 */
    height = (int)(e.y / (spacing * cellSizeFactor));
    depth = (int)(e.z / (spacing * cellSizeFactor));
  }
  ~SparseGrid() {
    for (auto& kv : cells) delete kv.second;
  }
  /* GridCell::GridCell(SparseGrid*, GridIndex&) -- GridCell.cpp:10-35 */
  GridCell* make_cell(int ci, int cj, int ck) {
    GridCell* self = new GridCell();
    self->neighbours.reserve(26);
    for (int i = std::max(ci - 1, 0); i <= std::min(width - 1, ci + 1); i++)
      for (int j = std::max(cj - 1, 0); j <= std::min(height - 1, cj + 1); j++)
        for (int k = std::max(ck - 1, 0); k <= std::min(depth - 1, ck + 1); k++) {
          const long long key = ((long long)k << 40) | ((long long)j << 20) | i;
          auto it = cells.find(key);
          if (it != cells.end()) {
            GridCell* nb = it->second;
            if (nb != self) {
              self->neighbours.push_back(nb);
              nb->neighbours.push_back(self);
            }
          }
        }
    return self;
  }
  /* GridCell::isDistant -- GridCell.cpp:43-58; squaredSpacing float is widened to double */
  static bool cell_is_distant(const GridCell* c, const V3& p, const double& sq) {
    for (const V3& q : c->points)
      if (squared_distance(p, q) < sq) return false;
    return true;
  }
  /* SparseGrid::isDistant -- SparseGrid.cpp:30-44 */
  bool isDistant(const V3& p, GridCell* cell) const {
    if (!cell_is_distant(cell, p, squaredSpacing)) return false;
    for (const GridCell* nb : cell->neighbours)
      if (!cell_is_distant(nb, p, squaredSpacing)) return false;
    return true;
  }
  /* SparseGrid::add -- SparseGrid.cpp:116-146 */
  bool add(const V3& p) {
    const V3 e = aabb.extent();
    int nx = (int)(width * (p.x - aabb.min.x) / e.x);
    int ny = (int)(height * (p.y - aabb.min.y) / e.y);
    int nz = (int)(depth * (p.z - aabb.min.z) / e.z);
    int i = std::max(0, std::min(nx, width - 1));
    int j = std::max(0, std::min(ny, height - 1));
    int k = std::max(0, std::min(nz, depth - 1));
    const long long key = ((long long)k << 40) | ((long long)j << 20) | (long long)i;
    auto it = cells.find(key);
    if (it == cells.end()) it = cells.emplace(key, make_cell(i, j, k)).first;
    if (isDistant(p, it->second)) {
      it->second->points.push_back(p);
      return true;
    }
    return false;
  }
};

/* ---------------------------------------------------------------- samplers */
#define SWZ_JITTER_TABLE(W) static const uint8_t PERMUTATIONS_##W[16 * W]
#include "jitter_tables.inc"
#undef SWZ_JITTER_TABLE

struct SampleCtx {
  int sampler;
  uint64_t max_points_per_node;
  Positions pos;
  uint32_t levels; /* MaxLevels of the keys */
};

static bool take_all(const SampleCtx& c, int behaviour, uint64_t n) {
  /* Sampling.h:201-208 (identical in :328-335, :435-442, :612-619) */
  return behaviour == ORC_TAKE_ALL_WHEN_BELOW_MAX && n <= c.max_points_per_node;
}

/* candidate_level_in_octree -- Sampling.h:210-229 / :337-343 */
static int candidate_level(const AABB& root_bounds, float spacing_at_root, int32_t node_level) {
  const auto spacing_at_this_node = spacing_at_root / std::pow(2, node_level + 1);
  return std::max(-1, (int)std::floor(std::log2f(root_bounds.extent().x / spacing_at_this_node)) - 1);
}

/* RandomSortedGridSampling::sample_points -- Sampling.h:187-308 */
static int64_t sample_random_grid(const SampleCtx& c, IP* begin, IP* end, int32_t node_level,
                                  const AABB& root_bounds, float spacing_at_root, int behaviour) {
  const uint64_t n = (uint64_t)(end - begin);
  if (take_all(c, behaviour, n)) return (int64_t)n;
  const int cand = candidate_level(root_bounds, spacing_at_root, node_level);
  if (cand == -1) return (begin == end) ? 0 : 1; /* partition_at_root :290-295 */
  if (cand >= (int)c.levels) return ORC_ERR_BAD_ARG; /* truncate_to_level would assert */
  const uint32_t level = (uint32_t)cand, levels = c.levels;
  IP* pp = stable_partition_with_jumps(begin, end, [level, levels](IP* cur, IP* e) {
    IP* taken = cur;
    const uint64_t cell = truncate_to_level(taken->key, level, levels);
    IP* next = std::partition_point(taken + 1, e, [cell, level, levels](const IP& o) {
      return truncate_to_level(o.key, level, levels) <= cell;
    });
    return std::make_pair(taken, next);
  });
  return pp - begin;
}

/* GridCenterSampling::sample_points -- Sampling.h:314-416 */
static int64_t sample_grid_center(const SampleCtx& c, IP* begin, IP* end, int32_t node_level,
                                  const AABB& root_bounds, float spacing_at_root, int behaviour) {
  const uint64_t n = (uint64_t)(end - begin);
  if (take_all(c, behaviour, n)) return (int64_t)n;
  const int cand = candidate_level(root_bounds, spacing_at_root, node_level);
  if (cand == -1) return 1; /* :346-348 returns ++partition_point unconditionally */
  if (cand >= (int)c.levels) return ORC_ERR_BAD_ARG;
  const uint32_t level = (uint32_t)cand, levels = c.levels;
  const Positions& pos = c.pos;
  IP* pp = stable_partition_with_jumps(begin, end, [&](IP* cur, IP* e) {
    const uint64_t cell = truncate_to_level(cur->key, level, levels);
    IP* same_cell_end = std::partition_point(cur + 1, e, [cell, level, levels](const IP& o) {
      return truncate_to_level(o.key, level, levels) <= cell;
    });
    const AABB cell_bounds = get_bounds_from_morton_index(cur->key, levels, root_bounds, level + 1);
    const V3 center = cell_bounds.getCenter();
    IP* min_point = std::min_element(cur, same_cell_end, [&](const IP& l, const IP& r) {
      return squared_distance(pos.at(l.idx), center) < squared_distance(pos.at(r.idx), center);
    });
    return std::make_pair(min_point, same_cell_end);
  });
  return pp - begin;
}

/* PoissonDiskSampling::sample_points (MIN_DISTANCE) -- Sampling.h:421-471 */
static int64_t sample_min_distance(const SampleCtx& c, IP* begin, IP* end, uint64_t node_key,
                                   int32_t node_level, const AABB& root_bounds,
                                   float spacing_at_root, int behaviour) {
  const uint64_t n = (uint64_t)(end - begin);
  if (take_all(c, behaviour, n)) return (int64_t)n;
  const AABB bounds_at_this_node =
    get_bounds_from_morton_index(node_key, c.levels, root_bounds, (uint32_t)(node_level + 1));
  const auto spacing_at_this_node = spacing_at_root / std::pow(2, node_level + 1);
  SparseGrid grid{bounds_at_this_node, static_cast<float>(spacing_at_this_node)};
  const Positions& pos = c.pos;
  IP* pp = std::stable_partition(begin, end, [&](const IP& p) { return grid.add(pos.at(p.idx)); });
  return pp - begin;
}

/* OctreeNodeIndex64::to_grid_index -- core/datastructures/OctreeNodeIndex.h:357-363 */
static void to_grid_index(uint64_t index, uint32_t levels, uint64_t out[3]) {
  const uint64_t m = (uint64_t)((1 << levels) - 1);
  out[2] = contract_bits_by_3(index) & m;
  out[1] = contract_bits_by_3(index >> 1) & m;
  out[0] = contract_bits_by_3(index >> 2) & m;
}

/* JitteredSampling::sample_points -- Sampling.h:598-759 */
static int64_t sample_jittered(const SampleCtx& c, IP* begin, IP* end, uint64_t node_key,
                               int32_t node_level, const AABB& root_bounds, float spacing_at_root,
                               int behaviour) {
  const uint64_t n = (uint64_t)(end - begin);
  if (take_all(c, behaviour, n)) return (int64_t)n;
  const AABB bounds_at_this_node =
    get_bounds_from_morton_index(node_key, c.levels, root_bounds, (uint32_t)(node_level + 1));
  const auto spacing_at_this_node = spacing_at_root / std::pow(2, node_level + 1);
  const auto perfect_cell_count = bounds_at_this_node.extent().x / spacing_at_this_node;
  const auto actual_cell_count = get_prev_power_of_two(static_cast<uint32_t>(perfect_cell_count));
  if (actual_cell_count < 16) return ORC_ERR_JITTER_GRID_TOO_SMALL;
  const uint32_t levels = static_cast<uint32_t>(std::log2(actual_cell_count));
  const uint32_t grid_level = static_cast<uint32_t>(node_level + levels);
  if (grid_level >= 21 /* MortonIndex64Levels */ || grid_level >= c.levels)
    return ORC_ERR_JITTER_NODE_TOO_DEEP;
  const uint64_t grid_mask = (1ull << (3 * levels)) - 1ull;
  const double grid_cell_size = bounds_at_this_node.extent().x / actual_cell_count;
  const double permutation_cell_size = grid_cell_size / actual_cell_count;

  const uint32_t start_index = (3 * static_cast<uint32_t>(node_level + 1)) % 16;
  const uint8_t* table;
  uint32_t width;
  if (actual_cell_count <= 16) {
    table = PERMUTATIONS_16;
    width = 16;
  } else if (actual_cell_count <= 32) {
    table = PERMUTATIONS_32;
    width = 32;
  } else {
    table = PERMUTATIONS_64;
    width = 64;
  }
  const uint8_t* permutations[3] = {table + (size_t)start_index * width,
                                    table + (size_t)((start_index + 1) % 16) * width,
                                    table + (size_t)((start_index + 2) % 16) * width};
  const uint32_t permutation_length = std::min<uint32_t>(actual_cell_count, 64);
  const uint32_t key_levels = c.levels;
  const Positions& pos = c.pos;

  IP* pp = stable_partition_with_jumps(begin, end, [&](IP* cur, IP* e) {
    const uint64_t rel = truncate_to_level(cur->key, grid_level, key_levels);
    uint64_t g[3];
    to_grid_index(rel & grid_mask, levels, g);
    IP* next_cell = std::partition_point(cur + 1, e, [rel, grid_level, key_levels](const IP& o) {
      return truncate_to_level(o.key, grid_level, key_levels) <= rel;
    });
    const uint32_t px = (uint32_t)permutations[0][(g[1] + g[2]) % permutation_length] - 1;
    const uint32_t py = (uint32_t)permutations[1][(g[0] + g[2]) % permutation_length] - 1;
    const uint32_t pz = (uint32_t)permutations[2][(g[0] + g[1]) % permutation_length] - 1;
    const V3 target = bounds_at_this_node.min + V3{g[0] * grid_cell_size + px * permutation_cell_size,
                                                   g[1] * grid_cell_size + py * permutation_cell_size,
                                                   g[2] * grid_cell_size + pz * permutation_cell_size};
    IP* min_point = std::min_element(cur, next_cell, [&](const IP& l, const IP& r) {
      return squared_distance(pos.at(l.idx), target) < squared_distance(pos.at(r.idx), target);
    });
    return std::make_pair(min_point, next_cell);
  });
  return pp - begin;
}

/* sample_points -- Sampling.h:799-821 */
static int64_t sample_points(const SampleCtx& c, IP* begin, IP* end, uint64_t node_key,
                             int32_t node_level, const AABB& root_bounds, float spacing_at_root,
                             int behaviour) {
  switch (c.sampler) {
    case ORC_RANDOM_GRID:
      return sample_random_grid(c, begin, end, node_level, root_bounds, spacing_at_root, behaviour);
    case ORC_GRID_CENTER:
      return sample_grid_center(c, begin, end, node_level, root_bounds, spacing_at_root, behaviour);
    case ORC_MIN_DISTANCE:
      return sample_min_distance(c, begin, end, node_key, node_level, root_bounds, spacing_at_root,
                                 behaviour);
    case ORC_JITTERED:
      return sample_jittered(c, begin, end, node_key, node_level, root_bounds, spacing_at_root,
                             behaviour);
  }
  return ORC_ERR_BAD_ARG;
}

/* partition_points_into_child_octants -- OctreeAlgorithms.h:240-265 */
static void partition_points_into_child_octants(const IP* begin, const IP* end, uint32_t level,
                                                uint32_t levels, const IP* bounds_out[9]) {
  const IP* cur = begin;
  for (uint8_t octant = 0; octant < 8; ++octant) {
    const IP* cur_end = std::find_if(cur, end, [octant, level, levels](const IP& p) {
      return get_octant_at_level(p.key, level, levels) > octant;
    });
    bounds_out[octant] = cur;
    cur = cur_end;
  }
  bounds_out[8] = cur;
}

/* ------------------------------------------------------------------ tiler */
constexpr uint32_t MAX_OCTREE_LEVELS = 21; /* TilingAlgorithms.cpp:20 */

/* Worker threads for node tasks, standing in for the reference's taskflow executor (Scheduler.cpp:49-61):
 * children with at least MIN_POINTS_FOR_ASYNC_PROCESSING points become tasks, the others are tiled by the
 * thread that produced them (TilingAlgorithms.cpp:25, 499-561). */
constexpr size_t MIN_POINTS_FOR_ASYNC_PROCESSING = 100000;
struct TaskPool {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> queue;
  size_t unfinished = 0;
  bool stop = false;
  std::vector<std::thread> workers;
  explicit TaskPool(unsigned threads) {
    for (unsigned i = 0; i < threads; ++i) workers.emplace_back([this] { run(); });
  }
  ~TaskPool() {
    {
      std::lock_guard<std::mutex> g(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& w : workers) w.join();
  }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> g(m);
      queue.push_back(std::move(f));
      ++unfinished;
    }
    cv.notify_one();
  }
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [this] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        f = std::move(queue.front());
        queue.pop_front();
      }
      f();
      {
        std::lock_guard<std::mutex> g(m);
        --unfinished;
      }
      cv.notify_all();
    }
  }
  void wait_all() {
    std::unique_lock<std::mutex> g(m);
    cv.wait(g, [this] { return unfinished == 0; });
  }
};

struct Tiler {
  SampleCtx ctx;
  orc_tile_params params;
  /* outputs, indexed by sorted position */
  const std::vector<IP>* sorted = nullptr;
  int8_t* level_out = nullptr;
  orc_tile_stats stats{};
  std::atomic<int32_t> error{ORC_OK};
  std::vector<uint32_t> pos_of_idx; /* original index -> sorted position */
  TaskPool* pool = nullptr;         /* NULL: everything on the calling thread */
  std::mutex stats_mutex;

  void persist(const IP* b, const IP* e, int32_t level) {
    for (const IP* p = b; p != e; ++p) level_out[pos_of_idx[p->idx]] = (int8_t)level;
    std::lock_guard<std::mutex> g(stats_mutex);
    stats.num_nodes += 1;
    if (e != b) stats.max_level = std::max(stats.max_level, level);
  }

  /* do_tiling_for_node + tile_node + tile_internal_node + tile_terminal_node +
   * split_range_into_child_nodes -- TilingAlgorithms.cpp:499-561, 351-492, 247-349, 206-241,
   * 116-162, for a single batch with an initially empty, lossless persistence (no cached points,
   * previously_taken_points_count == 0). */
  void do_tiling_for_node(std::vector<IP>&& node_data, const NodeStructure& node,
                          const NodeStructure& root) {
    if (error) return;
    {
      std::lock_guard<std::mutex> g(stats_mutex);
      stats.points_visited += node_data.size();
    }
    const int32_t req = required_morton_index_depth(ctx.sampler, node.level, root);
    const bool requires_deeper = req > node.level;
    const int32_t max_level = (int32_t)std::min(MAX_OCTREE_LEVELS - 1, node.max_depth);
    if (!requires_deeper) {
      if (req >= max_level) { /* :421-427 terminal */
        persist(node_data.data(), node_data.data() + node_data.size(), node.level);
        return;
      }
    } else {
      if (node.level >= max_level) { /* :436-442 terminal */
        persist(node_data.data(), node_data.data() + node_data.size(), node.level);
        return;
      }
      if (req >= (int32_t)MAX_OCTREE_LEVELS) { /* :444-483 re-root: out of scope */
        error = ORC_ERR_REROOT_UNSUPPORTED;
        return;
      }
    }
    /* tile_internal_node :247-349 */
    IP* b = node_data.data();
    IP* e = b + node_data.size();
    const int32_t rel_level = node.level - (root.level + 1); /* :277 */
    const int64_t taken = sample_points(ctx, b, e, node.morton_index, rel_level, root.bounds,
                                        root.max_spacing, ORC_TAKE_ALL_WHEN_BELOW_MAX);
    if (taken < 0) {
      error = (int32_t)taken;
      return;
    }
    persist(b, b + taken, node.level);
    /* split_range_into_child_nodes :116-162 */
    const int32_t child_level = node.level + 1;
    const IP* ranges[9];
    partition_points_into_child_octants(b + taken, e, (uint32_t)child_level, MAX_OCTREE_LEVELS, ranges);
    for (uint8_t octant = 0; octant < 8; ++octant) {
      if (ranges[octant + 1] == ranges[octant]) continue;
      NodeStructure child = node;
      child.morton_index = set_octant_at_level(child.morton_index, (uint32_t)child_level, octant,
                                               MAX_OCTREE_LEVELS);
      child.bounds = get_octant_bounds(octant, node.bounds);
      child.level = child_level;
      child.max_spacing /= 2;
      std::vector<IP> data(ranges[octant], ranges[octant + 1]);
      if (pool && data.size() >= MIN_POINTS_FOR_ASYNC_PROCESSING) {
        auto shared = std::make_shared<std::vector<IP>>(std::move(data));
        pool->submit([this, shared, child, root] { do_tiling_for_node(std::move(*shared), child, root); });
      } else {
        do_tiling_for_node(std::move(data), child, root);
      }
    }
  }
};

/* estimate_start_node_level_in_octree -- TilingAlgorithms.cpp:1473-1535 */
static size_t estimate_start_node_level(const std::vector<IP>& sorted, size_t concurrency) {
  using R = std::pair<const IP*, const IP*>;
  std::vector<R> splits{{sorted.data(), sorted.data() + sorted.size()}};
  constexpr uint32_t MIN_LEVEL = 3, MAX_LEVEL = 6;
  constexpr float MIN_SCORE = 1.f;
  for (uint32_t level = 0; level < MAX_LEVEL; ++level) {
    std::vector<R> next;
    for (const R& r : splits) {
      const IP* parts[9];
      partition_points_into_child_octants(r.first, r.second, level, MAX_OCTREE_LEVELS, parts);
      for (int o = 0; o < 8; ++o)
        if (parts[o + 1] != parts[o]) next.push_back({parts[o], parts[o + 1]});
    }
    splits.swap(next);
    float score = 0.f;
    if (!(splits.size() <= concurrency / 2)) {
      const auto large = std::count_if(splits.begin(), splits.end(), [](const R& r) {
        return (size_t)(r.second - r.first) >= 100000;
      });
      score = static_cast<float>(large) / static_cast<float>(concurrency);
    }
    if (score >= MIN_SCORE) return std::max(level + 1, MIN_LEVEL);
  }
  return MAX_LEVEL;
}

} // namespace

/* ================================================================== C API */
extern "C" {

uint64_t orc_calculate_morton_index(const double p[3], const double bmin[3], const double bmax[3],
                                    uint32_t levels) {
  return calculate_morton_index({p[0], p[1], p[2]}, make_aabb(bmin, bmax), levels);
}
uint64_t orc_calculate_morton_index_naive(const double p[3], const double bmin[3],
                                          const double bmax[3], uint32_t levels) {
  return calculate_morton_index_naive({p[0], p[1], p[2]}, make_aabb(bmin, bmax), levels);
}

void orc_index_points(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                      uint32_t levels, uint64_t* keys_out) {
  const AABB b = make_aabb(bmin, bmax);
  for (uint64_t i = 0; i < n; ++i) {
    /* index_point, OctreeAlgorithms.h:145-175 with ClampToBounds */
    V3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (!b.isInside(p)) {
      p.x = std::min(b.max.x, std::max(b.min.x, p.x));
      p.y = std::min(b.max.y, std::max(b.min.y, p.y));
      p.z = std::min(b.max.z, std::max(b.min.z, p.z));
      xyz[3 * i] = p.x;
      xyz[3 * i + 1] = p.y;
      xyz[3 * i + 2] = p.z;
    }
    keys_out[i] = calculate_morton_index(p, b, levels);
  }
}

void orc_sort_by_key(const uint64_t* keys, uint64_t n, uint32_t* perm_out) {
  for (uint64_t i = 0; i < n; ++i) perm_out[i] = (uint32_t)i;
  std::stable_sort(perm_out, perm_out + n, [keys](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
}

void orc_get_octant_bounds(uint8_t octant, const double bmin[3], const double bmax[3],
                           double omin[3], double omax[3]) {
  const AABB o = get_octant_bounds(octant, make_aabb(bmin, bmax));
  omin[0] = o.min.x; omin[1] = o.min.y; omin[2] = o.min.z;
  omax[0] = o.max.x; omax[1] = o.max.y; omax[2] = o.max.z;
}

void orc_get_bounds_from_morton_index(uint64_t key, uint32_t levels, const double bmin[3],
                                      const double bmax[3], uint32_t depth, double omin[3],
                                      double omax[3]) {
  const AABB o = get_bounds_from_morton_index(key, levels, make_aabb(bmin, bmax), depth);
  omin[0] = o.min.x; omin[1] = o.min.y; omin[2] = o.min.z;
  omax[0] = o.max.x; omax[1] = o.max.y; omax[2] = o.max.z;
}

void orc_partition_points_into_child_octants(const uint64_t* sorted_keys, uint64_t n,
                                             uint32_t level, uint32_t levels, uint64_t offsets[9]) {
  std::vector<IP> pts(n);
  for (uint64_t i = 0; i < n; ++i) pts[i] = {(uint32_t)i, sorted_keys[i]};
  const IP* r[9];
  partition_points_into_child_octants(pts.data(), pts.data() + n, level, levels, r);
  for (int o = 0; o < 9; ++o) offsets[o] = (uint64_t)(r[o] - pts.data());
}

uint64_t orc_truncate_to_level(uint64_t key, uint32_t level, uint32_t levels) {
  return truncate_to_level(key, level, levels);
}
uint8_t orc_get_octant_at_level(uint64_t key, uint32_t level, uint32_t levels) {
  return get_octant_at_level(key, level, levels);
}
uint64_t orc_set_octant_at_level(uint64_t key, uint32_t level, uint8_t octant, uint32_t levels) {
  return set_octant_at_level(key, level, octant, levels);
}
void orc_to_grid_index(uint64_t index, uint32_t levels, uint64_t out_xyz[3]) {
  to_grid_index(index, levels, out_xyz);
}
uint32_t orc_get_prev_power_of_two(uint32_t x) { return get_prev_power_of_two(x); }

int32_t orc_required_morton_index_depth(int sampler, int32_t node_level, const double root_min[3],
                                        const double root_max[3], float root_max_spacing) {
  NodeStructure root;
  root.bounds = make_aabb(root_min, root_max);
  root.level = -1;
  root.max_spacing = root_max_spacing;
  return required_morton_index_depth(sampler, node_level, root);
}

int64_t orc_sample_points(int sampler, uint64_t max_points_per_node, uint64_t* keys, uint32_t* idx,
                          uint64_t n, const double* xyz, uint64_t node_key, int32_t node_level,
                          uint32_t levels, const double root_min[3], const double root_max[3],
                          float spacing_at_root, int behaviour) {
  std::vector<IP> pts(n);
  for (uint64_t i = 0; i < n; ++i) pts[i] = {idx[i], keys[i]};
  SampleCtx c{sampler, max_points_per_node, Positions{xyz}, levels};
  const int64_t taken = sample_points(c, pts.data(), pts.data() + n, node_key, node_level,
                                      make_aabb(root_min, root_max), spacing_at_root, behaviour);
  if (taken < 0) return taken;
  for (uint64_t i = 0; i < n; ++i) {
    keys[i] = pts[i].key;
    idx[i] = pts[i].idx;
  }
  return taken;
}

void orc_sparse_grid_greedy(const double* xyz, const uint32_t* idx, uint64_t n, const double nmin[3],
                            const double nmax[3], float spacing, uint8_t* accepted) {
  SparseGrid grid{make_aabb(nmin, nmax), spacing};
  Positions pos{xyz};
  for (uint64_t i = 0; i < n; ++i) accepted[i] = grid.add(pos.at(idx[i])) ? 1 : 0;
}

int32_t orc_tile(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                 const orc_tile_params* params, uint64_t* keys_out, uint32_t* perm_out,
                 int8_t* level_out, uint32_t* dup_mask_out, orc_tile_stats* stats_out) {
  return orc_tile_mt(xyz, n, bmin, bmax, params, 1, keys_out, perm_out, level_out, dup_mask_out, stats_out);
}

static thread_local double* g_stage_seconds = nullptr; /* orc_tile_mt_timed: index, sort, tiling */
static double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int32_t orc_tile_mt_timed(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                          const orc_tile_params* params, uint32_t threads, uint64_t* keys_out, uint32_t* perm_out,
                          int8_t* level_out, uint32_t* dup_mask_out, orc_tile_stats* stats_out, double stage_seconds[3]) {
  g_stage_seconds = stage_seconds;
  const int32_t st = orc_tile_mt(xyz, n, bmin, bmax, params, threads, keys_out, perm_out, level_out, dup_mask_out, stats_out);
  g_stage_seconds = nullptr;
  return st;
}

int32_t orc_tile_mt(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                    const orc_tile_params* params, uint32_t threads, uint64_t* keys_out, uint32_t* perm_out,
                    int8_t* level_out, uint32_t* dup_mask_out, orc_tile_stats* stats_out) {
  if (!params || n > 0xFFFFFFFFull) return ORC_ERR_BAD_ARG;
  double* stage = g_stage_seconds;
  const double t_begin = now_seconds();
  if (threads == 0) threads = 1;
  const AABB bounds = make_aabb(bmin, bmax);
  /* index (V1 :588-598 / V3 :1262-1285): chunks of the batch on the indexing threads (Parallel.h:172-213) */
  std::vector<uint64_t> keys(n);
  if (threads == 1 || n < 2 * MIN_POINTS_FOR_ASYNC_PROCESSING) {
    orc_index_points(xyz, n, bmin, bmax, MAX_OCTREE_LEVELS, keys.data());
  } else {
    std::vector<std::thread> ts;
    for (uint32_t t = 0; t < threads; ++t) {
      const uint64_t b = n * t / threads, e = n * (t + 1) / threads;
      ts.emplace_back([=, &keys] { orc_index_points(xyz + 3 * b, e - b, bmin, bmax, MAX_OCTREE_LEVELS, keys.data() + b); });
    }
    for (auto& th : ts) th.join();
  }
  const double t_indexed = now_seconds();
  /* sort (V1 :600-604 / V3 :1292), canonical tie order */
  std::vector<uint32_t> perm(n);
  orc_sort_by_key(keys.data(), n, perm.data());
  const double t_sorted = now_seconds();
  if (stage) {
    stage[0] = t_indexed - t_begin;
    stage[1] = t_sorted - t_indexed;
    stage[2] = 0.0;
  }
  std::vector<IP> sorted(n);
  for (uint64_t i = 0; i < n; ++i) sorted[i] = {perm[i], keys[perm[i]]};
  for (uint64_t i = 0; i < n; ++i) {
    keys_out[i] = sorted[i].key;
    perm_out[i] = sorted[i].idx;
    level_out[i] = (int8_t)-128; /* "not persisted" sentinel, must not survive */
  }
  if (dup_mask_out) std::memset(dup_mask_out, 0, n * sizeof(uint32_t));

  Tiler t;
  t.ctx = SampleCtx{params->sampler, params->max_points_per_node, Positions{xyz}, MAX_OCTREE_LEVELS};
  t.params = *params;
  t.sorted = &sorted;
  t.level_out = level_out;
  t.stats.max_level = -1;
  t.stats.fast_start_levels = -1;
  t.pos_of_idx.resize(n);
  for (uint64_t i = 0; i < n; ++i) t.pos_of_idx[sorted[i].idx] = (uint32_t)i;

  NodeStructure root; /* V1 :606-612, V3 :1327-1333 */
  root.bounds = bounds;
  root.level = -1;
  root.max_depth = params->max_depth;
  root.max_spacing = params->spacing_at_root;
  root.morton_index = 0;

  if (n == 0) {
    if (stats_out) *stats_out = t.stats;
    return ORC_OK;
  }

  std::unique_ptr<TaskPool> pool;
  if (threads > 1) {
    pool.reset(new TaskPool(threads));
    t.pool = pool.get();
  }
  if (params->strategy == ORC_ACCURATE) {
    /* one task for the root; the whole-batch sort above and the root node run on one thread like
     * TilingAlgorithmV1's graph (:600-626) */
    t.do_tiling_for_node(std::vector<IP>(sorted), root, root);
    if (pool) pool->wait_all();
  } else {
    /* V3 first iteration :1287-1353 */
    const size_t S = estimate_start_node_level(sorted, params->fast_concurrency);
    t.stats.fast_start_levels = (int32_t)S;
    /* split_indexed_points_into_subranges :1537-1578: all non-empty prefixes with S octants */
    const uint32_t shift = (MAX_OCTREE_LEVELS - (uint32_t)S) * 3;
    uint64_t i = 0;
    while (i < n && !t.error) {
      const uint64_t prefix = sorted[i].key >> shift;
      uint64_t j = i;
      while (j < n && (sorted[j].key >> shift) == prefix) ++j;
      NodeStructure node; /* :1335-1343 */
      node.level = (int32_t)S - 1;
      node.max_depth = root.max_depth;
      node.max_spacing = (float)(root.max_spacing / std::pow(2, (double)S));
      node.morton_index = prefix << shift; /* to_static_morton_index, OctreeNodeIndex.h:347-351 */
      node.bounds = get_bounds_from_morton_index(node.morton_index, MAX_OCTREE_LEVELS, bounds, (uint32_t)S);
      if (pool) { /* the start nodes are independent tasks of V3's graph (:1345-1353) */
        auto data = std::make_shared<std::vector<IP>>(sorted.begin() + i, sorted.begin() + j);
        Tiler* tp = &t;
        pool->submit([tp, data, node, root] { tp->do_tiling_for_node(std::move(*data), node, root); });
      } else {
        t.do_tiling_for_node(std::vector<IP>(sorted.begin() + i, sorted.begin() + j), node, root);
      }
      i = j;
    }
    if (pool) pool->wait_all();
    t.pool = nullptr; /* the reconstruction below is sequential (finalize, :1717-1784) */
    /* finalize -> reconstruct_left_out_nodes :1717-1784, reconstruct_single_node :1661-1715.
     * stored[levels][index] = sorted positions persisted under that node, Morton order. */
    if (!t.error) {
      std::vector<std::map<uint64_t, std::vector<uint32_t>>> stored(S + 1);
      for (uint64_t p = 0; p < n; ++p)
        if (level_out[p] == (int8_t)(S - 1)) stored[S][sorted[p].key >> shift].push_back((uint32_t)p);
      for (size_t lv = S; lv-- > 0 && !t.error;) { /* node.levels() == lv, deepest first */
        /* every parent of an existing node one level below is reconstructed */
        std::map<uint64_t, std::vector<uint32_t>> parents;
        for (const auto& kv : stored[lv + 1]) parents[kv.first >> 3]; /* children iterate 0..7 via map order */
        for (auto& pkv : parents) {
          const uint64_t index = pkv.first;
          std::vector<IP> data;
          for (uint8_t octant = 0; octant < 8; ++octant) {
            auto it = stored[lv + 1].find((index << 3) | octant);
            if (it == stored[lv + 1].end()) continue;
            for (uint32_t p : it->second) data.push_back(sorted[p]);
          }
          const uint64_t node_key = (lv == 0) ? 0 : (index << ((MAX_OCTREE_LEVELS - (uint32_t)lv) * 3));
          const int64_t taken =
            sample_points(t.ctx, data.data(), data.data() + data.size(), node_key, (int32_t)lv - 1,
                          bounds, params->spacing_at_root, ORC_ALWAYS_ADHERE);
          if (taken < 0) {
            t.error = (int32_t)taken;
            break;
          }
          std::vector<uint32_t>& sel = stored[lv][index];
          for (int64_t q = 0; q < taken; ++q) {
            const uint32_t p = t.pos_of_idx[data[q].idx];
            sel.push_back(p);
            if (dup_mask_out) dup_mask_out[p] |= (1u << lv); /* level lv-1 -> bit (level+1) */
          }
          t.stats.num_nodes += 1;
        }
      }
    }
  }
  if (stats_out) *stats_out = t.stats;
  if (stage) stage[2] = now_seconds() - t_sorted;
  return t.error;
}

int64_t orc_stable_partition_take_multiples(int32_t* values, int64_t n, int32_t modulus) {
  /* the predicate used by test/TestAlgorithm.cpp:24-80 */
  auto is_match = [modulus](int32_t v) { return (v % modulus) == 0; };
  int32_t* pivot = stable_partition_with_jumps(values, values + n, [&](int32_t* cur, int32_t* e) {
    if (!is_match(*cur)) {
      int32_t* m = std::find_if(cur + 1, e, is_match);
      if (m == e) return std::make_pair(e, e);
      return std::make_pair(m, m + 1);
    }
    return std::make_pair(cur, cur + 1);
  });
  return pivot - values;
}

void orc_merge_ranges_i32(const int32_t* const* ranges, const int64_t* sizes, int64_t num_ranges,
                          int32_t* out) {
  /* merge_ranges -- util/algorithms/Algorithm.h:111-150: repeatedly take the lowest head, the first
   * range winning ties (comparator(*iter, *lowest) strict) */
  std::vector<std::pair<const int32_t*, const int32_t*>> heads;
  int64_t total = 0;
  for (int64_t r = 0; r < num_ranges; ++r) {
    heads.push_back({ranges[r], ranges[r] + sizes[r]});
    total += sizes[r];
  }
  for (int64_t o = 0; o < total; ++o) {
    int lowest = -1;
    for (size_t r = 0; r < heads.size(); ++r) {
      if (heads[r].first == heads[r].second) continue;
      if (lowest < 0 || *heads[r].first < *heads[(size_t)lowest].first) lowest = (int)r;
    }
    if (lowest < 0) return;
    out[o] = *heads[(size_t)lowest].first++;
  }
}

/* ==================================================================================================
 * Multi-batch tiler: what Tiler does when the input is larger than
 * internal_cache_size (executable/main.cpp:233-236, default 10 M points): build_execution_graph is called
 * once per batch on the SAME TilingAlgorithm object and every node re-reads what earlier batches persisted
 * under its name.  The persistence restated here is BinaryPersistence (core/io/BinaryPersistence.h:45-57:
 * one file per node name, opened with std::ios::out, i.e. REPLACED by every persist_points call; a call with
 * zero points returns before opening the file; is_lossless() == true).  (MemoryPersistence.h:23-32 appends
 * instead and would duplicate the cached points; it is "mostly for unit testing".)
 * PARITY UNPINNED: the reference's only multi-batch tests (test/TestTiler.cpp:85-190) are commented out.
 * ================================================================================================== */
namespace {

struct MBTiler {
  AABB bounds;
  orc_tile_params params;
  std::vector<double> xyz;                                           /* clamped positions by global id */
  std::map<std::pair<int32_t, uint64_t>, std::vector<uint32_t>> files; /* (node level, node morton index) -> ids */
  int32_t level_of_start_nodes = -1;                                  /* V3::_level_of_start_nodes */
  uint64_t batches = 0;
  bool finalized = false;
  int32_t error = ORC_OK;
  orc_tile_stats stats{};
  uint64_t unsorted_cached_nodes = 0;

  SampleCtx ctx() const { return SampleCtx{params.sampler, params.max_points_per_node, Positions{xyz.data()}, MAX_OCTREE_LEVELS}; }
  V3 pos(uint32_t id) const { return {xyz[3 * (size_t)id], xyz[3 * (size_t)id + 1], xyz[3 * (size_t)id + 2]}; }

  /* PointsPersistence::persist_points, BinaryPersistence.h:46-57 */
  void persist(const IP* b, const IP* e, const NodeStructure& node) {
    if (b == e) return;
    std::vector<uint32_t>& f = files[{node.level, node.morton_index}];
    f.clear();
    for (const IP* p = b; p != e; ++p) f.push_back(p->idx);
    stats.num_nodes += 1;
    stats.max_level = std::max(stats.max_level, node.level);
  }

  /* read_pnts_from_disk -- TilingAlgorithms.cpp:50-109: the cached points keep the node's own Morton index
   * and get the levels below it from an index computed relative to the NODE's bounds (no clamp); a lossless
   * persistence is not sorted again (:103-106). */
  std::vector<IP> read_pnts_from_disk(const NodeStructure& node) {
    std::vector<IP> out;
    auto it = files.find({node.level, node.morton_index});
    if (it == files.end() || it->second.empty()) return out;
    out.reserve(it->second.size());
    const uint32_t start_level = static_cast<uint32_t>(node.level + 1);
    for (uint32_t id : it->second) {
      uint64_t idx = node.morton_index;
      const uint64_t rel = calculate_morton_index(pos(id), node.bounds, MAX_OCTREE_LEVELS);
      for (uint32_t level = start_level; level < MAX_OCTREE_LEVELS; ++level)
        idx = set_octant_at_level(idx, level, get_octant_at_level(rel, level - start_level, MAX_OCTREE_LEVELS),
                                  MAX_OCTREE_LEVELS);
      out.push_back({id, idx});
    }
    for (size_t i = 1; i < out.size(); ++i)
      if (out[i].key < out[i - 1].key) {
        ++unsorted_cached_nodes; /* bookkeeping for the tests only: the reference does not look */
        break;
      }
    return out;
  }

  /* octree::merge_node_data_sorted / _unsorted -- core/tiling/Node.cpp:4-35 */
  static std::vector<IP> merge_sorted(std::vector<IP>&& first, std::vector<IP>&& second) {
    if (first.empty()) return std::move(second);
    if (second.empty()) return std::move(first);
    std::vector<IP> merged;
    merged.reserve(first.size() + second.size());
    std::merge(first.begin(), first.end(), second.begin(), second.end(), std::back_inserter(merged),
               [](const IP& l, const IP& r) { return l.key < r.key; });
    return merged;
  }
  static std::vector<IP> merge_unsorted(std::vector<IP>&& first, std::vector<IP>&& second) {
    if (first.empty()) return std::move(second);
    if (second.empty()) return std::move(first);
    first.insert(first.end(), second.begin(), second.end());
    return std::move(first);
  }

  /* tile_internal_node -- TilingAlgorithms.cpp:247-349, then split_range_into_child_nodes :116-162 and the
   * recursion of do_tiling_for_node :499-561 (sequential here; the order of sibling tasks does not matter,
   * every node name is touched by exactly one task per batch) */
  void tile_internal_node(std::vector<IP>& all, const NodeStructure& node, const NodeStructure& root,
                          size_t previously_taken) {
    const int behaviour = previously_taken > 0 ? ORC_ALWAYS_ADHERE : ORC_TAKE_ALL_WHEN_BELOW_MAX; /* :272-275 */
    const int32_t rel_level = node.level - (root.level + 1);                                       /* :277 */
    IP* b = all.data();
    IP* e = b + all.size();
    const SampleCtx c = ctx();
    const int64_t taken = sample_points(c, b, e, node.morton_index, rel_level, root.bounds, root.max_spacing, behaviour);
    if (taken < 0) {
      error = (int32_t)taken;
      return;
    }
    persist(b, b + taken, node);
    const int32_t child_level = node.level + 1;
    const IP* ranges[9];
    partition_points_into_child_octants(b + taken, e, (uint32_t)child_level, MAX_OCTREE_LEVELS, ranges);
    for (uint8_t octant = 0; octant < 8; ++octant) {
      if (ranges[octant + 1] == ranges[octant]) continue;
      NodeStructure child = node;
      child.morton_index = set_octant_at_level(child.morton_index, (uint32_t)child_level, octant, MAX_OCTREE_LEVELS);
      child.bounds = get_octant_bounds(octant, node.bounds);
      child.level = child_level;
      child.max_spacing /= 2;
      do_tiling_for_node(std::vector<IP>(ranges[octant], ranges[octant + 1]), child, root);
      if (error) return;
    }
  }

  /* tile_node -- TilingAlgorithms.cpp:351-492 */
  void do_tiling_for_node(std::vector<IP>&& node_data, const NodeStructure& node, const NodeStructure& root) {
    if (error) return;
    std::vector<IP> cached = read_pnts_from_disk(node);
    const size_t cached_count = cached.size();
    stats.points_visited += node_data.size() + cached_count;
    const int32_t req = required_morton_index_depth(params.sampler, node.level, root);
    const bool requires_deeper = req > node.level;
    const int32_t max_level = (int32_t)std::min(MAX_OCTREE_LEVELS - 1, node.max_depth);
    if (!requires_deeper) {
      if (req >= max_level) { /* :421-427 */
        const std::vector<IP> all = merge_unsorted(std::move(node_data), std::move(cached));
        persist(all.data(), all.data() + all.size(), node); /* tile_terminal_node :206-241 */
        return;
      }
      std::vector<IP> all = merge_sorted(std::move(node_data), std::move(cached));
      tile_internal_node(all, node, root, cached_count);
      return;
    }
    if (node.level >= max_level) { /* :436-442 */
      const std::vector<IP> all = merge_unsorted(std::move(node_data), std::move(cached));
      persist(all.data(), all.data() + all.size(), node);
      return;
    }
    if (req >= (int32_t)MAX_OCTREE_LEVELS) {
      /* :444-483: the node becomes the root of a new 21-level index.  Literal restatement, including that
       * (a) calculate_morton_index does not clamp, so a point outside new_root.bounds feeds a negative double
       * to static_cast<uint64_t> (the value x86-64 gcc produces is kept: this file is built by it), and
       * (b) the children are still split at the ABSOLUTE level node.level + 1 of the re-rooted keys (:124-125
       * via :479-482).  The reference's own test of this path (test/TestTiler.cpp:164-190) is disabled. */
      std::vector<IP> all = merge_unsorted(std::move(node_data), std::move(cached));
      NodeStructure new_root = node;
      new_root.max_depth = node.max_depth - (uint32_t)node.level;
      for (IP& p : all) p.key = calculate_morton_index(pos(p.idx), new_root.bounds, MAX_OCTREE_LEVELS);
      std::stable_sort(all.begin(), all.end(), [](const IP& l, const IP& r) { return l.key < r.key; }); /* :476 */
      tile_internal_node(all, node, new_root, cached_count);
      return;
    }
    std::vector<IP> all = merge_sorted(std::move(node_data), std::move(cached));
    tile_internal_node(all, node, root, cached_count);
  }

  NodeStructure root_node() const { /* V1 :606-612, V3 :1327-1333 */
    NodeStructure root;
    root.bounds = bounds;
    root.level = -1;
    root.max_depth = params.max_depth;
    root.max_spacing = params.spacing_at_root;
    root.morton_index = 0;
    return root;
  }

  int32_t add_batch(double* batch_xyz, uint64_t n) {
    if (finalized) return ORC_ERR_BAD_ARG;
    if (xyz.size() / 3 + n > 0xFFFFFFFFull) return ORC_ERR_BAD_ARG;
    const double bmin[3] = {bounds.min.x, bounds.min.y, bounds.min.z}, bmax[3] = {bounds.max.x, bounds.max.y, bounds.max.z};
    /* parallel::scatter throws when the batch has fewer points than indexing threads (Parallel.h:181-186) */
    if (params.strategy == ORC_FAST && n < params.fast_concurrency) return ORC_ERR_BAD_ARG;
    const uint32_t base = (uint32_t)(xyz.size() / 3);
    std::vector<uint64_t> keys(n);
    orc_index_points(batch_xyz, n, bmin, bmax, MAX_OCTREE_LEVELS, keys.data()); /* clamps in place */
    xyz.insert(xyz.end(), batch_xyz, batch_xyz + 3 * n);
    /* V1 :600-604: one sort; V3 first batch :1292: one sort; V3 later batches :1380-1395 + :1599-1606: per-thread
     * chunks sorted, split at the start level and k-way merged per start node, earlier chunk first on ties
     * (merge_ranges, Algorithm.h:111-150) -- with the canonical tie order (key, original index) and chunks
     * that are consecutive index ranges, all three are the same sequence: the stable sort by key. */
    std::vector<uint32_t> perm(n);
    orc_sort_by_key(keys.data(), n, perm.data());
    std::vector<IP> sorted(n);
    for (uint64_t i = 0; i < n; ++i) sorted[i] = {base + perm[i], keys[perm[i]]};
    ++batches;
    if (n == 0) return ORC_OK;
    const NodeStructure root = root_node();
    if (params.strategy == ORC_ACCURATE) {
      do_tiling_for_node(std::move(sorted), root, root);
      return error;
    }
    if (level_of_start_nodes < 0) /* first iteration :1294-1295 */
      level_of_start_nodes = (int32_t)estimate_start_node_level(sorted, params.fast_concurrency);
    stats.fast_start_levels = level_of_start_nodes;
    const uint32_t S = (uint32_t)level_of_start_nodes;
    const uint32_t shift = (MAX_OCTREE_LEVELS - S) * 3;
    uint64_t i = 0;
    while (i < n && !error) { /* split_indexed_points_into_subranges :1537-1578 / prepare_range_for_tiling :1620-1659 */
      const uint64_t prefix = sorted[i].key >> shift;
      uint64_t j = i;
      while (j < n && (sorted[j].key >> shift) == prefix) ++j;
      NodeStructure node;
      node.level = (int32_t)S - 1;
      node.max_depth = root.max_depth;
      node.max_spacing = (float)(root.max_spacing / std::pow(2, (double)S));
      node.morton_index = prefix << shift;
      node.bounds = get_bounds_from_morton_index(node.morton_index, MAX_OCTREE_LEVELS, bounds, S);
      do_tiling_for_node(std::vector<IP>(sorted.begin() + (ptrdiff_t)i, sorted.begin() + (ptrdiff_t)j), node, root);
      i = j;
    }
    return error;
  }

  /* TilingAlgorithmV3::finalize -> reconstruct_left_out_nodes :1717-1784, reconstruct_single_node :1661-1715 */
  int32_t finalize() {
    if (finalized) return error;
    finalized = true;
    if (params.strategy != ORC_FAST || level_of_start_nodes <= 0) return error;
    const uint32_t S = (uint32_t)level_of_start_nodes;
    /* ancestors of the existing nodes with S octants, deepest first */
    std::vector<std::map<uint64_t, int>> todo(S); /* todo[lv]: node indices (lv octants) to reconstruct */
    for (const auto& kv : files) {
      if (kv.first.first != (int32_t)S - 1 || kv.second.empty()) continue;
      uint64_t index = kv.first.second >> ((MAX_OCTREE_LEVELS - S) * 3);
      for (uint32_t lv = S; lv-- > 0;) {
        index >>= 3;
        todo[lv][index] = 1;
      }
    }
    const SampleCtx c = ctx();
    for (uint32_t lv = S; lv-- > 0 && !error;) {
      for (const auto& nk : todo[lv]) {
        const uint64_t index = nk.first;
        const uint64_t node_key = (lv == 0) ? 0 : (index << ((MAX_OCTREE_LEVELS - lv) * 3));
        std::vector<IP> data;
        for (uint8_t octant = 0; octant < 8; ++octant) { /* :1665-1679 */
          const uint64_t child_key = node_key | ((uint64_t)octant << ((MAX_OCTREE_LEVELS - lv - 1) * 3));
          auto it = files.find({(int32_t)lv, child_key});
          if (it == files.end()) continue;
          /* index_points<21>(..., root_bounds, ClampToBounds) :1682-1688 on a copy of the positions */
          for (uint32_t id : it->second) {
            V3 p = pos(id);
            if (!bounds.isInside(p)) {
              p.x = std::min(bounds.max.x, std::max(bounds.min.x, p.x));
              p.y = std::min(bounds.max.y, std::max(bounds.min.y, p.y));
              p.z = std::min(bounds.max.z, std::max(bounds.min.z, p.z));
            }
            data.push_back({id, calculate_morton_index(p, bounds, MAX_OCTREE_LEVELS)});
          }
        }
        if (data.empty()) continue;
        for (size_t q = 1; q < data.size(); ++q)
          if (data[q].key < data[q - 1].key) {
            ++unsorted_cached_nodes;
            break;
          }
        const int64_t taken = sample_points(c, data.data(), data.data() + data.size(), node_key, (int32_t)lv - 1, bounds,
                                            params.spacing_at_root, ORC_ALWAYS_ADHERE);
        if (taken < 0) {
          error = (int32_t)taken;
          break;
        }
        NodeStructure node;
        node.level = (int32_t)lv - 1;
        node.morton_index = node_key;
        persist(data.data(), data.data() + taken, node);
      }
    }
    return error;
  }
};

}  // namespace

struct orc_tiler {
  MBTiler t;
};

orc_tiler* orc_tiler_create(const double bmin[3], const double bmax[3], const orc_tile_params* params) {
  if (!params) return nullptr;
  orc_tiler* h = new orc_tiler();
  h->t.bounds = make_aabb(bmin, bmax);
  h->t.params = *params;
  h->t.stats.max_level = -1;
  h->t.stats.fast_start_levels = -1;
  return h;
}
void orc_tiler_destroy(orc_tiler* h) { delete h; }
int32_t orc_tiler_add_batch(orc_tiler* h, double* xyz, uint64_t n) { return h ? h->t.add_batch(xyz, n) : ORC_ERR_BAD_ARG; }
int32_t orc_tiler_finalize(orc_tiler* h) { return h ? h->t.finalize() : ORC_ERR_BAD_ARG; }
void orc_tiler_counts(const orc_tiler* h, uint64_t* num_nodes, uint64_t* num_stored, uint64_t* num_points,
                      uint64_t* unsorted_cached_nodes) {
  uint64_t nn = 0, ns = 0;
  for (const auto& kv : h->t.files) {
    if (kv.second.empty()) continue;
    ++nn;
    ns += kv.second.size();
  }
  if (num_nodes) *num_nodes = nn;
  if (num_stored) *num_stored = ns;
  if (num_points) *num_points = h->t.xyz.size() / 3;
  if (unsorted_cached_nodes) *unsorted_cached_nodes = h->t.unsorted_cached_nodes;
}
void orc_tiler_stats(const orc_tiler* h, orc_tile_stats* out) { *out = h->t.stats; }
void orc_tiler_export(const orc_tiler* h, int8_t* node_level, uint64_t* node_key, uint64_t* node_offset,
                      uint64_t* node_count, uint32_t* ids, double* xyz_out) {
  uint64_t k = 0, off = 0;
  for (const auto& kv : h->t.files) { /* std::map order = (level, key) ascending */
    if (kv.second.empty()) continue;
    node_level[k] = (int8_t)kv.first.first;
    node_key[k] = kv.first.second;
    node_offset[k] = off;
    node_count[k] = kv.second.size();
    for (uint32_t id : kv.second) ids[off++] = id;
    ++k;
  }
  if (xyz_out) std::memcpy(xyz_out, h->t.xyz.data(), h->t.xyz.size() * sizeof(double));
}

/* ---- BinaryPersistence, uncompressed --------------------------------------------------------------- */
namespace {
const size_t kAttrBytes[12] = {3, 12, 2, 1, 1, 8, 1, 1, 2, 1, 1, 1};
/* persist_points writes, after the positions: colors, normals, intensities, classifications, edge of
 * flight lines, gps times, number of returns, return numbers, point source ids, scan angle ranks, scan
 * direction flags, user data (BinaryPersistence.h:120-190); retrieve_points reads in the same order
 * (BinaryPersistence.cpp:270-360). */
const int kFileOrder[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 9, 11};
}  // namespace

int32_t orc_bin_persist_points(const char* path, const uint32_t* point_refs, uint64_t count, const double* xyz,
                               const void* const columns[12]) {
  if (count == 0) return ORC_OK; /* BinaryPersistence.h:51-53 */
  FILE* f = std::fopen(path, "wb");
  if (!f) return ORC_ERR_BAD_ARG;
  uint32_t bitmask = 0;
  for (int a = 0; a < 12; ++a)
    if (columns && columns[a]) bitmask |= 1u << a; /* has_attribute && pointer != nullptr, :73-112 */
  std::fwrite(&bitmask, sizeof(bitmask), 1, f); /* write_binary, util/io/io_util.h:15-18 */
  std::fwrite(&count, sizeof(count), 1, f);
  for (uint64_t i = 0; i < count; ++i) std::fwrite(xyz + 3 * (size_t)point_refs[i], 24, 1, f); /* :117-118 */
  for (int k = 0; k < 12; ++k) {
    const int a = kFileOrder[k];
    if (!(bitmask & (1u << a))) continue;
    const unsigned char* col = static_cast<const unsigned char*>(columns[a]);
    for (uint64_t i = 0; i < count; ++i) std::fwrite(col + kAttrBytes[a] * (size_t)point_refs[i], kAttrBytes[a], 1, f);
  }
  return std::fclose(f) == 0 ? ORC_OK : ORC_ERR_BAD_ARG;
}

int32_t orc_bin_retrieve_points(const char* path, uint32_t* bitmask_out, uint64_t* count_out, double* xyz_out,
                                void* const columns_out[12]) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return ORC_ERR_BAD_ARG;
  uint32_t bitmask = 0;
  uint64_t count = 0;
  bool ok = std::fread(&bitmask, 4, 1, f) == 1 && std::fread(&count, 8, 1, f) == 1; /* BinaryPersistence.cpp:230-234 */
  if (ok && xyz_out) {
    ok = count == 0 || std::fread(xyz_out, 24, count, f) == count;
    for (int k = 0; ok && k < 12; ++k) {
      const int a = kFileOrder[k];
      if (!(bitmask & (1u << a))) continue;
      if (columns_out && columns_out[a])
        ok = count == 0 || std::fread(columns_out[a], kAttrBytes[a], count, f) == count;
      else
        ok = std::fseek(f, (long)(kAttrBytes[a] * count), SEEK_CUR) == 0;
    }
  }
  std::fclose(f);
  if (bitmask_out) *bitmask_out = bitmask;
  if (count_out) *count_out = count;
  return ok ? ORC_OK : ORC_ERR_BAD_ARG;
}

/* ---- LAS records ------------------------------------------------------------------------------------ */
namespace {
/* what LASzip's reader leaves in a laszip_point for point data record formats 0-5 (LAS 1.2 / 1.3 specification,
 * "Point Data Record Format 0..3"; laszip_api.h: return_number:3, number_of_returns:3, scan_direction_flag:1,
 * edge_of_flight_line:1, classification:5 + 3 flag bits) */
struct LasPoint {
  int32_t X, Y, Z;
  uint16_t intensity;
  uint8_t return_number, number_of_returns, scan_direction_flag, edge_of_flight_line, classification;
  int8_t scan_angle_rank;
  uint8_t user_data;
  uint16_t point_source_ID;
  double gps_time;
  uint16_t rgb[3];
};
LasPoint las_read_record(const uint8_t* r, uint32_t format) {
  LasPoint p{};
  std::memcpy(&p.X, r, 4);
  std::memcpy(&p.Y, r + 4, 4);
  std::memcpy(&p.Z, r + 8, 4);
  std::memcpy(&p.intensity, r + 12, 2);
  if (format >= 6) {
    /* LAS 1.4 point data record formats 6-10 ("Point Data Record Format 6": return number:4, number of returns:4 |
     * classification flags:4, scanner channel:2, scan direction:1, edge of flight line:1 | classification u8 | user data |
     * scan angle i16 in 0.006 degree | point source id | gps time), as LASzip's raw reader maps them onto the fields of
     * a laszip_point the reference reads (LASzip src/lasreaditemraw.hpp, LASreadItemRaw_POINT14_LE::read -- LASzip is not
     * part of the reference tree and its version is not pinned there: restated from its published source, PARITY
     * UNPINNED): returns above 7 saturate, classifications above 31 do not fit the 5-bit field and read 0, the scan
     * angle is rounded to whole degrees in float and clamped to a signed byte. */
    const uint32_t rn = r[14] & 15u, nor = (r[14] >> 4) & 15u;
    if (nor > 7) {
      p.return_number = rn > 6 ? (rn >= nor ? 7 : 6) : (uint8_t)rn;
      p.number_of_returns = 7;
    } else {
      p.return_number = (uint8_t)(rn & 7u);  /* (the field has 3 bits) */
      p.number_of_returns = (uint8_t)nor;
    }
    p.scan_direction_flag = (r[15] >> 6) & 1;
    p.edge_of_flight_line = (r[15] >> 7) & 1;
    p.classification = r[16] < 32 ? r[16] : 0;
    p.user_data = r[17];
    int16_t angle;
    std::memcpy(&angle, r + 18, 2);
    const float deg = 0.006f * angle;
    const int16_t q = deg >= 0 ? (int16_t)(deg + 0.5f) : (int16_t)(deg - 0.5f);      /* I16_QUANTIZE */
    p.scan_angle_rank = (int8_t)(q <= -128 ? -128 : (q >= 127 ? 127 : q));            /* I8_CLAMP */
    std::memcpy(&p.point_source_ID, r + 20, 2);
    std::memcpy(&p.gps_time, r + 22, 8);
    if (format == 7 || format == 8 || format == 10) std::memcpy(p.rgb, r + 30, 6);
    return p;
  }
  p.return_number = r[14] & 7;
  p.number_of_returns = (r[14] >> 3) & 7;
  p.scan_direction_flag = (r[14] >> 6) & 1;
  p.edge_of_flight_line = (r[14] >> 7) & 1;
  p.classification = r[15] & 31;
  p.scan_angle_rank = (int8_t)r[16];
  p.user_data = r[17];
  std::memcpy(&p.point_source_ID, r + 18, 2);
  size_t at = 20;
  if (format == 1 || format == 3 || format == 4 || format == 5) {  /* 4, 5 (LAS 1.3): formats 1, 3 + 29 bytes of wave packet */
    std::memcpy(&p.gps_time, r + at, 8);
    at += 8;
  }
  if (format == 2 || format == 3 || format == 5) std::memcpy(p.rgb, r + at, 6);
  return p;
}
}  // namespace

int32_t orc_las_decode(const uint8_t* records, uint64_t n, const orc_las_layout* L, double* xyz_out,
                       void* const col[12]) {
  static const uint32_t min_bytes[11] = {20, 28, 26, 34, 57, 63, 30, 36, 38, 59, 67};
  if (!L || L->point_format > 10 || L->record_bytes < min_bytes[L->point_format]) return ORC_ERR_BAD_ARG;
  for (uint64_t i = 0; i < n; ++i) {
    const LasPoint p = las_read_record(records + i * L->record_bytes, L->point_format);
    if (xyz_out) {
      /* position_from_las_point, LASFile.cpp:82-92 */
      double x = L->offset[0] + p.X * L->scale[0];
      double y = L->offset[1] + p.Y * L->scale[1];
      double z = L->offset[2] + p.Z * L->scale[2];
      x = std::min(L->max[0], std::max(L->min[0], x));
      y = std::min(L->max[1], std::max(L->min[1], y));
      z = std::min(L->max[2], std::max(L->min[2], z));
      xyz_out[3 * i] = x;
      xyz_out[3 * i + 1] = y;
      xyz_out[3 * i + 2] = z;
    }
    if (!col) continue;
    /* las_read_points_into, LASFile.cpp:590-628 */
    if (col[0]) {
      uint8_t* c = static_cast<uint8_t*>(col[0]) + 3 * i;
      c[0] = static_cast<uint8_t>(p.rgb[0] >> 8);
      c[1] = static_cast<uint8_t>(p.rgb[1] >> 8);
      c[2] = static_cast<uint8_t>(p.rgb[2] >> 8);
    }
    if (col[2]) static_cast<uint16_t*>(col[2])[i] = p.intensity;
    if (col[3]) static_cast<uint8_t*>(col[3])[i] = p.classification;
    if (col[5]) static_cast<double*>(col[5])[i] = p.gps_time;
    if (col[4]) static_cast<uint8_t*>(col[4])[i] = p.edge_of_flight_line;
    if (col[6]) static_cast<uint8_t*>(col[6])[i] = p.number_of_returns;
    if (col[7]) static_cast<uint8_t*>(col[7])[i] = p.return_number;
    if (col[8]) static_cast<uint16_t*>(col[8])[i] = p.point_source_ID;
    if (col[10]) static_cast<int8_t*>(col[10])[i] = p.scan_angle_rank;
    if (col[9]) static_cast<uint8_t*>(col[9])[i] = p.scan_direction_flag;
    if (col[11]) static_cast<uint8_t*>(col[11])[i] = p.user_data;
  }
  return ORC_OK;
}

void orc_generate_uniform(uint64_t seed, uint64_t first_point, uint64_t n, double* xyz) {
  /* splitmix64 stream; draw k of the stream is mix(seed + (k+1)*GOLDEN).  Point i uses draws
   * 3i, 3i+1, 3i+2 so any slice can be generated independently. */
  const uint64_t G = 0x9E3779B97F4A7C15ull;
  for (uint64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      uint64_t z = seed + (3 * (first_point + i) + (uint64_t)a + 1) * G;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z = z ^ (z >> 31);
      xyz[3 * i + a] = (double)(z >> 11) * 0x1.0p-53;
    }
}

} /* extern "C" */
