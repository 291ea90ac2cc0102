/*
 * ref_driver.cpp -- thin C wrappers around the parts of the reference that compile in this image
 * WITHOUT any stand-in header: core/datastructures/MortonIndex.h, util/algorithms/Algorithm.h and
 * util/containers/Range.h depend on the C++ standard library only.  Everything else on the hot
 * path (OctreeAlgorithms.h, Sampling.h, SparseGrid, Vector3.h, ...) includes Boost / GSL headers
 * that are absent here, so it is NOT built (DESIGN.md "Oracle").
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile from the sources where they lie under
 * /root/reference into oracle/_ref/libswzref.so (git-ignored); used by tests to validate the
 * restatement in oracle.cpp.  No reference source is copied into this repository.
 */
#include <cassert>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "datastructures/MortonIndex.h" /* reference: schwarzwald/core/datastructures/MortonIndex.h */
#include "algorithms/Algorithm.h"       /* reference: schwarzwald/util/algorithms/Algorithm.h */

namespace {
template <unsigned L>
uint64_t truncate_l(uint64_t key, uint32_t level) {
  return (uint64_t)MortonIndex<L>{(typename MortonIndex<L>::Store_t)key}.truncate_to_level(level).get();
}
template <unsigned L>
uint8_t octant_l(uint64_t key, uint32_t level) {
  return MortonIndex<L>{(typename MortonIndex<L>::Store_t)key}.get_octant_at_level(level);
}
template <unsigned L>
uint64_t set_octant_l(uint64_t key, uint32_t level, uint8_t octant) {
  MortonIndex<L> k{(typename MortonIndex<L>::Store_t)key};
  k.set_octant_at_level(level, octant);
  return (uint64_t)k.get();
}
template <unsigned L>
uint64_t from_levels_l(const uint8_t* octants) {
  std::array<uint8_t, L> a;
  for (unsigned i = 0; i < L; ++i) a[i] = octants[i];
  return (uint64_t)MortonIndex<L>{a}.get();
}
template <unsigned L>
uint64_t ctor_l(uint64_t v) {
  return (uint64_t)MortonIndex<L>{(typename MortonIndex<L>::Store_t)v}.get();
}
#define DISPATCH(levels, fn, ...)                 \
  switch (levels) {                               \
    case 1: return fn<1>(__VA_ARGS__);            \
    case 2: return fn<2>(__VA_ARGS__);            \
    case 4: return fn<4>(__VA_ARGS__);            \
    case 5: return fn<5>(__VA_ARGS__);            \
    case 10: return fn<10>(__VA_ARGS__);          \
    case 20: return fn<20>(__VA_ARGS__);          \
    case 21: return fn<21>(__VA_ARGS__);          \
    default: return 0;                            \
  }

struct KeyIdx {
  uint32_t idx;
  MortonIndex64 morton_index;
};
} // namespace

extern "C" {
uint64_t ref_truncate_to_level(uint64_t key, uint32_t level, uint32_t levels) { DISPATCH(levels, truncate_l, key, level) }
uint8_t ref_get_octant_at_level(uint64_t key, uint32_t level, uint32_t levels) { DISPATCH(levels, octant_l, key, level) }
uint64_t ref_set_octant_at_level(uint64_t key, uint32_t level, uint8_t octant, uint32_t levels) { DISPATCH(levels, set_octant_l, key, level, octant) }
uint64_t ref_morton_from_levels(const uint8_t* octants, uint32_t levels) { DISPATCH(levels, from_levels_l, octants) }
uint64_t ref_morton_ctor(uint64_t value, uint32_t levels) { DISPATCH(levels, ctor_l, value) }

/* to_string(MortonIndex64, levels) / from_string<21> -- MortonIndex.h:178-207 */
void ref_morton64_to_string(uint64_t key, uint32_t levels, char* out, uint32_t out_size) {
  const std::string s = to_string(MortonIndex64{key}, levels);
  std::strncpy(out, s.c_str(), out_size);
  if (out_size) out[out_size - 1] = 0;
}
uint64_t ref_morton64_from_string(const char* s) { return from_string<21>(std::string(s)).get(); }

/* The reference's stable_partition_with_jumps driven by the "first point of every cell" jump
 * predicate of RandomSortedGridSampling (Sampling.h:253-284): partition routine, MortonIndex and
 * std::partition_point are the reference's / libstdc++'s, only the lambda is restated. */
int64_t ref_partition_first_of_cell(uint64_t* keys, uint32_t* idx, int64_t n, uint32_t level) {
  std::vector<KeyIdx> v((size_t)n);
  for (int64_t i = 0; i < n; ++i) v[(size_t)i] = {idx[i], MortonIndex64{keys[i]}};
  auto pivot = stable_partition_with_jumps(v.begin(), v.end(), [level](const auto cur, const auto end) {
    const auto cell = cur->morton_index.truncate_to_level(level);
    const auto next = std::partition_point(cur + 1, end, [cell, level](const auto& o) {
      return o.morton_index.truncate_to_level(level).get() <= cell.get();
    });
    return std::make_pair(cur, next);
  });
  for (int64_t i = 0; i < n; ++i) {
    keys[i] = v[(size_t)i].morton_index.get();
    idx[i] = v[(size_t)i].idx;
  }
  return pivot - v.begin();
}

/* test/TestAlgorithm.cpp:24-80 predicate through the reference's stable_partition_with_jumps */
int64_t ref_stable_partition_take_multiples(int32_t* values, int64_t n, int32_t modulus) {
  std::vector<int32_t> v(values, values + n);
  auto is_match = [modulus](int32_t x) { return (x % modulus) == 0; };
  auto pivot = stable_partition_with_jumps(v.begin(), v.end(), [&](auto cur, auto end) {
    if (!is_match(*cur)) {
      const auto m = std::find_if(cur + 1, end, is_match);
      if (m == end) return std::make_pair(end, end);
      return std::make_pair(m, m + 1);
    }
    return std::make_pair(cur, cur + 1);
  });
  std::copy(v.begin(), v.end(), values);
  return pivot - v.begin();
}

/* merge_ranges -- Algorithm.h:111-150 */
void ref_merge_ranges_i32(const int32_t* const* ranges, const int64_t* sizes, int64_t num_ranges, int32_t* out) {
  using It = std::vector<int32_t>::iterator;
  std::vector<std::vector<int32_t>> store;
  int64_t total = 0;
  for (int64_t r = 0; r < num_ranges; ++r) {
    store.emplace_back(ranges[r], ranges[r] + sizes[r]);
    total += sizes[r];
  }
  std::vector<util::Range<It>> rs;
  for (auto& s : store) rs.push_back(util::Range<It>{s.begin(), s.end()});
  std::vector<int32_t> o((size_t)total);
  merge_ranges(util::Range<std::vector<util::Range<It>>::iterator>{rs.begin(), rs.end()},
               util::Range<It>{o.begin(), o.end()}, std::less<int32_t>{});
  std::copy(o.begin(), o.end(), out);
}

/* split_range_into_chunks -- Algorithm.h:85-100: returns chunk begin offsets (num_chunks+1) */
void ref_split_range_into_chunks(int64_t n, int64_t num_chunks, int64_t* offsets) {
  std::vector<int32_t> v((size_t)n);
  auto chunks = split_range_into_chunks((size_t)num_chunks, v.begin(), v.end());
  for (size_t c = 0; c < chunks.size(); ++c) offsets[c] = chunks[c].first - v.begin();
  offsets[chunks.size()] = chunks.back().second - v.begin();
}
}
