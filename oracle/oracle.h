/*
 * oracle.h -- C interface of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain CPU restatement of the
 * reference tiler hot path (igd-geo/schwarzwald) used as the checker in tests/,
 * in __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  Nothing in
 * schwarzwald_amd/ (the product) may include, link or call it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - Morton encode, MortonIndex ops, octant bounds, child-octant partition,
 *     stable_partition_with_jumps, merge_ranges, RANDOM_GRID: pinned by the
 *     reference's own unit-test vectors (tests/test_oracle_reference_kats.py)
 *     and, for MortonIndex.h / Algorithm.h, by oracle/_ref built from the
 *     reference's own headers.
 *   - GRID_CENTER, MIN_DISTANCE (SparseGrid), JITTERED, tile_node control flow,
 *     FAST start level + reconstruction: PARITY UNPINNED -- the reference holds
 *     no test vector for them and the reference sources that implement them
 *     need Boost/GSL/taskflow, which this image lacks, so they cannot be
 *     compiled here.  They are line-by-line restatements with file:line cites.
 */
#ifndef SWZ_ORACLE_H
#define SWZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_RANDOM_GRID = 0, ORC_GRID_CENTER = 1, ORC_MIN_DISTANCE = 2, ORC_JITTERED = 3 };
enum { ORC_TAKE_ALL_WHEN_BELOW_MAX = 0, ORC_ALWAYS_ADHERE = 1 };
enum { ORC_ACCURATE = 0, ORC_FAST = 1 };

/* error codes (negative return values) */
enum {
  ORC_OK = 0,
  ORC_ERR_JITTER_GRID_TOO_SMALL = -2, /* Sampling.h:632-635 throws */
  ORC_ERR_JITTER_NODE_TOO_DEEP = -3,  /* Sampling.h:642-653 throws */
  ORC_ERR_REROOT_UNSUPPORTED = -4,    /* TilingAlgorithms.cpp:444-483, out of scope */
  ORC_ERR_BAD_ARG = -5
};

/* calculate_morton_index<levels> -- OctreeAlgorithms.h:64-87 (no clamp). */
uint64_t orc_calculate_morton_index(const double p[3], const double bmin[3], const double bmax[3],
                                    uint32_t levels);
/* calculate_morton_index_naive<levels> -- OctreeAlgorithms.h:89-102. */
uint64_t orc_calculate_morton_index_naive(const double p[3], const double bmin[3],
                                          const double bmax[3], uint32_t levels);
/* index_points<levels> with ClampToBounds -- OctreeAlgorithms.h:145-197.  Clamps xyz IN PLACE. */
void orc_index_points(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                      uint32_t levels, uint64_t* keys_out);
/* Canonical order of Range::sort (util/containers/Range.h:62-66): ascending key, ties by
 * original index (std::sort leaves ties unspecified; this is the stable choice). */
void orc_sort_by_key(const uint64_t* keys, uint64_t n, uint32_t* perm_out);

/* get_octant_bounds -- OctreeAlgorithms.cpp:3-18 */
void orc_get_octant_bounds(uint8_t octant, const double bmin[3], const double bmax[3],
                           double omin[3], double omax[3]);
/* get_bounds_from_morton_index<21> -- OctreeAlgorithms.h:104-116 */
void orc_get_bounds_from_morton_index(uint64_t key, uint32_t levels, const double bmin[3],
                                      const double bmax[3], uint32_t depth, double omin[3],
                                      double omax[3]);
/* partition_points_into_child_octants -- OctreeAlgorithms.h:240-265; offsets[9] into sorted keys */
void orc_partition_points_into_child_octants(const uint64_t* sorted_keys, uint64_t n,
                                             uint32_t level, uint32_t levels, uint64_t offsets[9]);
/* MortonIndex<levels> helpers -- MortonIndex.h:123-145 */
uint64_t orc_truncate_to_level(uint64_t key, uint32_t level, uint32_t levels);
uint8_t orc_get_octant_at_level(uint64_t key, uint32_t level, uint32_t levels);
uint64_t orc_set_octant_at_level(uint64_t key, uint32_t level, uint8_t octant, uint32_t levels);
/* OctreeNodeIndex64::to_grid_index -- OctreeNodeIndex.h:357-363 */
void orc_to_grid_index(uint64_t index, uint32_t levels, uint64_t out_xyz[3]);
uint32_t orc_get_prev_power_of_two(uint32_t x); /* stuff.cpp:340-349 */

/* required_morton_index_depth -- Sampling.cpp:29-62 */
int32_t orc_required_morton_index_depth(int sampler, int32_t node_level, const double root_min[3],
                                        const double root_max[3], float root_max_spacing);

/* sample_points -- Sampling.h:799-821 dispatching to :187-308, :314-416, :421-471, :598-759.
 * keys/idx describe a Morton-sorted range of n points (idx = index into xyz).  The range is
 * stably partitioned in place into [taken | rest]; returns the number taken or a negative
 * error code. */
int64_t orc_sample_points(int sampler, uint64_t max_points_per_node, uint64_t* keys, uint32_t* idx,
                          uint64_t n, const double* xyz, uint64_t node_key, int32_t node_level,
                          uint32_t levels, const double root_min[3], const double root_max[3],
                          float spacing_at_root, int behaviour);

/* Reference-literal SparseGrid greedy (SparseGrid.cpp:116-146) over points in the given order;
 * accepted[i] in {0,1}.  Exposed so tests can compare it against a brute-force greedy. */
void orc_sparse_grid_greedy(const double* xyz, const uint32_t* idx, uint64_t n, const double nmin[3],
                            const double nmax[3], float spacing, uint8_t* accepted);

typedef struct {
  int32_t sampler;
  uint64_t max_points_per_node;
  float spacing_at_root;
  uint32_t max_depth;
  int32_t strategy;          /* ORC_ACCURATE | ORC_FAST */
  uint32_t fast_concurrency; /* FAST only: num_indexing_threads of the first batch */
} orc_tile_params;

typedef struct {
  uint64_t num_nodes;      /* nodes persisted (incl. reconstructed ancestors) */
  uint64_t points_visited; /* sum over tile_node calls of the points handed to the node */
  int32_t max_level;       /* deepest level that took points */
  int32_t fast_start_levels; /* FAST: _level_of_start_nodes, else -1 */
} orc_tile_stats;

/* One batch through TilingAlgorithmV1 (TilingAlgorithms.cpp:577-626) or the first iteration of
 * TilingAlgorithmV3 + finalize (:1250-1360, :1661-1784) with a lossless in-memory persistence.
 * Outputs are in SORTED order: keys_out[i] ascending, perm_out[i] = original index,
 * level_out[i] = level of the node that persisted point i (-1 = root), dup_mask_out[i] (FAST,
 * may be NULL) bit (l+1) set when the point is also stored in the reconstructed node at level l.
 * xyz is clamped in place like index_point does. */
int32_t orc_tile(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                 const orc_tile_params* params, uint64_t* keys_out, uint32_t* perm_out,
                 int8_t* level_out, uint32_t* dup_mask_out, orc_tile_stats* stats);

/* Same with `threads` worker threads used the way the reference uses its indexing executor: the encode runs in
 * chunks on all threads (util/Parallel.h:172-213), the whole-batch sort and (ACCURATE) the root node on one
 * thread (TilingAlgorithms.cpp:600-626), child nodes with >= 100 000 points as tasks, smaller ones inline
 * (:25, 499-561); FAST start nodes are independent tasks (:1345-1353).  Results do not depend on threads. */
int32_t orc_tile_mt(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                    const orc_tile_params* params, uint32_t threads, uint64_t* keys_out, uint32_t* perm_out,
                    int8_t* level_out, uint32_t* dup_mask_out, orc_tile_stats* stats_out);

/* ---- multi-batch tiler: one TilingAlgorithm object fed batch after batch the way
 * Tiler::index does when the input exceeds internal_cache_size (core/process/Tiler.cpp:509-510), over a
 * BinaryPersistence-like store (a node's file is replaced by every persist_points).  Restates
 * read_pnts_from_disk (TilingAlgorithms.cpp:50-109), merge_node_data_sorted/_unsorted (Node.cpp:4-35), the
 * behaviour switch on the cached count (:272-275), re-rooting (:444-483), V3's later iterations
 * (:1362-1453, 1620-1659) and finalize (:1661-1784).  Point ids count the points of all batches in input
 * order.  PARITY UNPINNED (the reference's multi-batch tests, test/TestTiler.cpp:85-190, are commented out). */
typedef struct orc_tiler orc_tiler;
orc_tiler* orc_tiler_create(const double bmin[3], const double bmax[3], const orc_tile_params* params);
void orc_tiler_destroy(orc_tiler* t);
/* one build_execution_graph + run; xyz (n x 3) is clamped in place like index_point does */
int32_t orc_tiler_add_batch(orc_tiler* t, double* xyz, uint64_t n);
/* TilingAlgorithmBase::finalize (FAST: reconstruct_left_out_nodes); no-op for ACCURATE */
int32_t orc_tiler_finalize(orc_tiler* t);
/* node files present / points stored in them (FAST stores copies in reconstructed nodes) / points added /
 * nodes whose re-keyed cached points were not in ascending order (the reference does not check, :103-106) */
void orc_tiler_counts(const orc_tiler* t, uint64_t* num_nodes, uint64_t* num_stored, uint64_t* num_points,
                      uint64_t* unsorted_cached_nodes);
void orc_tiler_stats(const orc_tiler* t, orc_tile_stats* out);
/* nodes ordered by (level, Morton index); ids[node_offset[k] .. +node_count[k]) = the node's points in file
 * order; xyz_out (may be NULL) receives the clamped positions of all points by id */
void orc_tiler_export(const orc_tiler* t, int8_t* node_level, uint64_t* node_key, uint64_t* node_offset,
                      uint64_t* node_count, uint32_t* ids, double* xyz_out);

/* orc_tile_mt that also reports wall seconds per stage: index (all threads), sort (one thread), tiling (node tasks) */
int32_t orc_tile_mt_timed(double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                          const orc_tile_params* params, uint32_t threads, uint64_t* keys_out, uint32_t* perm_out,
                          int8_t* level_out, uint32_t* dup_mask_out, orc_tile_stats* stats_out, double stage_seconds[3]);

/* util/algorithms/Algorithm.h restatements on int ranges, for the reference's TestAlgorithm vectors */
int64_t orc_stable_partition_take_multiples(int32_t* values, int64_t n, int32_t modulus);
void orc_merge_ranges_i32(const int32_t* const* ranges, const int64_t* sizes, int64_t num_ranges,
                          int32_t* out);

/* BinaryPersistence with Compressed::No (core/io/BinaryPersistence.h:45-193 persist_points,
 * core/io/BinaryPersistence.cpp:212-375 retrieve_points).  The points of the node are given the way the
 * tiler gives them: references (indices) into the SoA columns of the batch's PointBuffer
 * (PointBuffer.h:292-304); columns[a] == NULL means the attribute is absent.  Column index = bit number
 * of the properties bitmask (BinaryPersistence.h:24-35): 0 RGB (3 x u8), 1 normal (3 x f32), 2 intensity
 * (u16), 3 classification, 4 edge of flight line, 5 GPS time (f64), 6 number of returns, 7 return number,
 * 8 point source id (u16), 9 scan direction flag, 10 scan angle rank (i8), 11 user data.
 * PARITY UNPINNED beyond the reference's own round-trip test (test/TestBinaryPersistence.cpp:52-108), which
 * tests/test_bin_persistence.py reproduces: the reference ships no golden node file. */
int32_t orc_bin_persist_points(const char* path, const uint32_t* point_refs, uint64_t count, const double* xyz,
                               const void* const columns[12]);
/* Reads the header (and, when xyz_out != NULL, the arrays; buffers must hold *count_out points, so call
 * twice).  columns_out[a] may be NULL to skip an attribute. */
int32_t orc_bin_retrieve_points(const char* path, uint32_t* bitmask_out, uint64_t* count_out, double* xyz_out,
                                void* const columns_out[12]);

/* LAS point records -> positions + attributes: position_from_las_point (core/io/LASFile.cpp:79-94) and
 * las_read_points_into (:578-632) applied to the laszip_point that LASzip's reader fills from an UNCOMPRESSED
 * point data record (LAS 1.2 formats 0-3; LASzip is not in the reference tree, see include/swz_gpu.h).
 * columns_out uses the indices of orc_bin_persist_points; NULL entries are skipped.  PARITY UNPINNED: the
 * reference's LAS tests (test/TestLASFile.cpp) need LASzip and its test files. */
typedef struct {
  double scale[3], offset[3], min[3], max[3];
  uint32_t point_format, record_bytes;
} orc_las_layout;
int32_t orc_las_decode(const uint8_t* records, uint64_t n, const orc_las_layout* layout, double* xyz_out,
                       void* const columns_out[12]);

/* splitmix64 synthetic workload of SURVEY.md section 8(d): point i draws x,y,z consecutively. */
void orc_generate_uniform(uint64_t seed, uint64_t first_point, uint64_t n, double* xyz);

#ifdef __cplusplus
}
#endif
#endif
