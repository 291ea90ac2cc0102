/*
 * swz_gpu.h -- C ABI of the MI355X-native Schwarzwald tiler hot path (libswz_gpu.so).
 *
 * The reference (igd-geo/schwarzwald, paths below relative to schwarzwald/) has no plugin
 * loader or C ABI for this path; its "plugin surface" is three compile-time C++ seams
 * (SURVEY.md section 8(b)).  This header is the C-ABI boundary a maintainer binds behind
 * those seams (INTEGRATION.md shows the adapter).  Conventions follow the only C ABI the
 * reference itself consumes (LASzip, core/io/LASFile.cpp:14-75): every call returns an int
 * status, 0 = OK, and the message of the last failure is read with swz_last_error().
 *
 * Ownership: the library owns all device workspace inside swz_ctx.  "host" entry points take
 * caller-owned host buffers (what PointBuffer::positions().data() hands over,
 * core/datastructures/PointBuffer.h:291-304); "_device" entry points take device pointers
 * (hipMalloc'd or a torch tensor's data_ptr()) and run on the context's stream.
 * Threading: one call in flight per context; independent contexts (one per GPU) may run
 * concurrently.  Nothing here throws; the adapter turns a non-zero status into
 * std::runtime_error like the rest of the tiler does (executable/main.cpp:599-602).
 * There is NO CPU fallback: without a usable HIP device swz_create() fails.
 */
#ifndef SWZ_GPU_H
#define SWZ_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWZ_ABI_VERSION 3

/* status codes */
enum {
  SWZ_OK = 0,
  SWZ_ERR_HIP = 1,                   /* a HIP runtime call failed (message has the HIP error) */
  SWZ_ERR_BAD_ARG = 2,
  SWZ_ERR_JITTER_GRID_TOO_SMALL = 3, /* JitteredSampling throws: Sampling.h:632-635 */
  SWZ_ERR_JITTER_NODE_TOO_DEEP = 4,  /* JitteredSampling throws: Sampling.h:642-653 */
  SWZ_ERR_REROOT_UNSUPPORTED = 5,    /* node needs Morton re-rooting, TilingAlgorithms.cpp:444-483 */
  SWZ_ERR_TOO_MANY_POINTS = 6,       /* more than 2^32-65536 points in one batch */
  SWZ_ERR_INTERNAL = 7,
  SWZ_ERR_TILER_FAILED = 8           /* swz_tiler: an earlier batch failed part-way; the node store is incomplete */
};

/* --sampling values, TilerProcess::make_sampling_strategy (core/process/TilerProcess.cpp:491-516)
 * -> Sampling.h:187-308 / :314-416 / :421-471 / :598-759 */
enum { SWZ_RANDOM_GRID = 0, SWZ_GRID_CENTER = 1, SWZ_MIN_DISTANCE = 2, SWZ_JITTERED = 3 };
/* SamplingBehaviour, Sampling.h:170-181 */
enum { SWZ_TAKE_ALL_WHEN_COUNT_BELOW_MAX_POINTS = 0, SWZ_ALWAYS_ADHERE_TO_MIN_SPACING = 1 };
/* --tiling-strategy values, executable/main.cpp:484-497 -> TilingAlgorithmV1 / V3 */
enum { SWZ_ACCURATE = 0, SWZ_FAST = 1 };

typedef struct swz_ctx swz_ctx;

/* Creates a context on HIP device `device` (ordinal).  Fails with SWZ_ERR_HIP when no device. */
int swz_create(swz_ctx** ctx_out, int device);
int swz_destroy(swz_ctx* ctx);
const char* swz_last_error(const swz_ctx* ctx); /* ctx may be NULL: returns the create() error */
int swz_abi_version(void);
/* A context launches on its OWN non-blocking stream: device buffers handed to the *_device entry points must be
 * complete before the call (synchronise, or wait on an event), and the entry points return after their own work
 * finished (they synchronise that stream).  swz_set_stream makes the context run on an existing hipStream_t
 * instead (e.g. torch's current stream), which orders it with the caller's kernels on that stream.  NULL is the
 * device's default stream (a real stream: PyTorch's default); SWZ_OWN_STREAM restores the own stream. */
#define SWZ_OWN_STREAM ((void*)(intptr_t)-1)
int swz_set_stream(swz_ctx* ctx, void* hip_stream);
/* Debug / tuning switches ("SWZ_DEBUG", "SWZ_MD_*", ...).  A context reads the SWZ_* variables
 * of the environment once, when it is created; afterwards they change only through this call (value NULL removes the
 * switch).  None of them changes a result; production code never needs them.  Two can make a call FAIL instead:
 * SWZ_MD_PERSISTENT=1 (an experiment that needs all its workgroups resident at once; SWZ_ERR_INTERNAL when the device
 * is shared) and the limits SWZ_MD_TIME_LIMIT / SWZ_MD_ROUND_LIMIT (a level that exceeds them is abandoned with an
 * error).  A NEGATIVE SWZ_MD_KEYS_BAND is the one exception to "no result changes": it narrows the band in which
 * MIN_DISTANCE repeats a compare on the exact positions below its proven width and exists so that a test can show that
 * the band is needed (tests/test_min_distance_keys.py).
 * SWZ_MD_ROUNDS_BLOCK=0: property mode's first kill pass by per-cell records / the mask loop of round 5 instead of by blocks
 * of cells (same set); SWZ_MD_ROUNDS_BLOCK_MIN_POP (0): the points per cell of a level from which the blocks are used.
 * Round 6, the sparse MIN_DISTANCE levels by blocks of cells (swz_mdblock.hip): SWZ_SP_BLOCK=0 keeps them on the
 * thread-per-point path; SWZ_SP_BLOCK_WIDE=1, _CL, _OWN, _HALO, _PER_CU, _MIN force the point format, the cell level, the
 * LDS capacities, the workgroups per CU, the points a block should hold -- results unchanged (a block that does not fit
 * stops the launch: it is repeated with larger capacities or finer cells, with forced capacities the level falls back;
 * SWZ_SP_BLOCK_CAP_SCALE scales the estimated capacities so that a test can make that happen);
 * SWZ_SP_INCREMENTAL (multi-batch tilers: a batch merged with node files samples only what its points can change) = 0
 * never, = a share in (0, 1]: whenever the files are that much of a level (default: half of it, and files of
 * max_points/2 per node on average), SWZ_SP_INCREMENTAL_MAX (0.35): the largest share of a level that is still
 * sampled as a subset, SWZ_SP_INCREMENTAL_SPREAD: how far from Poisson the subset's blocks are assumed to be -- results
 * unchanged; SWZ_SP_BLOCK_TIMEOUT_MS (default 10 000) bounds how long a wavefront waits for an earlier
 * block: when it expires the call returns SWZ_ERR_INTERNAL, nothing is restarted; SWZ_SP_BLOCK_DBG switches parts of the
 * search off for timing experiments and DOES change the result -- never set it outside a timing experiment (tools/probe.sh variants). */
int swz_set_option(swz_ctx* ctx, const char* name, const char* value);
/* Frees all device workspace held by the context (it regrows on demand). */
int swz_release_workspace(swz_ctx* ctx);
/* Bytes of device workspace currently held. */
uint64_t swz_workspace_bytes(const swz_ctx* ctx);

/* ---- index_points<21>(..., ClampToBounds): core/tiling/OctreeAlgorithms.h:145-197 with
 * calculate_morton_index<21> :64-87.  xyz is N x 3 doubles (AoS).  Outliers are clamped to the
 * bounds IN PLACE (index_point mutates the PointBuffer, :167-169), so xyz is read-write.
 * keys_out[i] is the 63-bit MortonIndex64 of point i. */
int swz_morton_encode(swz_ctx* ctx, double* xyz, uint64_t n, const double bounds_min[3],
                      const double bounds_max[3], uint64_t* keys_out);
int swz_morton_encode_device(swz_ctx* ctx, double* d_xyz, uint64_t n, const double bounds_min[3],
                             const double bounds_max[3], uint64_t* d_keys_out);

/* ---- Range::sort of IndexedPoint64 by key: util/containers/Range.h:62-66, operator<
 * core/tiling/Sampling.h:159-164.  std::sort leaves the order of equal keys unspecified; this
 * sort is stable: perm_out lists original indices ordered by (key, original index).
 * keys_sorted_out may be NULL. */
int swz_sort_by_key(swz_ctx* ctx, const uint64_t* keys, uint64_t n, uint32_t* perm_out,
                    uint64_t* keys_sorted_out);
int swz_sort_by_key_device(swz_ctx* ctx, const uint64_t* d_keys, uint64_t n, uint32_t* d_perm_out,
                           uint64_t* d_keys_sorted_out);

/* ---- sample_points(strategy, begin, end, node_key, node_level, root_bounds, spacing_at_root,
 * behaviour): core/tiling/Sampling.h:799-821.  keys/idx describe a Morton-sorted range of n
 * IndexedPoint64 (idx[i] = row of xyz).  taken_out[i] = 1 when element i belongs to the
 * [begin, partition_point) half of the reference's stable partition, else 0 (both halves keep
 * their order, so the partition itself is a stable compaction by this flag).
 * RANDOM_GRID and GRID_CENTER ignore node_key like the reference does (the range is the node).  For MIN_DISTANCE
 * and JITTERED, which take the node's box from node_key (Sampling.h:441, 622), every key of the range must have
 * node_key's first node_level + 1 octants (checked on the device, SWZ_ERR_BAD_ARG otherwise). */
int swz_sample_points(swz_ctx* ctx, int sampler, uint64_t max_points_per_node, const uint64_t* keys,
                      const uint32_t* idx, uint64_t n, const double* xyz, uint64_t num_points,
                      uint64_t node_key, int32_t node_level, const double root_min[3],
                      const double root_max[3], float spacing_at_root, int behaviour,
                      uint8_t* taken_out, uint64_t* num_taken_out);
/* The same on device buffers (d_xyz: num_points x 3); the flags stay on the device. */
int swz_sample_points_device(swz_ctx* ctx, int sampler, uint64_t max_points_per_node, const uint64_t* d_keys,
                             const uint32_t* d_idx, uint64_t n, const double* d_xyz, uint64_t num_points,
                             uint64_t node_key, int32_t node_level, const double root_min[3],
                             const double root_max[3], float spacing_at_root, int behaviour,
                             uint8_t* d_taken_out, uint64_t* num_taken_out);

/* required_morton_index_depth (core/tiling/Sampling.cpp:29-62 with get_node_level_to_sample_from /
 * first_node_level_obeying_spacing, core/tiling/Node.cpp:37-57): the depth of key bits a sampler needs at a node of
 * level node_level -- what tile_node's terminal / re-root tests use (TilingAlgorithms.cpp:408-444).  Host function,
 * evaluated with the host's libm exactly like the reference (std::log2f of the ratio narrowed to float). */
int32_t swz_required_morton_index_depth(int sampler, int32_t node_level, const double root_min[3],
                                        const double root_max[3], float spacing_at_root);

/* ---- one batch through the tiling algorithm: TilingAlgorithmBase::build_execution_graph
 * (core/tiling/TilingAlgorithms.h:81-85), V1 = ACCURATE (TilingAlgorithms.cpp:577-626),
 * V3 = FAST first iteration + finalize (:1250-1360, :1661-1784). */
typedef struct {
  int32_t sampler;              /* SWZ_RANDOM_GRID ... */
  uint64_t max_points_per_node; /* TilerMetaParameters::max_points_per_node (Tiler.h:64-75) */
  float spacing_at_root;        /* TilerMetaParameters::spacing_at_root */
  uint32_t max_depth;           /* TilerMetaParameters::max_depth (<=0 means 100 upstream) */
  int32_t strategy;             /* SWZ_ACCURATE | SWZ_FAST */
  uint32_t fast_concurrency;    /* FAST: num_indexing_threads, decides the start level
                                   (TilingAlgorithms.cpp:1473-1535) */
  uint32_t flags;               /* SWZ_FLAG_*; 0 = the reference's semantics everywhere */
} swz_tile_params;
/* MIN_DISTANCE "property" mode (not the reference's result): the taken set of every sampled
 * node is still a maximal set of points pairwise >= the node's spacing apart under the reference's compare
 * (squared double distance against the float-squared spacing, GridCell.cpp:52) and deterministic, but the greedy
 * priority is (cell colour, Morton order) instead of Morton order alone: the node is cut into octree cells >= one
 * spacing wide, cells of equal colour (parity of the cell coordinates) are never adjacent and are decided together,
 * eight phases per level instead of thousands of dependent rounds.  Take-all, terminal and all other rules are
 * unchanged.  What the reference's author tests for this sampler (test/TestTiler.cpp:361-421: min distance on
 * sampled nodes) holds; point-for-point equality with the reference does not. */
#define SWZ_FLAG_MIN_DISTANCE_PROPERTY 1u

typedef struct {
  uint64_t num_nodes;         /* nodes that persisted points (incl. FAST reconstructed ones) */
  uint64_t points_visited;    /* sum over levels of the points handed to sample_points/terminal */
  int32_t max_level;          /* deepest node level that took points (-1 = root) */
  int32_t fast_start_levels;  /* FAST: _level_of_start_nodes, else -1 */
  uint32_t num_levels;        /* level iterations executed */
  uint32_t min_distance_rounds; /* MIN_DISTANCE: dependency rounds executed, all levels */
} swz_tile_stats;

/* Outputs are in Morton-sorted order: keys_out[i] ascending (ties by original index),
 * perm_out[i] = original point index, level_out[i] = level of the octree node that persists the
 * point (-1 = root "r"); the node is the first level_out[i]+1 octants of keys_out[i]
 * (node name "r" + digits, TilingAlgorithms.cpp:139).  dup_mask_out (may be NULL; FAST only)
 * has bit (l+1) set when the point is additionally stored in the reconstructed node at level l.
 * xyz is clamped in place like index_point does. */
int swz_tile(swz_ctx* ctx, double* xyz, uint64_t n, const double bounds_min[3],
             const double bounds_max[3], const swz_tile_params* params, uint64_t* keys_out,
             uint32_t* perm_out, int8_t* level_out, uint32_t* dup_mask_out, swz_tile_stats* stats);
int swz_tile_device(swz_ctx* ctx, double* d_xyz, uint64_t n, const double bounds_min[3],
                    const double bounds_max[3], const swz_tile_params* params, uint64_t* d_keys_out,
                    uint32_t* d_perm_out, int8_t* d_level_out, uint32_t* d_dup_mask_out,
                    swz_tile_stats* stats);

/* ---- what PointsPersistence::persist_points needs (core/io/PointsPersistence.h:23-31): the
 * points of every node, contiguous and in Morton order.  From swz_tile's outputs builds
 * order_out[n] (sorted positions grouped by node; nodes ordered by (level, key prefix)) and the
 * node table.  node_* arrays must hold max_nodes entries; *num_nodes_out is the count found
 * (SWZ_ERR_BAD_ARG when it exceeds max_nodes).  Host buffers. */
int swz_build_node_lists(swz_ctx* ctx, const uint64_t* keys_sorted, const int8_t* level, uint64_t n,
                         uint32_t* order_out, uint64_t max_nodes, int8_t* node_level_out,
                         uint64_t* node_key_out, uint64_t* node_offset_out, uint64_t* node_count_out,
                         uint64_t* num_nodes_out);

/* Same on the device: d_keys_sorted / d_level are swz_tile_device's outputs, d_order_out[n] is device
 * memory; the node table (at most max_nodes entries) is written to the host arrays. */
int swz_build_node_lists_device(swz_ctx* ctx, const uint64_t* d_keys_sorted, const int8_t* d_level, uint64_t n,
                                uint32_t* d_order_out, uint64_t max_nodes, int8_t* node_level_out,
                                uint64_t* node_key_out, uint64_t* node_offset_out, uint64_t* node_count_out,
                                uint64_t* num_nodes_out);

/* ---- node payload: what a lossless PointsPersistence stores for a node.
 * The attribute columns of a point batch, SoA like PointBuffer (core/datastructures/PointBuffer.h:109-121,
 * 292-304); a NULL column is absent.  Indices double as bit numbers of BinaryPersistence's properties
 * bitmask (core/io/BinaryPersistence.h:24-35) and give the order of the arrays in a node file. */
enum {
  SWZ_ATTR_RGB = 0,                 /* 3 x uint8  */
  SWZ_ATTR_NORMAL = 1,              /* 3 x float  */
  SWZ_ATTR_INTENSITY = 2,           /* uint16     */
  SWZ_ATTR_CLASSIFICATION = 3,      /* uint8      */
  SWZ_ATTR_EDGE_OF_FLIGHT_LINE = 4, /* uint8      */
  SWZ_ATTR_GPS_TIME = 5,            /* double     */
  SWZ_ATTR_NUMBER_OF_RETURNS = 6,   /* uint8      */
  SWZ_ATTR_RETURN_NUMBER = 7,       /* uint8      */
  SWZ_ATTR_POINT_SOURCE_ID = 8,     /* uint16     */
  SWZ_ATTR_SCAN_DIRECTION_FLAG = 9, /* uint8      */
  SWZ_ATTR_SCAN_ANGLE_RANK = 10,    /* int8       */
  SWZ_ATTR_USER_DATA = 11,          /* uint8      */
  SWZ_ATTR_COUNT = 12
};
typedef struct {
  void* column[SWZ_ATTR_COUNT];
} swz_attribute_columns;
/* bytes per point of attribute a (0 for an invalid index) */
uint32_t swz_attribute_row_bytes(int attribute);

/* Permuted gather into node order: row i of every output column = row d_perm[d_order[i]] of the input
 * column (positions N x 3 doubles plus every attribute present in BOTH column sets).  d_perm and d_order
 * are swz_tile_device's perm and swz_build_node_lists_device's order.  After this the points of node k
 * are rows [node_offset[k], node_offset[k] + node_count[k]) of every column, in Morton order -- exactly
 * the iterator range tile_node hands to persist_points (TilingAlgorithms.cpp:232-236, 316-322). */
int swz_gather_payload_device(swz_ctx* ctx, const uint32_t* d_perm, const uint32_t* d_order, uint64_t n,
                              const double* d_xyz, const swz_attribute_columns* d_in, double* d_xyz_out,
                              const swz_attribute_columns* d_out);

/* BinaryPersistence node files (core/io/BinaryPersistence.h:45-193, BinaryPersistence.cpp:200-375):
 * uint32 properties bitmask, uint64 point count, positions (24 B each), then every present attribute
 * array in the order RGB, normal, intensity, classification, edge of flight line, GPS time, number of
 * returns, return number, point source id, scan angle rank, scan direction flag, user data (bit order
 * except that the reference writes bit 10 before bit 9); little-endian, no padding.  compressed != 0 wraps the same bytes in one zlib stream (level 1,
 * "<name>.binz"), else "<name>.bin".  Host functions, no GPU involved; ctx only carries the error text
 * and may be NULL.
 *   swz_bin_write_node: columns point at the node's first row.  count == 0 writes nothing (like
 *     persist_points).
 *   swz_bin_read_header / swz_bin_read_node: retrieve_points; read_node fills xyz_out (count x 3) and
 *     every non-NULL column whose bit is set in the file.
 *   swz_bin_persist_nodes: one file per node of a node table, named "r" + octant digits
 *     (TilingAlgorithms.cpp:139) in directory dir; xyz / columns are the gathered payload of the batch.  The files are
 *     written by several host threads (option SWZ_BIN_WRITER_THREADS; default: the host's, at most 32). */
int swz_bin_write_node(swz_ctx* ctx, const char* path, uint64_t count, const double* xyz,
                       const swz_attribute_columns* columns, int compressed);
int swz_bin_read_header(swz_ctx* ctx, const char* path, int compressed, uint32_t* bitmask_out, uint64_t* count_out);
int swz_bin_read_node(swz_ctx* ctx, const char* path, int compressed, double* xyz_out,
                      const swz_attribute_columns* columns_out);
int swz_bin_persist_nodes(swz_ctx* ctx, const char* dir, uint64_t num_nodes, const int8_t* node_level,
                          const uint64_t* node_key, const uint64_t* node_offset, const uint64_t* node_count,
                          const double* xyz, const swz_attribute_columns* columns, int compressed);
/* "r" + octant digits of a node; name_out must hold 23 bytes */
int swz_node_name(int8_t node_level, uint64_t node_key, char* name_out);

/* ---- hierarchy metadata of the node table; host functions.
 *   swz_node_name_entwine / swz_node_from_entwine_name: "D-X-Y-Z" (depth = number of octants, then the node's
 *     grid coordinates at that depth; OctreeNodeIndex::to_string_entwine / from_string, core/datastructures/
 *     OctreeNodeIndex.h:556-573, 480-540).  name_out must hold 72 bytes.
 *   swz_node_bounds: the node's box by descending from the root box octant by octant (get_octant_bounds,
 *     core/tiling/OctreeAlgorithms.cpp:3-18, the way Cesium3DTilesPersistence::on_write_node walks down,
 *     core/io/Cesium3DTilesPersistence.cpp:100-112).
 *   swz_node_geometric_error: spacing_at_root / 2^depth (Cesium3DTilesPersistence.cpp:91-92). */
int swz_node_name_entwine(int8_t node_level, uint64_t node_key, char* name_out);
int swz_node_from_entwine_name(const char* name, int8_t* node_level_out, uint64_t* node_key_out);
int swz_node_bounds(int8_t node_level, uint64_t node_key, const double root_min[3], const double root_max[3],
                    double min_out[3], double max_out[3]);
double swz_node_geometric_error(int8_t node_level, float spacing_at_root);

/* Tileset tree of a node table (Cesium3DTilesPersistence::on_write_node / write_tilesets, core/io/
 * Cesium3DTilesPersistence.cpp:80-156, 175-199): one entry per node of the table AND per ancestor the table does not
 * list, ordered by (level, Morton index) so that the children of an entry are contiguous, by octant.
 *   has_content: the node is in the table (it has a point file); is_tileset_root: a new tileset.json starts here
 *   (every third level); bounds are the node box translated by global_offset (NULL = none).
 * out == NULL only counts (*num_out). */
typedef struct {
  int8_t level;              /* -1 = root "r" */
  uint8_t has_content;
  uint8_t is_tileset_root;
  uint8_t reserved;
  uint32_t num_children;
  uint64_t key;
  int64_t parent;            /* index into the output, -1 for the root */
  int64_t first_child;       /* -1 when there are none */
  double geometric_error;
  double bounds_min[3], bounds_max[3];
} swz_tileset_node;
int swz_tileset_build(uint64_t num_nodes, const int8_t* node_level, const uint64_t* node_key, const double root_min[3],
                      const double root_max[3], float spacing_at_root, const double global_offset[3], uint64_t max_out,
                      swz_tileset_node* out, uint64_t* num_out);

/* ---- LAS point records -> positions + attribute columns: the step right in
 * front of the path.  The reference reads points through LASzip into a laszip_point and converts them in
 * position_from_las_point (core/io/LASFile.cpp:79-94: offset + X * scale per axis, then clamped into the
 * header's bounding box) and las_read_points_into (:578-632: RGB >> 8, the other attributes copied).  LASzip
 * itself (github.com/LASzip/LASzip, built from LAStools per the reference's README.md:24-37, version not
 * pinned there) is not part of the reference tree; for UNCOMPRESSED files its reader only unpacks the fixed
 * point data record of the LAS specification, which is restated here for formats 0-10 (LASFile.cpp:421-426 lists the
 * formats that carry RGB: 2, 3, 5, 7, 8, 10):
 *   formats 0-5 (LAS 1.2 / 1.3): X,Y,Z i32 | intensity u16 | return number:3, number of returns:3, scan direction:1, edge
 *   of flight line:1 | classification:5 (+3 flag bits) | scan angle rank i8 | user data u8 | point source id u16 (20
 *   bytes) | format 1,3,4,5: gps time f64 | format 2,3,5: R,G,B u16 | format 4,5: 29 bytes of wave packet (skipped);
 *   formats 6-10 (LAS 1.4): X,Y,Z i32 | intensity u16 | return number:4, number of returns:4 | classification flags:4,
 *   scanner channel:2, scan direction:1, edge of flight line:1 | classification u8 | user data u8 | scan angle i16 (0.006
 *   degree) | point source id u16 | gps time f64 (30 bytes) | format 7,8,10: R,G,B u16 | 8,10: NIR u16 | 9,10: wave
 *   packet -- mapped onto the legacy fields the reference reads the way LASzip's raw reader does (returns above 7
 *   saturate, classes above 31 read 0, the scan angle is rounded to degrees and clamped to a signed byte).
 * d_records: n records of record_bytes each (>= the format's size; trailing extra bytes are skipped), device
 * memory, 4-byte aligned.  Columns absent from d_out are skipped; attributes the format lacks (gps time in
 * format 0/2, RGB in 0/1) are written as 0 like an untouched laszip_point. */
typedef struct {
  double scale[3];       /* laszip_header::x_scale_factor, y_, z_ */
  double offset[3];      /* x_offset, ... */
  double min[3];         /* min_x, ... : positions are clamped into [min, max] */
  double max[3];
  uint32_t point_format; /* 0..10 */
  uint32_t record_bytes; /* point data record length */
} swz_las_layout;
int swz_las_decode_device(swz_ctx* ctx, const uint8_t* d_records, uint64_t n, const swz_las_layout* layout,
                          double* d_xyz_out, const swz_attribute_columns* d_out);

/* ---- multi-GPU sharding: one context per GPU, points owned by their top
 * Morton bits (level-0 octant, MortonIndex::get_octant_at_level(0), MortonIndex.h:133-138), so every
 * node at level >= 0 lives on exactly one GPU.  The reference has no counterpart (it is a single
 * process); the exchange itself (RCCL all-to-all) is done by the host driver, the library provides
 * the device-side pieces:
 *   swz_partition_by_octant_device: perm groups point indices by octant 0..7 (stable inside an
 *     octant); counts_out[o] = points of octant o.
 *   swz_shard_begin_device: indexes + sorts the shard's points (d_xyz_local stays referenced until
 *     swz_shard_finish_device returns and may be clamped in place) and samples the ROOT node, whose
 *     take-all/sample decision uses shard->global_points.  For MIN_DISTANCE the root couples the
 *     shards: the points the root took on all lower octants are passed as ghosts (they sort first and
 *     are accepted again, rejecting exactly what the single-GPU sweep would reject).
 *   swz_shard_root_taken_device: positions (N x 3) of the LOCAL points the root took, Morton order:
 *     the ghosts to hand to the shards owning higher octants.
 *   swz_shard_finish_device: tiles levels >= 0 and writes the outputs of swz_tile_device for the n
 *     local points (keys ascending, perm = index into the local xyz, level).
 *   A shard without points (n == 0: its octants are empty) is valid everywhere: presort is a no-op, begin takes
 *   nothing of the root, finish reports zero points. */
typedef struct {
  uint64_t global_points;      /* points of the whole batch over all shards */
  const double* d_ghost_xyz;   /* device, num_ghosts x 3; all ghosts must lie in lower octants.  When the
                                  ghosts end exactly where d_xyz_local starts, no staging copy is made */
  uint64_t num_ghosts;
} swz_shard_info;
int swz_partition_by_octant_device(swz_ctx* ctx, const uint64_t* d_keys, uint64_t n, uint32_t* d_perm_out,
                                   uint64_t counts_out[8]);
/* Optional, before swz_shard_begin_device with the same d_xyz_local / n: does the part of it that does not
 * depend on the ghosts (index + sort + gather of the local points), leaving room for up to ghost_capacity
 * ghosts.  When the root couples the shards (MIN_DISTANCE) every shard calls this right after the exchange,
 * all at the same time, and only the root node itself remains in the chain that hands the ghosts from shard
 * to shard (ghosts sort in front of all local points: they lie in lower octants). */
int swz_shard_presort_device(swz_ctx* ctx, const double* d_xyz_local, uint64_t n, const double bounds_min[3],
                             const double bounds_max[3], const swz_tile_params* params, uint64_t ghost_capacity);
int swz_shard_begin_device(swz_ctx* ctx, const double* d_xyz_local, uint64_t n, const double bounds_min[3],
                           const double bounds_max[3], const swz_tile_params* params,
                           const swz_shard_info* shard, uint64_t* num_root_taken_out);
int swz_shard_root_taken_device(swz_ctx* ctx, double* d_xyz_out);
int swz_shard_finish_device(swz_ctx* ctx, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out,
                            swz_tile_stats* stats);
/* FAST (TilingAlgorithmV3, the reference's default: executable/main.cpp:299-301) on a sharded batch.  The start level comes
 * from the distribution of the WHOLE batch (estimate_start_node_level_in_octree, TilingAlgorithms.cpp:1473-1535); start nodes
 * lie at level >= 2, inside one shard's octants, so the levels from there down and the reconstruction of the skipped levels
 * down to level 0 (reconstruct_left_out_nodes, :1717-1784) are local, and only the root is reconstructed across the shards:
 *   swz_shard_fast_begin_device: indexes + sorts the shard's points; prefix_counts_out[8^6] (host) = its points per
 *     6-octant prefix.  The driver sums the counts of all shards and calls swz_fast_start_level_from_counts.
 *   swz_shard_fast_run: the levels from start_level - 1 down, then the skipped levels down to 0;
 *     num_root_candidates_out = the points this shard's level-0 nodes hold.
 *   swz_shard_fast_root_candidates_device: their keys and positions, in key order.  The root samples the candidates of ALL
 *     shards, one behind the other in shard order (= octant order, the order the reference appends its children in), with
 *     AlwaysAdhereToMinSpacing: swz_sample_points_device(..., node_level -1, SWZ_ALWAYS_ADHERE_TO_MIN_SPACING, ...).
 *   swz_shard_fast_set_root_device: d_taken = the flags of this shard's candidates (bit 0 of dup is set for them).
 *   swz_shard_fast_finish_device: the outputs of swz_tile_device with the FAST strategy for the local points (d_dup_out:
 *     bit l + 1 set = the point is also stored in its ancestor at node level l).
 *   A shard without points takes part in every step with zero counts. */
int swz_shard_fast_begin_device(swz_ctx* ctx, const double* d_xyz_local, uint64_t n, const double bounds_min[3],
                                const double bounds_max[3], const swz_tile_params* params, uint32_t* prefix_counts_out);
int swz_shard_fast_run(swz_ctx* ctx, int32_t start_level, uint64_t* num_root_candidates_out);
int swz_shard_fast_root_candidates_device(swz_ctx* ctx, uint64_t* d_keys_out, double* d_xyz_out);
int swz_shard_fast_set_root_device(swz_ctx* ctx, const uint8_t* d_taken);
int swz_shard_fast_finish_device(swz_ctx* ctx, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out,
                                 uint32_t* d_dup_out, swz_tile_stats* stats);

/* ---- multi-batch tiling (BASELINE config 5's single-GPU shape).
 * The reference calls TilingAlgorithmBase::build_execution_graph once per batch of at most internal_cache_size
 * points (core/process/Tiler.cpp:509-510; executable/main.cpp:233-236, default 10 M) on one algorithm object;
 * every node a batch reaches re-reads its file, re-keys those points relative to the node
 * (read_pnts_from_disk, core/tiling/TilingAlgorithms.cpp:50-109), merges them with the new ones
 * (merge_node_data_sorted, core/tiling/Node.cpp:4-22), samples with AlwaysAdhereToMinSpacing (:272-275) and
 * REPLACES the file (BinaryPersistence.h:46-57).  FAST: later iterations (:1362-1453, 1620-1659) + finalize
 * (:1661-1784).  swz_tiler keeps the node files in device memory: one context, one tiler per data set.
 *   Point ids count the points of all batches in input order (batch 0's points are 0..n0-1, ...).
 *   swz_tiler_add_batch_device: d_xyz (n x 3, device) is clamped in place like index_point does and copied
 *     into the tiler's position pool.
 *   swz_tiler_stage_batch / swz_tiler_tile_staged: the same from HOST memory (pinned: swz_host_alloc_pinned, or
 *     hipHostRegister'ed by the caller), copied with hipMemcpyAsync on a copy stream straight into the pools; up
 *     to two batches may be staged, so the copy of batch k+1 runs while batch k is tiled:
 *         stage(0); for k: { if (k+1 < K) stage(k+1); tile_staged(); }
 *     attrs_host (may be NULL) are the batch's attribute columns (host pointers); every batch must carry the same
 *     set.  The host copy of the positions is NOT clamped; the pools (swz_tiler_pools_device) hold the clamped ones.
 *   swz_tiler_finalize: TilingAlgorithmBase::finalize (FAST: reconstruct_left_out_nodes); no more batches after.
 *   swz_tiler_export_device / swz_tiler_node_table: the node files: nodes ordered by (level, Morton index);
 *     entries [node_offset[k], +node_count[k]) of keys/ids/level are node k's points in file order (FAST stores
 *     copies of a point in reconstructed ancestors, so num_stored >= num_points).  ids index the pools: rows of
 *     the files are swz_gather_payload_device(ctx, d_ids, NULL, num_stored, pool_xyz, pool_attrs, ...).
 *   rekey_inversions: re-keyed file contents whose order was not ascending any more (a point within an ulp of a
 *     cell boundary); the reference merges them unsorted for a lossless persistence (:103-106), this library sorts
 *     them like the reference does for a lossy one -- results may differ from the reference only when > 0.
 *   A node that needs Morton re-rooting (:444-483) is re-rooted like the reference does (its subtree's files are
 *     ordered by the re-rooted keys); only the single-batch swz_tile*, whose outputs cannot express that order,
 *     returns SWZ_ERR_REROOT_UNSUPPORTED there. */
typedef struct swz_tiler swz_tiler;
typedef struct {
  uint64_t num_points;       /* points added so far */
  uint64_t num_stored;       /* entries in node files */
  uint64_t num_nodes;        /* node files */
  uint64_t num_batches;
  uint64_t rekey_inversions;
  int32_t fast_start_levels; /* FAST: _level_of_start_nodes once known, else -1 */
  uint64_t staged_bytes;     /* bytes copied by swz_tiler_stage_batch */
  double staged_wait_ms;     /* time swz_tiler_tile_staged waited for its copy (0 = fully hidden behind tiling) */
} swz_tiler_info;
int swz_tiler_create(swz_ctx* ctx, const double bounds_min[3], const double bounds_max[3],
                     const swz_tile_params* params, uint64_t capacity_hint_points, swz_tiler** tiler_out);
int swz_tiler_destroy(swz_tiler* tiler); /* before swz_destroy of its context */
int swz_tiler_add_batch_device(swz_tiler* tiler, double* d_xyz, uint64_t n, swz_tile_stats* stats);
int swz_tiler_stage_batch(swz_tiler* tiler, const double* xyz_host, uint64_t n, const swz_attribute_columns* attrs_host);
int swz_tiler_tile_staged(swz_tiler* tiler, swz_tile_stats* stats);
/* stage + tile of one host batch, no overlap */
int swz_tiler_add_batch(swz_tiler* tiler, const double* xyz_host, uint64_t n, const swz_attribute_columns* attrs_host,
                        swz_tile_stats* stats);
int swz_tiler_finalize(swz_tiler* tiler, swz_tile_stats* stats);
int swz_tiler_get_info(swz_tiler* tiler, swz_tiler_info* info);
/* SPILL.  The pools (24 bytes + the attribute rows of every point of the data set, by point id) are the bulk of a
 * tiler's memory.  When they cannot grow in device memory -- hipMalloc is out of memory, or the context's workspace
 * would pass SWZ_TILER_DEVICE_BUDGET_MB -- they move to page-locked host memory that is mapped into the device's address
 * space, and the kernels keep reading and writing them in place over the host link (a batch's own points once, the
 * cached points a batch pulls in, by id; the attribute columns only when files are exported).  Results do not change;
 * swz_tiler_pools_device then hands out device-accessible HOST pointers.  The node store (12 bytes per stored point and
 * side) stays in device memory.  Option SWZ_TILER_SPILL: "auto" (default), "host" (from the start), "off" (fail with
 * SWZ_ERR_HIP as before).  Replaces nothing in the reference, whose node files live on disk between batches
 * (core/tiling/TilingAlgorithms.cpp:50-109); this is what bounds the size of a data set per GPU here. */
int swz_tiler_pool_residency(swz_tiler* tiler, uint64_t* device_bytes_out, uint64_t* host_bytes_out);
/* Makes room in the pools for `total_points` points of the data set NOW (what the next batch would do on its own).  A
 * driver that must know where the pools will live while a batch is tiled -- the joint MIN_DISTANCE root of one process per
 * GPU exports them as IPC handles, which page-locked host memory does not have -- reserves first and asks
 * swz_tiler_pool_residency afterwards. */
int swz_tiler_reserve(swz_tiler* tiler, uint64_t total_points);
/* The same for the node store (the files' {key, point id} entries, two sides per octree level): a side that finds no device
 * memory, or would push the workspace over SWZ_TILER_DEVICE_BUDGET_MB, is placed in mapped pinned host memory like a pool
 * and the merges stream through it over the host link. */
int swz_tiler_store_residency(swz_tiler* tiler, uint64_t* device_bytes_out, uint64_t* host_bytes_out);
int swz_tiler_export_device(swz_tiler* tiler, uint64_t* d_keys_out, uint32_t* d_ids_out, int8_t* d_level_out);
int swz_tiler_node_table(swz_tiler* tiler, uint64_t max_nodes, int8_t* node_level_out, uint64_t* node_key_out,
                         uint64_t* node_offset_out, uint64_t* node_count_out, uint64_t* num_nodes_out);
/* ONE batch as node files -- the single-batch call for batches whose nodes may need Morton re-rooting
 * (core/tiling/TilingAlgorithms.cpp:444-483).  swz_tile's outputs (a level per point of the sorted batch, the node being
 * a prefix of the point's root key) cannot express a re-rooted subtree -- its children are contiguous chunks of the
 * re-indexed points, :479-482 -- and swz_tile returns SWZ_ERR_REROOT_UNSUPPORTED there; this pair of calls returns the
 * node table and the files' contents instead, in the form swz_tiler_node_table / swz_tiler_export_device hand out (it
 * runs the batch through a tiler of its own: one batch, finalize).  In two steps so that the caller can size its buffers:
 *   swz_tile_nodes_begin_device: tiles d_xyz (device, clamped in place like index_point does); *num_stored_out entries
 *     in *num_nodes_out node files;
 *   swz_tile_nodes_end_device: the files -- d_keys_out / d_ids_out / d_level_out (device, num_stored entries, any may be
 *     NULL; ids are rows of d_xyz) -- and the node table (host arrays of max_nodes >= num_nodes entries: level, Morton
 *     index, offset and count of every node, nodes ordered by (level, Morton index)); closes the call.
 * swz_tile_nodes_end_device with every pointer NULL and max_nodes 0 just closes it.  One open call per context; a context
 * that has a swz_tiler open cannot take it (SWZ_ERR_BAD_ARG). */
int swz_tile_nodes_begin_device(swz_ctx* ctx, double* d_xyz, uint64_t n, const double bounds_min[3], const double bounds_max[3],
                                const swz_tile_params* params, uint64_t* num_stored_out, uint64_t* num_nodes_out,
                                swz_tile_stats* stats);
int swz_tile_nodes_end_device(swz_ctx* ctx, uint64_t* d_keys_out, uint32_t* d_ids_out, int8_t* d_level_out, uint64_t max_nodes,
                              int8_t* node_level_out, uint64_t* node_key_out, uint64_t* node_offset_out,
                              uint64_t* node_count_out);
/* the pools by point id: positions (num_points x 3, clamped) and the attribute columns staged so far */
int swz_tiler_pools_device(swz_tiler* tiler, const double** d_xyz_out, swz_attribute_columns* d_attrs_out);
/* ---- one tiler per GPU of a multi-GPU run (BASELINE config 5: sharded + multi-batch).  Points are owned by their
 * level-0 octant as in swz_shard_* above; every shard keeps the subtrees of its octants and ITS part of the root's
 * file.  Per batch, after the exchange of the batch's points (and attribute columns) by octant:
 *   swz_tiler_shard_begin_device: d_xyz / d_attrs (device; attrs may be NULL) are the points this shard received
 *     (n may be 0).  Indexes + sorts them and decides the ROOT node: take-all / sample from the counts of the whole
 *     root (tile_internal_node, TilingAlgorithms.cpp:272-275: global_root_stored > 0 forces sampling); the whole
 *     local part of the root's file takes part whenever the batch has points anywhere; for MIN_DISTANCE
 *     d_ghost_xyz are the positions of what the root holds on all LOWER shards after their begin of THIS batch
 *     (swz_tiler_level_positions_device(tiler, -1, ...) there): they sort first and are taken again.
 *     *root_file_count_out = points of the root's file on this shard afterwards.
 *   swz_tiler_shard_finish: the levels >= 0 (local).
 * ACCURATE and exact MIN_DISTANCE only.  The node table of a shard lists the root with the shard's part of its file;
 * the root's file is the concatenation over the shards in rank order. */
typedef struct {
  uint64_t global_new_points;  /* points of this batch over all shards */
  uint64_t global_root_stored; /* points in the root's file over all shards before this batch */
  const double* d_ghost_xyz;   /* device, num_ghosts x 3 */
  uint64_t num_ghosts;
} swz_tiler_shard_info;
int swz_tiler_shard_begin_device(swz_tiler* tiler, double* d_xyz, uint64_t n, const swz_attribute_columns* d_attrs,
                                 const swz_tiler_shard_info* info, uint64_t* root_file_count_out);
int swz_tiler_shard_finish(swz_tiler* tiler, swz_tile_stats* stats);
/* FAST (TilingAlgorithmV3) on a sharded data set: swz_tiler_shard_begin_device indexes and sorts only (FAST has no root
 * step per batch); after the FIRST batch's begin every shard reports its points per 6-octant prefix (2^18 counts, host),
 * the driver sums them over the shards, derives the start level (TilingAlgorithms.cpp:1473-1535) and tells every shard
 * before swz_tiler_shard_finish.  At the end of the data set: swz_tiler_shard_fast_finalize_local rebuilds the skipped
 * levels of the shard's octants down to level 0; the driver samples the root from the level-0 files of ALL shards (in
 * shard order, swz_tiler_level_positions_device(0)) and hands every shard the flags of its entries. */
int swz_tiler_shard_fast_histogram(swz_tiler* tiler, uint32_t* counts_out /* 262144, host */);
int swz_fast_start_level_from_counts(const uint64_t* counts /* 262144, host */, uint32_t fast_concurrency, int32_t* start_level_out);
int swz_tiler_shard_set_start_level(swz_tiler* tiler, int32_t start_level);
int swz_tiler_shard_fast_finalize_local(swz_tiler* tiler, swz_tile_stats* stats);
int swz_tiler_shard_fast_set_root(swz_tiler* tiler, const uint8_t* d_taken);
/* A batch that fails part-way (workspace out of memory, a MIN_DISTANCE level that does not terminate, a HIP error)
 * leaves the node store half updated: the tiler is then POISONED -- every later swz_tiler_* call except
 * swz_tiler_destroy / swz_tiler_get_info returns SWZ_ERR_TILER_FAILED and names the original failure.  A driver that
 * keeps several tilers in step (one per GPU) poisons the others itself when one of them fails. */
int swz_tiler_poison(swz_tiler* tiler, const char* why);
/* points stored in the files of one octree level (-1 = root) and their positions in file order */
int swz_tiler_level_count(swz_tiler* tiler, int level, uint64_t* count_out);
int swz_tiler_level_positions_device(swz_tiler* tiler, int level, double* d_xyz_out);
/* ---- one batch sharded over several GPUs from ONE host process (the reference's host is a
 * single C++ process, so this is its multi-GPU drop-in: TilingAlgorithmBase::build_execution_graph,
 * core/tiling/TilingAlgorithms.cpp:1362-1784, for a batch whose points lie on several devices).
 *   swz_group_create: one context per shard on devices[shard] (1, 2, 4 or 8 shards; shard s owns the level-0 octants
 *     o with o * num_shards / 8 == s).  transport 0 = peer copies (hipMemcpyPeerAsync; several shards may share a
 *     device), 1 = RCCL grouped ncclSend / ncclRecv on communicators made by ncclCommInitAll (distinct devices only;
 *     librccl is loaded on demand, the library does not link it).
 *   swz_group_tile: d_xyz[s] / n[s] are the points that currently lie on shard s's device (any octants; clamped in
 *     place), d_attrs (may be NULL) is an array of num_shards column sets: d_attrs[s] holds shard s's attribute
 *     columns (device, n[s] rows; the same attributes on every shard), which travel with the points.  One host thread per shard: encode, group by destination, ONE exchange step, root node (for
 *     MIN_DISTANCE swept by all shards at once, cells at a lower octant's face reading that shard's records through peer
 *     access -- cubic bounds, one address space; otherwise the chain of ghosts from lower to higher shards), levels.  results[s] describes what shard s
 *     ended up with, exactly as swz_shard_finish_device does: device pointers owned by the shard's context, valid
 *     until the group's next call.  ACCURATE, or FAST (TilingAlgorithmV3: the start level from the whole batch's distribution, the
 *     skipped levels rebuilt per shard and the root from all shards' level-0 nodes on shard 0 -- swz_shard_fast_*; d_dup then
 *     says in which ancestors a point is stored as well); MIN_DISTANCE exact or in property mode (below the root).
 *   swz_group_ctx: the shard's context, e.g. for swz_copy_to_host, swz_build_node_lists_device or
 *     swz_gather_payload_device on that shard's results. */
typedef struct swz_group swz_group;
typedef struct {
  const double* d_xyz;     /* the shard's points after the exchange (num_points x 3) */
  const uint64_t* d_keys;  /* ascending */
  const uint32_t* d_perm;  /* index into d_xyz */
  const int8_t* d_level;
  swz_attribute_columns attrs; /* the attribute columns that travelled with the points (rows like d_xyz) */
  uint64_t num_points;
  swz_tile_stats stats;
  const uint32_t* d_dup;   /* FAST: bit l + 1 set = the point is also stored in its ancestor at node level l (else NULL) */
} swz_group_result;
int swz_group_create(int num_shards, const int* devices, int transport, swz_group** group_out);
int swz_group_destroy(swz_group* group);
const char* swz_group_last_error(const swz_group* group);
int swz_group_num_shards(const swz_group* group);
swz_ctx* swz_group_ctx(swz_group* group, int shard);
/* host wall clock of shard `shard` in the last swz_group_tile, ms since the call began: exchange done, root begun, root
 * done, levels done (the MIN_DISTANCE root: all shards at once when the level can be decided on keys, else in turns) */
int swz_group_shard_timing(const swz_group* group, int shard, double ms_out[4]);
int swz_group_tile(swz_group* group, double* const* d_xyz, const swz_attribute_columns* d_attrs, const uint64_t* n,
                   const double bounds_min[3], const double bounds_max[3], const swz_tile_params* params,
                   swz_group_result* results);

/* A data set that arrives in several batches, sharded over the group's GPUs from ONE C++ process (BASELINE config 5;
 * the reference is one process too: core/process/Tiler.cpp:189-198, 499-527).  One swz_tiler per shard owns the subtrees
 * of the shard's level-0 octants and ITS part of the root's file.  Per batch: encode, group by destination, ONE
 * exchange step of the point rows and of every attribute column, then the root node -- its take-all / sample decision
 * uses the counts of the whole root, and for MIN_DISTANCE the shards take turns with the lower shards' root files as
 * ghosts -- and, without communication, the levels below.
 *   swz_group_tiler_open / _close: creates / destroys the shards' tilers (ACCURATE or FAST strategy, exact samplers).
 *     FAST: per batch every shard adds the prefix histogram of its keys, the sum over the shards gives the start level the
 *     single tiler would choose (TilingAlgorithms.cpp:1473-1535), the levels below it run per shard.
 *   swz_group_add_batch: d_xyz[s] / n[s] / d_attrs[s] as in swz_group_tile; stats (may be NULL) receives one entry per shard.
 *   swz_group_stage_batch / swz_group_tile_staged: the same from PINNED host memory -- the copies of batch k + 1 run on a
 *     copy stream per shard (hipMemcpyAsync) beside the kernels of batch k; at most two batches are staged.
 *   swz_group_finalize: ends the data set.  FAST rebuilds the skipped levels: each shard those of its octants down to
 *     level 0, shard 0 the root from the level-0 files of all shards (TilingAlgorithms.cpp:1661-1784); ACCURATE has
 *     nothing left to do.
 *   swz_group_tiler: shard s's tiler, for swz_tiler_node_table / _export_device / _pools_device / _get_info.  The
 *     root's file is the concatenation of the shards' parts in shard order.
 * Every exchange is preceded by a status vote: a shard that failed makes all shards skip the step and return its error
 * (nobody waits for a peer that is gone).  A batch that fails on one shard poisons every shard's tiler (swz_tiler_poison). */
int swz_group_tiler_open(swz_group* group, const double bounds_min[3], const double bounds_max[3], const swz_tile_params* params,
                         uint64_t capacity_hint_per_shard);
int swz_group_tiler_close(swz_group* group);
swz_tiler* swz_group_tiler(swz_group* group, int shard);
int swz_group_add_batch(swz_group* group, double* const* d_xyz, const swz_attribute_columns* d_attrs, const uint64_t* n,
                        swz_tile_stats* stats_per_shard);
int swz_group_stage_batch(swz_group* group, const double* const* xyz_host, const swz_attribute_columns* attrs_host, const uint64_t* n);
int swz_group_tile_staged(swz_group* group, swz_tile_stats* stats_per_shard);
int swz_group_finalize(swz_group* group, swz_tile_stats* stats_per_shard);

/* ---- the MIN_DISTANCE root of a sharded batch swept by all ranks at once, ONE PROCESS PER GPU (the torch driver;
 * swz_group_tile does this by itself for the shards of one process).  Instead of the chain of ghosts from rank to rank
 * (swz_shard_begin_device with ghosts), every rank sweeps the root cells of its own octants concurrently and cells at the
 * face of a lower octant read that rank's records in place, through IPC mappings (hipIpcGetMemHandle /
 * hipIpcOpenMemHandle) that the library exchanges through the driver's all-gather:
 *   swz_shard_joint_root_possible: exact MIN_DISTANCE, cubic bounds, a root that can be decided on key coordinates.
 *   swz_shard_joint_root_begin: before swz_shard_begin_device (without ghosts).  exchange(arg, mine, bytes, all) must
 *     all-gather `bytes` bytes of every rank into all[rank * bytes] and return 0; the library calls it twice per batch
 *     on every rank, from inside swz_shard_begin_device (or from swz_shard_joint_root_meet).
 *   swz_shard_joint_root_meet: a rank whose swz_shard_begin_device was not called (it owns no points) or failed before
 *     its sweep began calls this instead, so that the others are not left waiting in the exchange.
 *   swz_shard_joint_root_end: after a barrier of the driver behind swz_shard_begin_device (the other ranks may read this
 *     rank's root arrays until they have finished theirs); unmaps.
 * Results are those of the chain, point for point (tests/test_sharded_gloo.py, two processes on one GPU). */
typedef int (*swz_exchange_fn)(void* arg, const void* mine, uint64_t bytes, void* all);
int swz_shard_joint_root_possible(swz_ctx* ctx, const swz_tile_params* params, const double bounds_min[3], const double bounds_max[3]);
int swz_shard_joint_root_begin(swz_ctx* ctx, int shard, int num_shards, swz_exchange_fn exchange, void* arg);
/* Collective, before the first batch: can every rank map and read the lower ranks' device memory (IPC handles through the
 * same all-gather callback: two exchanges, then one per rank while the mappings are closed in turns)?  *usable = 1 only when all ranks could; a driver that gets 0 keeps the chain of
 * ghosts -- nothing has been started that would have to be undone. */
int swz_shard_joint_root_probe(swz_ctx* ctx, int shard, int num_shards, swz_exchange_fn exchange, void* arg, int* usable);
int swz_shard_joint_root_meet(swz_ctx* ctx, int ok);
int swz_shard_joint_root_end(swz_ctx* ctx);

/* page-locked host memory for the staging entry points (hipHostMalloc / hipHostFree) */
int swz_host_alloc_pinned(uint64_t bytes, void** out);
int swz_host_free_pinned(void* p);
/* device memory and copies for host code that does not include HIP headers itself (hipMalloc / hipFree on the
 * current device; the copies run on the context's stream and return when done) */
int swz_device_alloc(uint64_t bytes, void** d_out);
int swz_device_alloc_on(swz_ctx* ctx, uint64_t bytes, void** d_out); /* on the context's device (several GPUs) */
int swz_device_free(void* d_ptr);
int swz_copy_to_host(swz_ctx* ctx, void* dst_host, const void* d_src, uint64_t bytes);
int swz_copy_to_device(swz_ctx* ctx, void* d_dst, const void* src_host, uint64_t bytes);

/* ---- synthetic workload of BASELINE.json / SURVEY.md section 8(d): uniform points in the unit
 * cube from a counter-based splitmix64 stream (point i draws x,y,z = draws 3i..3i+2). */
int swz_generate_uniform_device(swz_ctx* ctx, uint64_t seed, uint64_t first_point, uint64_t n,
                                double* d_xyz_out);

/* ---- per-kernel timing (HIP events on the context's stream) for bench.py's roofline.
 * When enabled every kernel class is bracketed by events; totals accumulate until reset. */
typedef struct {
  char name[48];
  uint64_t launches;
  double total_ms;
  uint64_t algorithmic_bytes; /* bytes the launches had to move at minimum */
} swz_kernel_stat;
int swz_profile_enable(swz_ctx* ctx, int enabled);
int swz_profile_reset(swz_ctx* ctx);
/* Copies up to max_stats entries; returns the number of kernel classes via *num_out. */
int swz_profile_get(swz_ctx* ctx, swz_kernel_stat* stats_out, uint32_t max_stats, uint32_t* num_out);

#ifdef __cplusplus
}
#endif
#endif
