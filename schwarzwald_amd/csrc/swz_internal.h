// swz_internal.h -- context, workspace and launch bookkeeping shared by the HIP translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/swz_gpu.h"

namespace swz {

constexpr uint32_t MAX_LEVELS = 21;  // MortonIndex64Levels, core/datastructures/MortonIndex.h:222-226

struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  uint64_t epoch = 0;  // swz_ctx::scratch_epoch at the last get()
  bool host = false;   // page-locked host memory mapped into the device's address space (a spilled pool of swz_tiler)
};

struct KernelStat {
  uint64_t launches = 0;
  double total_ms = 0.0;
  uint64_t bytes = 0;
};

struct PendingEvent {
  std::string name;
  hipEvent_t e0, e1;
  uint64_t launches;
  uint64_t bytes;
};

}  // namespace swz

struct swz_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  std::map<std::string, swz::DevBuf> bufs;
  // profiling
  bool profile = false;
  std::map<std::string, swz::KernelStat> kstats;
  std::vector<swz::PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  std::vector<hipStream_t> aux_streams;  // side streams for work that may overlap (MIN_DISTANCE node groups), made on demand

  // ---- error helpers
  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    err = std::string(what) + ": " + hipGetErrorString(e) + " (" + file + ":" + std::to_string(line) + ")";
    return SWZ_ERR_HIP;
  }

  // ---- grow-only named workspace.  The MIN_DISTANCE samplers (buffers "md_*", "sp_*", "pm_*") use theirs for one level
  // at a time and say so (next_scratch_epoch() at the start of a level): when the device is out of memory, get() frees
  // such buffers of earlier levels or calls and tries again -- a cloud whose levels need very different per-cell and
  // per-point arrays then needs the largest level's memory, not the sum over the levels.
  uint64_t scratch_epoch = 1;
  void next_scratch_epoch() { ++scratch_epoch; }
  int get(const char* name, size_t bytes, void** out);
  template <typename T>
  int get(const char* name, size_t count, T** out) {
    return get(name, count * sizeof(T), reinterpret_cast<void**>(out));
  }
  void release_all();
  void free_buf(swz::DevBuf& b);   // device or mapped host memory, whichever it is
  uint64_t held_bytes() const;      // device memory only
  uint64_t held_host_bytes() const; // spilled to page-locked host memory

  // ---- timing: bracket [begin, end) launches of one kernel class with events when profiling
  hipEvent_t take_event();
  void prof_begin(const char* name);
  void prof_end(const char* name, uint64_t launches, uint64_t bytes);
  void prof_collect();  // after a stream sync: fold pending events into kstats
  hipEvent_t cur_e0_ = nullptr;
  void* shard = nullptr;  // swz::ShardState of an open sharded batch (swz_level.hip)
  const void* md_shard_root = nullptr;  // swz::MdShardRoot while swz_group runs the MIN_DISTANCE root of a sharded batch on all shards at once
  bool md_shard_root_published = false;  // the "md_*_sr" arrays of this context are mapped by other shards: they stay (see get())
  // swz_mdblock.hip, sb_incremental: [sbi_clean_ptr, + sbi_clean_bytes) of the "sbi_bits" buffer is known to be zero
  const void* sbi_clean_ptr = nullptr;
  size_t sbi_clean_bytes = 0;
  bool tiler_active = false;  // a swz_tiler lives on this context: its node store is part of the workspace
  swz_tiler* nodes_tiler = nullptr;  // the tiler of an open swz_tile_nodes_begin_device / _end_device pair
  // no batch of the tiler is open: its per-batch scratch ("tl_*") holds nothing anybody will read again, and get() may free
  // what was not asked for since the last next_scratch_epoch() when the device runs out of memory (a data set of 3 B points
  // could be tiled but not exported: 100 GB of merge and survivor buffers of the last batch stood in the way)
  bool tiler_scratch_dead = false;
  // Debug / tuning switches ("SWZ_DEBUG", "SWZ_MD_*", ...): read from the environment ONCE, when the context is
  // created, and changed afterwards only through swz_set_option -- no entry point looks at the environment.
  std::map<std::string, std::string> options;
  const char* opt(const char* name) const {
    const auto it = options.find(name);
    return it == options.end() ? nullptr : it->second.c_str();
  }
};

#define SWZ_HIP(ctx, expr)                                                \
  do {                                                                    \
    hipError_t _e = (expr);                                               \
    if (_e != hipSuccess) return (ctx)->hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define SWZ_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != SWZ_OK) return _s; \
  } while (0)

// checks the launch itself (configuration errors); execution errors surface at the next sync
#define SWZ_LAUNCH_CHECK(ctx) SWZ_HIP(ctx, hipGetLastError())

namespace swz {

inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// RAII-less helper: time one kernel class
struct ProfScope {
  swz_ctx* c;
  const char* name;
  uint64_t launches, bytes;
  ProfScope(swz_ctx* ctx, const char* n, uint64_t b, uint64_t l = 1) : c(ctx), name(n), launches(l), bytes(b) {
    c->prof_begin(name);
  }
  ~ProfScope() { c->prof_end(name, launches, bytes); }
};

// SWZ_TRACE=1: synchronise and report after a stage (debugging hangs; never on in production)
#define SWZ_STAGE(ctx, what)                                                              \
  do {                                                                                    \
    if ((ctx)->opt("SWZ_TRACE")) {                                                            \
      const hipError_t _e = hipStreamSynchronize((ctx)->stream);                          \
      fprintf(stderr, "[swz trace] %s:%d %s -> %s\n", __FILE__, __LINE__, what, hipGetErrorString(_e)); \
      fflush(stderr);                                                                     \
    }                                                                                     \
  } while (0)

// hipMemsetAsync in pieces of 1 GiB: a single call of exactly 4 GiB (512 nodes x 2^21 cells x 4 bytes) was
// observed not to fill the buffer on ROCm 7.0 / gfx950 (tools/debug_fullsize.py: neighbour cells went missing)
inline hipError_t memset_large(void* p, int value, size_t bytes, hipStream_t stream) {
  const size_t piece = size_t(1) << 30;
  for (size_t at = 0; at < bytes; at += piece) {
    const hipError_t e = hipMemsetAsync(static_cast<char*>(p) + at, value, bytes - at < piece ? bytes - at : piece, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---- stage entry points (each in its own .hip file) -------------------------------------------
int encode_device(swz_ctx* c, double* d_xyz, uint32_t n, const double bmin[3], const double bmax[3],
                  uint64_t* d_keys);
int generate_uniform_device(swz_ctx* c, uint64_t seed, uint64_t first, uint64_t n, double* d_xyz);

// Stable LSD radix sort of (key, value) pairs.  The keys in the FIRST pair (d_keys_in, and
// d_vals_tmp unless vals_identity) are the input; both pairs are clobbered by the ping-pong and the
// sorted result ends in the first pair for an even number of passes, in the second pair for an odd
// one (radix_result_in_second()).  With vals_identity the first pass synthesises value = element
// index instead of reading the values.
bool radix_result_in_second();
int radix_sort_pairs(swz_ctx* c, uint64_t* d_keys_in, uint32_t* d_vals_tmp, uint64_t* d_keys_out,
                     uint32_t* d_vals_out, uint32_t n, bool vals_identity);

// Device-wide exclusive scan of a u32 array (in place allowed).  *d_total receives the sum.
int scan_exclusive_u32(swz_ctx* c, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint32_t* d_total,
                       const char* tag);

// Gather positions into Morton order, SoA: X[i] = xyz[3*perm[i]] ...
int gather_positions(swz_ctx* c, const double* d_xyz, const uint32_t* d_perm, uint32_t n, double* d_x,
                     double* d_y, double* d_z);

struct TileDeviceOut {
  uint64_t* keys;   // sorted keys (n)
  uint32_t* perm;   // original index (n)
  int8_t* level;    // taken level per sorted position (n)
  uint32_t* dup;    // may be null
};

int tile_device(swz_ctx* c, double* d_xyz, uint32_t n, const double bmin[3], const double bmax[3],
                const swz_tile_params& p, const TileDeviceOut& out, swz_tile_stats* stats);

int shard_begin_device(swz_ctx* c, const double* d_xyz_local, uint32_t n, const double bmin[3],
                       const double bmax[3], const swz_tile_params& p, uint64_t global_points,
                       const double* d_ghost_xyz, uint32_t ghosts, uint64_t* num_root_taken);
int shard_presort_device(swz_ctx* c, const double* d_xyz_local, uint32_t n, const double bmin[3], const double bmax[3],
                         const swz_tile_params& p, uint32_t ghost_capacity);
int shard_root_taken_device(swz_ctx* c, double* d_xyz_out);
int shard_finish_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out,
                        swz_tile_stats* stats);
int shard_begin_empty(swz_ctx* c);
void shard_free(swz_ctx* c);
// FAST on a sharded batch (swz_shard_fast_*)
int shard_fast_begin_device(swz_ctx* c, const double* d_xyz_local, uint32_t n, const double bmin[3], const double bmax[3],
                            const swz_tile_params& p, uint32_t* counts_host);
int shard_fast_run_device(swz_ctx* c, int start_level, uint64_t* num_root_candidates);
int shard_fast_root_candidates_device(swz_ctx* c, uint64_t* d_keys_out, double* d_xyz_out);
int shard_fast_set_root_device(swz_ctx* c, const uint8_t* d_taken);
int shard_fast_finish_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out, uint32_t* d_dup_out,
                             swz_tile_stats* stats);
// One radix pass on the top key digit: perm groups the points by octant (stable); octants (host)
// receives the eight counts.
int partition_top_digit(swz_ctx* c, const uint64_t* d_keys, uint32_t n, uint32_t* d_perm_out, uint64_t octants[8]);
// Stable partition by the top byte of the keys: d_perm_out[i] = index of the i-th element, starts[d] = first
// position of byte value d.
int partition_by_top_byte(swz_ctx* c, const uint64_t* d_keys, uint32_t n, uint32_t* d_perm_out, uint32_t starts[256]);

int sample_points_device(swz_ctx* c, int sampler, uint64_t max_points, const uint64_t* d_keys,
                         const uint32_t* d_idx, uint32_t n, const double* d_xyz, uint64_t node_key,
                         int32_t node_level, const double rmin[3], const double rmax[3], float spacing,
                         int behaviour, uint8_t* d_taken, uint64_t* num_taken);

}  // namespace swz
