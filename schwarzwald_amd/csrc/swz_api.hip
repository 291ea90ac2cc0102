// swz_api.hip -- the C-ABI entry points of include/swz_gpu.h: context, workspace, profiling,
// host-buffer wrappers around the device-resident stages.
#include <algorithm>
#include <cstring>

#include "swz_device.h"
#include "swz_internal.h"

// ------------------------------------------------------------------------------ context plumbing
int swz_ctx::get(const char* name, size_t bytes, void** out) {
  swz::DevBuf& b = bufs[name];
  if (bytes == 0) bytes = 16;
  if (b.cap < bytes) {
    const size_t old_cap = b.cap;
    if (b.ptr) {
      SWZ_HIP(this, hipStreamSynchronize(stream));
      free_buf(b);
    }
    // a little head-room so slowly growing requests do not reallocate every call
    size_t want = bytes + bytes / 16;
    // The multi-batch tiler's buffers grow with every batch of a data set (the store, and everything sized by what a
    // batch pulls out of it): they double instead, or a data set of 100 batches frees and allocates GB-sized blocks a
    // thousand times -- 5.4 of the 5.9 s of a cold run of 30 batches were spent in hipFree / hipMalloc, not in kernels.
    // Every other buffer that has to grow takes up to twice its old size as well, but at most 1 GiB beyond the request (the
    // sampler scratch of a data set's batches grows by a per cent per batch; the 16-GB arrays of a 1 B batch must not
    // double because the next batch is a little larger).
    const size_t minimal = (want + 255) & ~size_t(255);
    if (old_cap) {
      if (strncmp(name, "tl_", 3) == 0 || strncmp(name, "tiler_store", 11) == 0) want = std::max(want, 2 * old_cap);
      else want = std::max(want, std::min(2 * old_cap, bytes + (size_t(1) << 30)));
    }
    want = (want + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&b.ptr, want);
    // SWZ_FAIL_ALLOC=<name>: treat every first attempt to allocate that buffer as out of memory (tests of the path below)
    // ("md_*": any buffer whose name starts with what stands in front of the star)
    if (const char* fa = opt("SWZ_FAIL_ALLOC")) {
      const size_t fl = strlen(fa);
      const bool hit = fl && fa[fl - 1] == '*' ? strncmp(fa, name, fl - 1) == 0 : strcmp(fa, name) == 0;
      if (e == hipSuccess && hit) {
        (void)hipFree(b.ptr);
        b.ptr = nullptr;
        e = hipErrorOutOfMemory;
      }
    }
    if (e == hipErrorOutOfMemory) {  // give back the level scratch nobody has asked for since an earlier level
      (void)hipGetLastError();
      SWZ_HIP(this, hipStreamSynchronize(stream));
      for (auto& kv : bufs) {
        const std::string& nm = kv.first;
        const bool level_scratch = nm.compare(0, 3, "md_") == 0 || nm.compare(0, 3, "sp_") == 0 || nm.compare(0, 3, "pm_") == 0;
        // ("..._sr": the root level of a sharded batch -- the other shards read those arrays through peer access / IPC
        // mappings until the batch ends, long after this shard has gone on to its own levels: never evicted here)
        const bool shared_root = md_shard_root_published && nm.size() >= 3 && nm.compare(nm.size() - 3, 3, "_sr") == 0;
        const bool tiler_scratch = tiler_scratch_dead && nm.compare(0, 3, "tl_") == 0;
        if ((level_scratch || tiler_scratch) && !shared_root && kv.second.ptr && kv.second.epoch < scratch_epoch && &kv.second != &b)
          free_buf(kv.second);
      }
      e = hipMalloc(&b.ptr, want);
      if (e == hipErrorOutOfMemory && want > minimal) {  // without the growth head-room, then
        (void)hipGetLastError();
        want = minimal;
        e = hipMalloc(&b.ptr, want);
      }
    }
    if (e != hipSuccess) {
      b.ptr = nullptr;
      // the largest buffers the workspace holds, for whoever has to find the memory
      std::vector<std::pair<size_t, std::string>> held;
      size_t total = 0;
      for (const auto& kv : bufs) {
        if (kv.second.host) continue;
        total += kv.second.cap;
        if (kv.second.cap) held.emplace_back(kv.second.cap, kv.first);
      }
      std::sort(held.rbegin(), held.rend());
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      std::string top;
      for (size_t i = 0; i < held.size() && i < 14; ++i)
        top += " " + held[i].second + "=" + std::to_string(held[i].first >> 20) + "M";
      return fail(SWZ_ERR_HIP, std::string("hipMalloc(") + name + ", " + std::to_string(want) + " bytes): " + hipGetErrorString(e) +
                                 "; workspace holds " + std::to_string(total >> 20) + " MiB, device free " +
                                 std::to_string(free_b >> 20) + " of " + std::to_string(total_b >> 20) + " MiB; largest:" + top);
    }
    b.cap = want;
    // SWZ_POISON=<byte>: fill new workspace memory (hipMalloc does not): shakes out reads of never-written memory
    if (const char* e = opt("SWZ_POISON")) {
      const char* only = opt("SWZ_POISON_ONLY");
      if (!only || strstr(name, only)) SWZ_HIP(this, hipMemsetAsync(b.ptr, atoi(e), want, stream));
    }
  }
  b.epoch = scratch_epoch;
  *out = b.ptr;
  return SWZ_OK;
}

void swz_ctx::release_all() {
  if (stream) (void)hipStreamSynchronize(stream);
  for (auto& kv : bufs) free_buf(kv.second);
  bufs.clear();
}

void swz_ctx::free_buf(swz::DevBuf& b) {
  if (b.ptr && b.ptr == sbi_clean_ptr) sbi_clean_ptr = nullptr;  // (what comes back at this address later is not known to be zero)
  if (b.ptr) (void)(b.host ? hipHostFree(b.ptr) : hipFree(b.ptr));
  b.ptr = nullptr;
  b.cap = 0;
  b.host = false;
}

uint64_t swz_ctx::held_bytes() const {
  uint64_t s = 0;
  for (const auto& kv : bufs)
    if (!kv.second.host) s += kv.second.cap;
  return s;
}

uint64_t swz_ctx::held_host_bytes() const {
  uint64_t s = 0;
  for (const auto& kv : bufs)
    if (kv.second.host) s += kv.second.cap;
  return s;
}

hipEvent_t swz_ctx::take_event() {
  if (!event_pool.empty()) {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void swz_ctx::prof_begin(const char*) {
  if (!profile) return;
  cur_e0_ = take_event();
  (void)hipEventRecord(cur_e0_, stream);
}

void swz_ctx::prof_end(const char* name, uint64_t launches, uint64_t bytes) {
  if (!profile || !cur_e0_) return;
  hipEvent_t e1 = take_event();
  (void)hipEventRecord(e1, stream);
  pending.push_back({name, cur_e0_, e1, launches, bytes});
  cur_e0_ = nullptr;
}

void swz_ctx::prof_collect() {
  for (auto& p : pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      swz::KernelStat& k = kstats[p.name];
      k.launches += p.launches;
      k.total_ms += ms;
      k.bytes += p.bytes;
    }
    event_pool.push_back(p.e0);
    event_pool.push_back(p.e1);
  }
  pending.clear();
}

static std::string g_create_error;
static int check_params(swz_ctx* c, const swz_tile_params* p);

namespace {

// Host-buffer helpers ---------------------------------------------------------------------------
int upload(swz_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return SWZ_OK;
  SWZ_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return SWZ_OK;
}
int download(swz_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return SWZ_OK;
  SWZ_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return SWZ_OK;
}
int sync(swz_ctx* c) {
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  c->prof_collect();
  return SWZ_OK;
}
int check_n(swz_ctx* c, uint64_t n) {
  // 32-bit point indices, and one-thread-per-point launches rounded up to whole workgroups must stay below the
  // 2^32 work-items of a dispatch
  if (n > 0xFFFF0000ull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points in one batch");
  return SWZ_OK;
}
int check_bounds(swz_ctx* c, const double mn[3], const double mx[3]) {
  if (!mn || !mx) return c->fail(SWZ_ERR_BAD_ARG, "bounds must not be NULL");
  for (int a = 0; a < 3; ++a)
    if (!(mx[a] > mn[a])) return c->fail(SWZ_ERR_BAD_ARG, "bounds must have positive extent on every axis");
  return SWZ_OK;
}

}  // namespace

extern "C" {

int swz_abi_version(void) { return SWZ_ABI_VERSION; }

int swz_create(swz_ctx** ctx_out, int device) {
  if (!ctx_out) return SWZ_ERR_BAD_ARG;
  *ctx_out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("swz_create: no usable HIP device (") +
                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                     "); this library has no CPU fallback";
    return SWZ_ERR_HIP;
  }
  if (device < 0 || device >= count) {
    g_create_error = "swz_create: device ordinal out of range";
    return SWZ_ERR_BAD_ARG;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    g_create_error = std::string("swz_create: hipSetDevice: ") + hipGetErrorString(e);
    return SWZ_ERR_HIP;
  }
  swz_ctx* c = new swz_ctx();
  c->device = device;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    g_create_error = std::string("swz_create: hipStreamCreate: ") + hipGetErrorString(e);
    delete c;
    return SWZ_ERR_HIP;
  }
  c->stream = c->own_stream;
  extern char** environ;
  for (char** e = environ; e && *e; ++e) {  // the only look at the environment: SWZ_* switches, once per context
    if (std::strncmp(*e, "SWZ_", 4) != 0) continue;
    const char* eq = std::strchr(*e, '=');
    if (eq) c->options[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
  }
  *ctx_out = c;
  return SWZ_OK;
}

int swz_set_option(swz_ctx* c, const char* name, const char* value) {
  if (!c || !name) return SWZ_ERR_BAD_ARG;
  if (value) c->options[name] = value; else c->options.erase(name);
  return SWZ_OK;
}

int swz_destroy(swz_ctx* c) {
  if (!c) return SWZ_OK;
  (void)hipSetDevice(c->device);
  if (c->nodes_tiler) {  // an swz_tile_nodes_begin_device that was never closed
    (void)swz_tiler_destroy(c->nodes_tiler);
    c->nodes_tiler = nullptr;
  }
  swz::shard_free(c);
  c->release_all();
  c->prof_collect();
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  for (hipStream_t st : c->aux_streams) (void)hipStreamDestroy(st);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return SWZ_OK;
}

const char* swz_last_error(const swz_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int swz_set_stream(swz_ctx* c, void* hip_stream) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  // hipStream_t 0 is a real stream (the device's default stream, which is what torch uses unless told otherwise)
  c->stream = hip_stream == SWZ_OWN_STREAM ? c->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
  return SWZ_OK;
}

int swz_release_workspace(swz_ctx* c) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (c->tiler_active) return c->fail(SWZ_ERR_BAD_ARG, "swz_release_workspace: a tiler of this context keeps its node store in the workspace");
  swz::shard_free(c);  // an open or presorted sharded batch points into the workspace that goes away
  c->release_all();
  return SWZ_OK;
}

uint64_t swz_workspace_bytes(const swz_ctx* c) { return c ? c->held_bytes() : 0; }

int swz_profile_enable(swz_ctx* c, int enabled) {
  if (!c) return SWZ_ERR_BAD_ARG;
  c->profile = enabled != 0;
  return SWZ_OK;
}
int swz_profile_reset(swz_ctx* c) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_TRY(sync(c));
  c->kstats.clear();
  return SWZ_OK;
}
int swz_profile_get(swz_ctx* c, swz_kernel_stat* out, uint32_t max_stats, uint32_t* num_out) {
  if (!c || !num_out) return SWZ_ERR_BAD_ARG;
  SWZ_TRY(sync(c));
  uint32_t k = 0;
  for (const auto& kv : c->kstats) {
    if (k < max_stats && out) {
      std::memset(&out[k], 0, sizeof(out[k]));
      std::strncpy(out[k].name, kv.first.c_str(), sizeof(out[k].name) - 1);
      out[k].launches = kv.second.launches;
      out[k].total_ms = kv.second.total_ms;
      out[k].algorithmic_bytes = kv.second.bytes;
    }
    ++k;
  }
  *num_out = k;
  return SWZ_OK;
}

// ------------------------------------------------------------------------------ encode
int swz_morton_encode_device(swz_ctx* c, double* d_xyz, uint64_t n, const double bmin[3], const double bmax[3],
                             uint64_t* d_keys_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  if (n && (!d_xyz || !d_keys_out)) return c->fail(SWZ_ERR_BAD_ARG, "swz_morton_encode_device: NULL buffer");
  SWZ_TRY(swz::encode_device(c, d_xyz, (uint32_t)n, bmin, bmax, d_keys_out));
  return sync(c);
}

int swz_morton_encode(swz_ctx* c, double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
                      uint64_t* keys_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  if (n == 0) return SWZ_OK;
  if (!xyz || !keys_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_morton_encode: NULL buffer");
  double* d_xyz = nullptr;
  uint64_t* d_keys = nullptr;
  SWZ_TRY(c->get("h_xyz", (size_t)n * 3, &d_xyz));
  SWZ_TRY(c->get("h_keys", (size_t)n, &d_keys));
  SWZ_TRY(upload(c, d_xyz, xyz, (size_t)n * 24));
  SWZ_TRY(swz::encode_device(c, d_xyz, (uint32_t)n, bmin, bmax, d_keys));
  SWZ_TRY(download(c, keys_out, d_keys, (size_t)n * 8));
  SWZ_TRY(download(c, xyz, d_xyz, (size_t)n * 24));  // clamped in place, like index_point
  return sync(c);
}

// ------------------------------------------------------------------------------ sort
int swz_sort_by_key_device(swz_ctx* c, const uint64_t* d_keys, uint64_t n, uint32_t* d_perm_out,
                           uint64_t* d_keys_sorted_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  if (n == 0) return SWZ_OK;
  if (!d_keys || !d_perm_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_sort_by_key_device: NULL buffer");
  uint64_t *kin = nullptr, *kout = d_keys_sorted_out;
  uint32_t* vtmp = nullptr;
  SWZ_TRY(c->get("sort_keys_b", (size_t)n, &kin));
  SWZ_TRY(c->get("sort_vals_b", (size_t)n, &vtmp));
  if (!kout) SWZ_TRY(c->get("sort_keys_a", (size_t)n, &kout));
  if (swz::radix_result_in_second()) {
    SWZ_HIP(c, hipMemcpyAsync(kin, d_keys, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
    SWZ_TRY(swz::radix_sort_pairs(c, kin, vtmp, kout, d_perm_out, (uint32_t)n, true));
  } else {
    SWZ_HIP(c, hipMemcpyAsync(kout, d_keys, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
    SWZ_TRY(swz::radix_sort_pairs(c, kout, d_perm_out, kin, vtmp, (uint32_t)n, true));
  }
  return sync(c);
}

int swz_sort_by_key(swz_ctx* c, const uint64_t* keys, uint64_t n, uint32_t* perm_out, uint64_t* keys_sorted_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  if (n == 0) return SWZ_OK;
  if (!keys || !perm_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_sort_by_key: NULL buffer");
  uint64_t *ka = nullptr, *kb = nullptr;
  uint32_t *va = nullptr, *vb = nullptr;
  SWZ_TRY(c->get("sort_keys_a", (size_t)n, &ka));
  SWZ_TRY(c->get("sort_keys_b", (size_t)n, &kb));
  SWZ_TRY(c->get("sort_vals_a", (size_t)n, &va));
  SWZ_TRY(c->get("sort_vals_b", (size_t)n, &vb));
  if (swz::radix_result_in_second()) {
    SWZ_TRY(upload(c, kb, keys, (size_t)n * 8));
    SWZ_TRY(swz::radix_sort_pairs(c, kb, vb, ka, va, (uint32_t)n, true));
  } else {
    SWZ_TRY(upload(c, ka, keys, (size_t)n * 8));
    SWZ_TRY(swz::radix_sort_pairs(c, ka, va, kb, vb, (uint32_t)n, true));
  }
  SWZ_TRY(download(c, perm_out, va, (size_t)n * 4));
  if (keys_sorted_out) SWZ_TRY(download(c, keys_sorted_out, ka, (size_t)n * 8));
  return sync(c);
}

// ------------------------------------------------------------------------------ synthetic input
int swz_generate_uniform_device(swz_ctx* c, uint64_t seed, uint64_t first_point, uint64_t n, double* d_xyz_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (n && !d_xyz_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_generate_uniform_device: NULL buffer");
  SWZ_TRY(swz::generate_uniform_device(c, seed, first_point, n, d_xyz_out));
  return sync(c);
}

// ------------------------------------------------------------------------------ tile
static int check_params(swz_ctx* c, const swz_tile_params* p) {
  if (!p) return c->fail(SWZ_ERR_BAD_ARG, "params must not be NULL");
  if (p->sampler < SWZ_RANDOM_GRID || p->sampler > SWZ_JITTERED) return c->fail(SWZ_ERR_BAD_ARG, "unknown sampler");
  if (p->strategy != SWZ_ACCURATE && p->strategy != SWZ_FAST) return c->fail(SWZ_ERR_BAD_ARG, "unknown strategy");
  if (!(p->spacing_at_root > 0.f)) return c->fail(SWZ_ERR_BAD_ARG, "spacing_at_root must be > 0");
  if (p->strategy == SWZ_FAST && p->fast_concurrency == 0)
    return c->fail(SWZ_ERR_BAD_ARG, "FAST needs fast_concurrency >= 1");
  return SWZ_OK;
}

int swz_tile_device(swz_ctx* c, double* d_xyz, uint64_t n, const double bmin[3], const double bmax[3],
                    const swz_tile_params* params, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out,
                    uint32_t* d_dup_mask_out, swz_tile_stats* stats) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (c->opt("SWZ_SYNC_ENTRY")) SWZ_HIP(c, hipDeviceSynchronize());
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  SWZ_TRY(check_params(c, params));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (stats) {
    stats->max_level = -1;
    stats->fast_start_levels = -1;
  }
  if (n == 0) return SWZ_OK;
  if (!d_xyz || !d_keys_out || !d_perm_out || !d_level_out)
    return c->fail(SWZ_ERR_BAD_ARG, "swz_tile_device: NULL buffer");
  swz::TileDeviceOut out{d_keys_out, d_perm_out, d_level_out, d_dup_mask_out};
  int st = swz::tile_device(c, d_xyz, (uint32_t)n, bmin, bmax, *params, out, stats);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}

int swz_tile(swz_ctx* c, double* xyz, uint64_t n, const double bmin[3], const double bmax[3],
             const swz_tile_params* params, uint64_t* keys_out, uint32_t* perm_out, int8_t* level_out,
             uint32_t* dup_mask_out, swz_tile_stats* stats) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  SWZ_TRY(check_params(c, params));
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->max_level = -1;
    stats->fast_start_levels = -1;
  }
  if (n == 0) return SWZ_OK;
  if (!xyz || !keys_out || !perm_out || !level_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_tile: NULL buffer");
  double* d_xyz = nullptr;
  uint64_t* d_keys = nullptr;
  uint32_t *d_perm = nullptr, *d_dup = nullptr;
  int8_t* d_level = nullptr;
  SWZ_TRY(c->get("h_xyz", (size_t)n * 3, &d_xyz));
  SWZ_TRY(c->get("h_keys", (size_t)n, &d_keys));
  SWZ_TRY(c->get("h_perm", (size_t)n, &d_perm));
  SWZ_TRY(c->get("h_level", (size_t)n, &d_level));
  if (dup_mask_out) SWZ_TRY(c->get("h_dup", (size_t)n, &d_dup));
  SWZ_TRY(upload(c, d_xyz, xyz, (size_t)n * 24));
  swz::TileDeviceOut out{d_keys, d_perm, d_level, d_dup};
  int st = swz::tile_device(c, d_xyz, (uint32_t)n, bmin, bmax, *params, out, stats);
  if (st != SWZ_OK) {
    (void)sync(c);
    return st;
  }
  SWZ_TRY(download(c, keys_out, d_keys, (size_t)n * 8));
  SWZ_TRY(download(c, perm_out, d_perm, (size_t)n * 4));
  SWZ_TRY(download(c, level_out, d_level, (size_t)n));
  if (dup_mask_out) SWZ_TRY(download(c, dup_mask_out, d_dup, (size_t)n * 4));
  SWZ_TRY(download(c, xyz, d_xyz, (size_t)n * 24));
  return sync(c);
}

// ------------------------------------------------------------------------------ sharded batches
int swz_partition_by_octant_device(swz_ctx* c, const uint64_t* d_keys, uint64_t n, uint32_t* d_perm_out,
                                   uint64_t counts_out[8]) {
  if (!c || !counts_out) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  for (int o = 0; o < 8; ++o) counts_out[o] = 0;
  if (n == 0) return SWZ_OK;
  if (!d_keys || !d_perm_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_partition_by_octant_device: NULL buffer");
  SWZ_TRY(swz::partition_top_digit(c, d_keys, (uint32_t)n, d_perm_out, counts_out));
  return sync(c);
}

int swz_shard_presort_device(swz_ctx* c, const double* d_xyz_local, uint64_t n, const double bmin[3],
                             const double bmax[3], const swz_tile_params* params, uint64_t ghost_capacity) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n + ghost_capacity));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  SWZ_TRY(check_params(c, params));
  if (n && !d_xyz_local) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_presort_device: NULL buffer");
  if (n == 0) return SWZ_OK;  // a shard without points (its octants are empty) has nothing to prepare
  int st = swz::shard_presort_device(c, d_xyz_local, (uint32_t)n, bmin, bmax, *params, (uint32_t)ghost_capacity);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}

int swz_shard_begin_device(swz_ctx* c, const double* d_xyz_local, uint64_t n, const double bmin[3],
                           const double bmax[3], const swz_tile_params* params, const swz_shard_info* shard,
                           uint64_t* num_root_taken_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (!shard) return c->fail(SWZ_ERR_BAD_ARG, "shard info must not be NULL");
  SWZ_TRY(check_n(c, n + shard->num_ghosts));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  SWZ_TRY(check_params(c, params));
  if ((n && !d_xyz_local) || (shard->num_ghosts && !shard->d_ghost_xyz))
    return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_begin_device: NULL buffer");
  if (num_root_taken_out) *num_root_taken_out = 0;
  if (n == 0) {
    // the octants this shard owns are empty (flat terrain in a cubic root box leaves the upper octants without
    // points): nothing of the root is decided here whatever the ghosts are; the batch stays "open" so that
    // swz_shard_finish_device pairs up and reports zero points
    return swz::shard_begin_empty(c);
  }
  int st = swz::shard_begin_device(c, d_xyz_local, (uint32_t)n, bmin, bmax, *params, shard->global_points,
                                   shard->d_ghost_xyz, (uint32_t)shard->num_ghosts, num_root_taken_out);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}

int swz_shard_root_taken_device(swz_ctx* c, double* d_xyz_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(swz::shard_root_taken_device(c, d_xyz_out));
  return sync(c);
}

int swz_shard_finish_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out,
                            swz_tile_stats* stats) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->max_level = -1;
    stats->fast_start_levels = -1;
  }
  int st = swz::shard_finish_device(c, d_keys_out, d_perm_out, d_level_out, stats);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}

// ---- FAST (TilingAlgorithmV3) on a sharded batch
int swz_shard_fast_begin_device(swz_ctx* c, const double* d_xyz_local, uint64_t n, const double bmin[3], const double bmax[3],
                                const swz_tile_params* params, uint32_t* prefix_counts_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_bounds(c, bmin, bmax));
  SWZ_TRY(check_params(c, params));
  if ((n && !d_xyz_local) || !prefix_counts_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_begin_device: NULL buffer");
  int st = swz::shard_fast_begin_device(c, d_xyz_local, (uint32_t)n, bmin, bmax, *params, prefix_counts_out);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}
int swz_shard_fast_run(swz_ctx* c, int32_t start_level, uint64_t* num_root_candidates_out) {
  if (!c || !num_root_candidates_out) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  int st = swz::shard_fast_run_device(c, start_level, num_root_candidates_out);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}
int swz_shard_fast_root_candidates_device(swz_ctx* c, uint64_t* d_keys_out, double* d_xyz_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(swz::shard_fast_root_candidates_device(c, d_keys_out, d_xyz_out));
  return sync(c);
}
int swz_shard_fast_set_root_device(swz_ctx* c, const uint8_t* d_taken) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(swz::shard_fast_set_root_device(c, d_taken));
  return sync(c);
}
int swz_shard_fast_finish_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out, uint32_t* d_dup_out,
                                 swz_tile_stats* stats) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->max_level = -1;
    stats->fast_start_levels = -1;
  }
  int st = swz::shard_fast_finish_device(c, d_keys_out, d_perm_out, d_level_out, d_dup_out, stats);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}

// ------------------------------------------------------------------------------ sample_points
int swz_sample_points(swz_ctx* c, int sampler, uint64_t max_points_per_node, const uint64_t* keys,
                      const uint32_t* idx, uint64_t n, const double* xyz, uint64_t num_points, uint64_t node_key,
                      int32_t node_level, const double root_min[3], const double root_max[3], float spacing_at_root,
                      int behaviour, uint8_t* taken_out, uint64_t* num_taken_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_n(c, num_points));
  SWZ_TRY(check_bounds(c, root_min, root_max));
  if (sampler < SWZ_RANDOM_GRID || sampler > SWZ_JITTERED) return c->fail(SWZ_ERR_BAD_ARG, "unknown sampler");
  if (node_level < -1 || node_level > 20) return c->fail(SWZ_ERR_BAD_ARG, "node_level out of range");
  if (num_taken_out) *num_taken_out = 0;
  if (n == 0) return SWZ_OK;
  if (!keys || !idx || !xyz || !taken_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_sample_points: NULL buffer");
  double* d_xyz = nullptr;
  uint64_t* d_keys = nullptr;
  uint32_t* d_idx = nullptr;
  uint8_t* d_taken = nullptr;
  SWZ_TRY(c->get("h_xyz", (size_t)num_points * 3, &d_xyz));
  SWZ_TRY(c->get("h_keys", (size_t)n, &d_keys));
  SWZ_TRY(c->get("h_perm", (size_t)n, &d_idx));
  SWZ_TRY(c->get("h_taken", (size_t)n, &d_taken));
  SWZ_TRY(upload(c, d_xyz, xyz, (size_t)num_points * 24));
  SWZ_TRY(upload(c, d_keys, keys, (size_t)n * 8));
  SWZ_TRY(upload(c, d_idx, idx, (size_t)n * 4));
  int st = swz::sample_points_device(c, sampler, max_points_per_node, d_keys, d_idx, (uint32_t)n, d_xyz, node_key,
                                     node_level, root_min, root_max, spacing_at_root, behaviour, d_taken,
                                     num_taken_out);
  if (st != SWZ_OK) {
    (void)sync(c);
    return st;
  }
  SWZ_TRY(download(c, taken_out, d_taken, (size_t)n));
  return sync(c);
}

// the same on device buffers (d_xyz: num_points x 3; results stay on the device)
int swz_sample_points_device(swz_ctx* c, int sampler, uint64_t max_points_per_node, const uint64_t* d_keys, const uint32_t* d_idx,
                             uint64_t n, const double* d_xyz, uint64_t num_points, uint64_t node_key, int32_t node_level,
                             const double root_min[3], const double root_max[3], float spacing_at_root, int behaviour,
                             uint8_t* d_taken_out, uint64_t* num_taken_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(check_n(c, n));
  SWZ_TRY(check_n(c, num_points));
  SWZ_TRY(check_bounds(c, root_min, root_max));
  if (sampler < SWZ_RANDOM_GRID || sampler > SWZ_JITTERED) return c->fail(SWZ_ERR_BAD_ARG, "unknown sampler");
  if (node_level < -1 || node_level > 20) return c->fail(SWZ_ERR_BAD_ARG, "node_level out of range");
  if (num_taken_out) *num_taken_out = 0;
  if (n == 0) return SWZ_OK;
  if (!d_keys || !d_idx || !d_xyz || !d_taken_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_sample_points_device: NULL buffer");
  int st = swz::sample_points_device(c, sampler, max_points_per_node, d_keys, d_idx, (uint32_t)n, d_xyz, node_key, node_level, root_min,
                                     root_max, spacing_at_root, behaviour, d_taken_out, num_taken_out);
  int st2 = sync(c);
  return st != SWZ_OK ? st : st2;
}

// ------------------------------------------------------------------------------ node lists
// Pure host bookkeeping on swz_tile's outputs (the adapter calls persist_points per node with it):
// a stable counting sort of the sorted positions by level, then run detection on the key prefix.
int swz_build_node_lists(swz_ctx* c, const uint64_t* keys_sorted, const int8_t* level, uint64_t n,
                         uint32_t* order_out, uint64_t max_nodes, int8_t* node_level_out, uint64_t* node_key_out,
                         uint64_t* node_offset_out, uint64_t* node_count_out, uint64_t* num_nodes_out) {
  if (!c || !num_nodes_out) return SWZ_ERR_BAD_ARG;
  *num_nodes_out = 0;
  if (n == 0) return SWZ_OK;
  if (!keys_sorted || !level || !order_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_build_node_lists: NULL buffer");
  uint64_t hist[23] = {0};
  for (uint64_t i = 0; i < n; ++i) {
    const int l = level[i];
    if (l < -1 || l > 20) return c->fail(SWZ_ERR_BAD_ARG, "level out of range");
    hist[l + 2]++;
  }
  for (int l = 1; l < 23; ++l) hist[l] += hist[l - 1];
  for (uint64_t i = 0; i < n; ++i) order_out[hist[level[i] + 1]++] = (uint32_t)i;
  uint64_t nn = 0;
  uint64_t i = 0;
  while (i < n) {
    const uint64_t p = order_out[i];
    const int l = level[p];
    const uint32_t sh = (l < 0) ? 63u : swz::level_shift(l);
    const uint64_t prefix = keys_sorted[p] >> sh;
    uint64_t j = i + 1;
    while (j < n && level[order_out[j]] == l && (keys_sorted[order_out[j]] >> sh) == prefix) ++j;
    if (nn < max_nodes && node_level_out && node_key_out && node_offset_out && node_count_out) {
      node_level_out[nn] = (int8_t)l;
      node_key_out[nn] = (l < 0) ? 0 : (prefix << sh);
      node_offset_out[nn] = i;
      node_count_out[nn] = j - i;
    }
    ++nn;
    i = j;
  }
  *num_nodes_out = nn;
  if (nn > max_nodes) return c->fail(SWZ_ERR_BAD_ARG, "max_nodes too small");
  return SWZ_OK;
}

}  // extern "C"
