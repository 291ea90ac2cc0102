// swz_group.hip -- a batch sharded over several GPUs, driven from ONE host process in C++ (SURVEY.md section 8(e);
// BASELINE north star: "host code stays C++ ... points shard across the GPUs of one node by the top-3 Morton bits
// with a single RCCL all-to-all after the encode step").
//
// Schwarzwald is a single process, so its drop-in for several GPUs is one object that owns one context per shard and
// one host thread per shard: encode -> group the rows by destination -> ONE exchange step -> root node -> levels,
// every step through the same C ABI the single-GPU path uses (swz_morton_encode_device, swz_partition_by_octant_device,
// swz_shard_*).  The exchange is either grouped RCCL send/recv (ncclCommInitAll in this process; librccl is loaded
// on demand so that nothing else depends on it) or peer copies (hipMemcpyPeerAsync), which also work with several
// shards on one GPU -- that is how the shard logic is tested on a one-GPU box.  The Python driver (sharded.py) is the
// multi-process counterpart used by bench.py.
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "swz_internal.h"
#include "swz_level.h"

namespace {

// the few RCCL entry points the exchange needs (rccl.h: ncclCommInitAll :236, ncclCommDestroy :260, ncclSend :700,
// ncclRecv :722, ncclGroupStart :923, ncclGroupEnd :933), resolved with dlsym
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*Send)(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
  int (*Recv)(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static constexpr int kUint8 = 1;  // ncclUint8, rccl.h:460
  bool load(std::string& err) {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      err = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !Send || !Recv || !GroupStart || !GroupEnd) {
      err = "librccl lacks an expected symbol";
      return false;
    }
    return true;
  }
};

struct Barrier {  // all shard threads meet here between the steps
  std::mutex m;
  std::condition_variable cv;
  int count = 0, generation = 0, parties = 1;
  void wait() {
    std::unique_lock<std::mutex> g(m);
    const int gen = generation;
    if (++count == parties) {
      count = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(g, [&] { return gen != generation; });
    }
  }
};

constexpr size_t kChunkBytes = size_t(1) << 30;  // per message: RCCL 2.26 was measured to lose the tail of one > 2 GB block
constexpr uint64_t kGhostHeadroom = 4u << 20;    // rows kept free in front of the received points (MIN_DISTANCE root ghosts)

}  // namespace

struct swz_group {
  int n = 0;
  int transport = 0;
  std::vector<int> devices;
  std::vector<swz_ctx*> ctx;
  std::vector<void*> comms;
  Rccl rccl;
  std::string err;
  Barrier barrier;
  // per batch, shared between the shard threads
  std::vector<std::vector<uint64_t>> send_counts;  // [source][destination]
  std::vector<double*> recv_rows;                  // device buffers of every shard: positions ...
  std::vector<std::vector<char*>> recv_attr;       // ... and attribute columns [shard][attribute]
  std::vector<uint64_t> recv_total;
  std::vector<double*> root_taken;                 // positions the root took on every shard (ghosts of the higher ones)
  std::vector<uint64_t> root_taken_count;
  std::vector<int> status;
  std::vector<std::string> status_msg;
  int turn = 0;  // MIN_DISTANCE root: the shard whose turn it is
  std::mutex turn_m;
  std::condition_variable turn_cv;
  // a data set in several batches: one swz_tiler per shard (swz_group_tiler_open)
  std::vector<swz_tiler*> tiler;
  std::vector<uint64_t> root_stored;  // points in the root's file on every shard (before / after the batch's root step)
  double tbmin[3] = {0, 0, 0}, tbmax[3] = {0, 0, 0};
  swz_tile_params tparams{};
  std::vector<std::array<double, 4>> timing;  // per shard, ms since the call began: exchange done, root begun, root done, levels done
  std::chrono::steady_clock::time_point t_call;
  std::vector<swz::MdPeerView> views;        // MIN_DISTANCE root swept by all shards at once: what every shard publishes
  int fast_start = -1;                       // FAST: the start level, known after the first batch
  bool peer_access = true;                   // kernels of one shard may read the memory of every other (the joint MIN_DISTANCE root does)
  std::vector<std::vector<uint32_t>> hist;   // FAST, first batch: every shard's points per 6-octant prefix
  // FAST in swz_group_tile: what every shard's level-0 nodes hold (the root is reconstructed from all of them on shard 0)
  int fast_tile_start = -1;
  std::vector<uint64_t*> cand_keys;
  std::vector<double*> cand_xyz;
  std::vector<uint64_t> cand_count;
  uint8_t* cand_flags = nullptr;
  // batches staged from pinned host memory: two device buffers per shard, filled on a copy stream of their own
  struct Staged {
    std::vector<uint64_t> n;  // per shard
    int slot = 0;
  };
  std::vector<Staged> staged;  // at most two, oldest first
  std::vector<hipStream_t> copy_stream;
  std::vector<hipEvent_t> copy_done[2];
  std::vector<double*> stage_xyz[2];
  std::vector<swz_attribute_columns> stage_attr[2];
  std::vector<uint64_t> stage_cap[2];
  uint32_t stage_mask = 0;  // attribute columns of the staged batches (the same for every batch)
  int next_slot = 0;
};

namespace {

int owner_of_octant(int octant, int shards) { return octant * shards / 8; }

void group_barrier(void* p) { static_cast<swz_group*>(p)->barrier.wait(); }

// Can the shards sweep the MIN_DISTANCE root together (swz_mdkeys.hip, MdShardRoot)?  The same answer on every shard: it
// hangs on the bounds, the spacing and the options only.
bool joint_root_possible(const swz_ctx* c, const swz_tile_params& p, const double bmin[3], const double bmax[3]) {
  if (p.sampler != SWZ_MIN_DISTANCE) return false;  // (with SWZ_FLAG_MIN_DISTANCE_PROPERTY too: the root is always sampled exactly)
  if (const char* e = c->opt("SWZ_GROUP_JOINT_ROOT"))
    if (atoi(e) == 0) return false;
  const swz::LevelPlan plan = swz::make_plan(-1, p.sampler, p.max_points_per_node, p.spacing_at_root, p.max_depth, bmin, bmax, true, true);
  static const double dummy_xyz = 0.0;
  static const uint32_t dummy_perm = 0;
  swz::SortedPoints sp;
  sp.xyz = &dummy_xyz;
  sp.perm = &dummy_perm;
  return swz::key_metric(c, plan, sp).ok && plan.cell_levels_geo >= 1 && !plan.terminal && !plan.reroot;
}

__global__ __launch_bounds__(256) void grp_iota_kernel(uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = i;
}

struct ShardCall {
  swz_group* g;
  int r;
  double* d_xyz;
  uint64_t n;
  const double* bmin;
  const double* bmax;
  const swz_tile_params* params;
  const swz_attribute_columns* attrs;  // may be NULL
  swz_group_result* result;
  bool batch = false;            // true: one batch of the group's tilers (swz_group_add_batch), result unused
  swz_tile_stats* stats = nullptr;
};

// true when no shard has reported a failure (looked at by every shard thread right after a barrier, so all of them see
// the same answer and leave the collective steps together)
bool all_ok(const swz_group* g) {
  for (int s : g->status)
    if (s != SWZ_OK) return false;
  return true;
}

bool fail(swz_group* g, int r, int code, const std::string& msg) {
  g->status[r] = code;
  g->status_msg[r] = msg;
  return false;
}
#define GRP_TRY(expr)                                                                  \
  do {                                                                                 \
    if (ok) {                                                                          \
      const int _s = (expr);                                                           \
      if (_s != SWZ_OK) ok = fail(g, r, _s, swz_last_error(c));                        \
    }                                                                                  \
  } while (0)
#define GRP_HIP(expr)                                                                  \
  do {                                                                                 \
    if (ok) {                                                                          \
      const hipError_t _e = (expr);                                                    \
      if (_e != hipSuccess) ok = fail(g, r, SWZ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    }                                                                                  \
  } while (0)

// One shard's part of a batch.  A failing shard keeps walking through the barriers (with nothing to contribute) so
// that the others are not stranded; swz_group_tile reports the first failure.
void shard_thread(ShardCall a) {
  swz_group* g = a.g;
  const int r = a.r, N = g->n;
  swz_ctx* c = g->ctx[r];
  bool ok = true;
  (void)hipSetDevice(g->devices[r]);
  const uint64_t n = a.n;

  // 1. encode locally, group the rows by destination shard
  uint64_t* d_keys = nullptr;
  uint32_t* d_perm = nullptr;
  if (ok && c->get("grp_keys", (size_t)n, &d_keys) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  if (ok && c->get("grp_perm", (size_t)n, &d_perm) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  uint64_t oct[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n) {
    GRP_TRY(swz_morton_encode_device(c, a.d_xyz, n, a.bmin, a.bmax, d_keys));
    GRP_TRY(swz_partition_by_octant_device(c, d_keys, n, d_perm, oct));
  }
  std::fill(g->send_counts[r].begin(), g->send_counts[r].end(), 0);
  for (int o = 0; o < 8; ++o) g->send_counts[r][owner_of_octant(o, N)] += ok ? oct[o] : 0;
  double* send = nullptr;
  if (ok && c->get("grp_send", (size_t)n * 3, &send) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  swz_attribute_columns send_attr{};
  uint32_t row_bytes[SWZ_ATTR_COUNT];
  for (int t = 0; t < SWZ_ATTR_COUNT; ++t) {
    row_bytes[t] = (a.attrs && a.attrs->column[t]) ? swz_attribute_row_bytes(t) : 0;
    if (!row_bytes[t]) continue;
    const std::string name = "grp_send_attr" + std::to_string(t);
    if (ok && c->get(name.c_str(), (size_t)n * row_bytes[t], &send_attr.column[t]) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  }
  if (n) GRP_TRY(swz_gather_payload_device(c, d_perm, nullptr, n, a.d_xyz, a.attrs, send, a.attrs ? &send_attr : nullptr));
  g->barrier.wait();  // ---- every shard knows every count

  // 2. the one exchange step
  uint64_t m = 0;
  std::vector<uint64_t> recv_off(N, 0), send_off(N, 0);
  for (int s = 0; s < N; ++s) {
    recv_off[s] = m;
    m += g->send_counts[s][r];
  }
  for (int d = 1; d < N; ++d) send_off[d] = send_off[d - 1] + g->send_counts[r][d - 1];
  uint64_t global_points = 0;
  for (int s = 0; s < N; ++s)
    for (int d = 0; d < N; ++d) global_points += g->send_counts[s][d];
  double* buf = nullptr;
  if (ok && c->get("grp_recv", (size_t)(m + kGhostHeadroom) * 3, &buf) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  double* recv = buf ? buf + kGhostHeadroom * 3 : nullptr;
  g->recv_rows[r] = recv;
  g->recv_total[r] = m;
  swz_attribute_columns recv_attr{};
  for (int t = 0; t < SWZ_ATTR_COUNT; ++t) {
    g->recv_attr[r][t] = nullptr;
    if (!row_bytes[t]) continue;
    const std::string name = "grp_recv_attr" + std::to_string(t);
    if (ok && c->get(name.c_str(), (size_t)std::max<uint64_t>(m, 1) * row_bytes[t], &recv_attr.column[t]) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    g->recv_attr[r][t] = ok ? (char*)recv_attr.column[t] : nullptr;
  }
  // the columns of one row travel as separate blocks: [0] positions, [1 + t] attribute t
  struct Column {
    const char* src;
    char* dst_here;
    uint32_t bytes;
    int attr;
  };
  std::vector<Column> columns{{(const char*)send, (char*)recv, 24u, -1}};
  for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
    if (row_bytes[t]) columns.push_back({(const char*)send_attr.column[t], (char*)recv_attr.column[t], row_bytes[t], t});
  g->barrier.wait();  // ---- every receive buffer exists
  // A shard that has failed so far posts no sends or receives, and its peers would wait for them for ever (RCCL) or
  // copy into buffers that do not exist: every shard looks at every status HERE -- nobody changes one between the
  // barrier above and the one after the exchange -- and they skip the collective steps together.
  const bool go = all_ok(g);
  if (!go) {
    ok = false;  // (this shard's own status stays as it is: the group reports the shard that failed first)
  } else if (g->transport == 1 && N > 1) {
    // grouped RCCL point-to-point = the all-to-all(v): bounded messages, both sides cut their block the same way
    bool started = false;
    if (ok) {
      started = g->rccl.GroupStart() == 0;
      if (!started) ok = fail(g, r, SWZ_ERR_HIP, "ncclGroupStart failed");
    }
    for (const Column& col : columns)
      for (int p = 0; p < N && ok; ++p) {
        if (p == r) continue;
        const size_t sb = (size_t)g->send_counts[r][p] * col.bytes, rb = (size_t)g->send_counts[p][r] * col.bytes;
        for (size_t at = 0; at < sb && ok; at += kChunkBytes)
          if (g->rccl.Send(col.src + send_off[p] * col.bytes + at, std::min(kChunkBytes, sb - at), Rccl::kUint8, p, g->comms[r], c->stream) != 0)
            ok = fail(g, r, SWZ_ERR_HIP, "ncclSend failed");
        for (size_t at = 0; at < rb && ok; at += kChunkBytes)
          if (g->rccl.Recv(col.dst_here + recv_off[p] * col.bytes + at, std::min(kChunkBytes, rb - at), Rccl::kUint8, p, g->comms[r], c->stream) != 0)
            ok = fail(g, r, SWZ_ERR_HIP, "ncclRecv failed");
      }
    if (started && g->rccl.GroupEnd() != 0 && ok) ok = fail(g, r, SWZ_ERR_HIP, "ncclGroupEnd failed");
    for (const Column& col : columns)
      if (ok && g->send_counts[r][r])
        GRP_HIP(hipMemcpyAsync(col.dst_here + recv_off[r] * col.bytes, col.src + send_off[r] * col.bytes,
                               (size_t)g->send_counts[r][r] * col.bytes, hipMemcpyDeviceToDevice, c->stream));
  } else {
    // peer copies: every source pushes its blocks
    for (int d = 0; d < N && ok; ++d) {
      const uint64_t cnt = g->send_counts[r][d];
      if (!cnt || !g->recv_rows[d]) continue;
      uint64_t off = 0;  // where this source's block starts on the destination
      for (int s = 0; s < r; ++s) off += g->send_counts[s][d];
      for (const Column& col : columns) {
        char* dst = col.attr < 0 ? (char*)g->recv_rows[d] : g->recv_attr[d][col.attr];
        if (!dst) continue;
        GRP_HIP(hipMemcpyPeerAsync(dst + off * col.bytes, g->devices[d], col.src + send_off[d] * col.bytes, g->devices[r],
                                   (size_t)cnt * col.bytes, c->stream));
      }
    }
  }
  GRP_HIP(hipStreamSynchronize(c->stream));
  g->barrier.wait();  // ---- all rows have arrived everywhere

  auto stamp = [&](int k) { g->timing[r][k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g->t_call).count(); };
  stamp(0);
  // (the same vote again: a shard whose exchange failed must not keep the others waiting at the root)
  if (!all_ok(g)) ok = false;
  const bool proceed = ok || a.batch;  // a batch keeps every shard's tiler in step with the collective decisions below

  if (!a.batch && a.params->strategy == SWZ_FAST) {
    // 3F. TilingAlgorithmV3: no root step.  The start level from the distribution of the whole batch, the levels from
    // there down and the skipped levels down to 0 on every shard by itself, the root from the level-0 nodes of all shards
    // on shard 0 (swz_shard_fast_*).
    stamp(1);
    g->hist[r].assign(1u << 18, 0u);
    if (ok && global_points < a.params->fast_concurrency) ok = fail(g, r, SWZ_ERR_BAD_ARG, "FAST: a batch needs at least fast_concurrency points");
    GRP_TRY(swz_shard_fast_begin_device(c, recv, m, a.bmin, a.bmax, a.params, g->hist[r].data()));
    g->barrier.wait();
    if (r == 0) {
      std::vector<uint64_t> sum(1u << 18, 0);
      for (int s = 0; s < N; ++s)
        for (uint32_t b = 0; b < (1u << 18); ++b) sum[b] += g->hist[s][b];
      int32_t S = -1;
      const int rc = swz_fast_start_level_from_counts(sum.data(), a.params->fast_concurrency, &S);
      g->fast_tile_start = S;
      if (ok && (rc != SWZ_OK || S < 0))  // (raised before the barrier: all_ok() then stops every shard with the real cause)
        ok = fail(g, 0, rc != SWZ_OK ? rc : SWZ_ERR_INTERNAL, "FAST: no start level from the summed prefix histograms of the batch");
    }
    g->barrier.wait();
    if (!all_ok(g)) ok = false;
    uint64_t ncand = 0;
    GRP_TRY(swz_shard_fast_run(c, g->fast_tile_start, &ncand));
    uint64_t* ck = nullptr;
    double* cx = nullptr;
    if (ok && c->get("grp_cand_keys", (size_t)std::max<uint64_t>(ncand, 1), &ck) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    if (ok && c->get("grp_cand_xyz", (size_t)std::max<uint64_t>(ncand, 1) * 3, &cx) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    if (ncand) GRP_TRY(swz_shard_fast_root_candidates_device(c, ck, cx));
    GRP_HIP(hipStreamSynchronize(c->stream));
    g->cand_keys[r] = ck;
    g->cand_xyz[r] = cx;
    g->cand_count[r] = ok ? ncand : 0;
    stamp(2);
    g->barrier.wait();  // ---- every shard's candidates are in place
    const bool go_root = all_ok(g);
    if (!go_root) ok = false;
    std::vector<uint64_t> off(N + 1, 0);
    for (int s = 0; s < N; ++s) off[s + 1] = off[s] + g->cand_count[s];
    if (r == 0 && go_root) {
      const uint64_t total = off[N];
      g->cand_flags = nullptr;
      if (total > 0xFFFF0000ull) ok = fail(g, r, SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points in the level-0 nodes");
      if (ok && total) {
        uint64_t* keys = nullptr;
        double* all = nullptr;
        uint32_t* iota = nullptr;
        uint8_t* taken = nullptr;
        if (c->get("grp_root_skeys", (size_t)total, &keys) != SWZ_OK || c->get("grp_root_xyz", (size_t)total * 3, &all) != SWZ_OK ||
            c->get("grp_root_perm", (size_t)total, &iota) != SWZ_OK || c->get("grp_root_taken", (size_t)total, &taken) != SWZ_OK)
          ok = fail(g, r, SWZ_ERR_HIP, c->err);
        GRP_HIP(hipStreamSynchronize(c->stream));  // (the peer copies are not ordered with this shard's stream)
        for (int s = 0; s < N && ok; ++s) {
          if (!g->cand_count[s]) continue;
          GRP_HIP(hipMemcpyPeer(keys + off[s], g->devices[0], g->cand_keys[s], g->devices[s], (size_t)g->cand_count[s] * 8));
          GRP_HIP(hipMemcpyPeer(all + off[s] * 3, g->devices[0], g->cand_xyz[s], g->devices[s], (size_t)g->cand_count[s] * 24));
        }
        GRP_HIP(hipDeviceSynchronize());  // (a device-to-device hipMemcpyPeer may return before it is done; this shard's stream does not wait for it)
        if (ok) {
          hipLaunchKernelGGL(grp_iota_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, iota, (uint32_t)total);
          // the candidates of the shards one behind the other are in key order: the shards own ascending octants
          GRP_TRY(swz::sample_points_device(c, a.params->sampler, a.params->max_points_per_node, keys, iota, (uint32_t)total, all, 0, -1, a.bmin,
                                            a.bmax, a.params->spacing_at_root, SWZ_ALWAYS_ADHERE_TO_MIN_SPACING, taken, nullptr));
          GRP_HIP(hipStreamSynchronize(c->stream));
        }
        g->cand_flags = taken;
      }
    }
    g->barrier.wait();  // ---- the root is sampled (or shard 0 has failed)
    if (!all_ok(g)) ok = false;
    if (ok && ncand) {
      uint8_t* mine = nullptr;
      if (r == 0) {
        mine = g->cand_flags;
      } else {
        if (c->get("grp_cand_flags", (size_t)ncand, &mine) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
        GRP_HIP(hipStreamSynchronize(c->stream));
        GRP_HIP(hipMemcpyPeer(mine, g->devices[r], g->cand_flags + off[r], g->devices[0], (size_t)ncand));
        GRP_HIP(hipDeviceSynchronize());
      }
      GRP_TRY(swz_shard_fast_set_root_device(c, mine));
    }
    uint64_t* okeys = nullptr;
    uint32_t *operm = nullptr, *odup = nullptr;
    int8_t* olevel = nullptr;
    if (ok && c->get("grp_out_keys", (size_t)m, &okeys) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    if (ok && c->get("grp_out_perm", (size_t)m, &operm) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    if (ok && c->get("grp_out_level", (size_t)m, &olevel) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    if (ok && c->get("grp_out_dup", (size_t)m, &odup) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    swz_tile_stats stats{};
    GRP_TRY(swz_shard_fast_finish_device(c, okeys, operm, olevel, odup, &stats));
    GRP_HIP(hipStreamSynchronize(c->stream));
    stamp(3);
    if (a.result) {
      a.result->d_xyz = recv;
      a.result->d_keys = okeys;
      a.result->d_perm = operm;
      a.result->d_level = olevel;
      a.result->attrs = recv_attr;
      a.result->num_points = ok ? m : 0;
      a.result->stats = stats;
      a.result->d_dup = odup;
    }
  } else if (!a.batch) {
  // 3. the root node
  stamp(1);
  const bool sequential_root = a.params->sampler == SWZ_MIN_DISTANCE && global_points > a.params->max_points_per_node;
  uint64_t taken = 0;
  g->root_taken_count[r] = 0;
  if (!sequential_root) {
    swz_shard_info info{global_points, nullptr, 0};
    GRP_TRY(swz_shard_begin_device(c, recv, m, a.bmin, a.bmax, a.params, &info, &taken));
  } else if (N > 1 && g->peer_access && joint_root_possible(c, *a.params, a.bmin, a.bmax)) {
    // All shards sweep the root cells of their own octants at the same time; a cell at the face of a lower octant reads
    // that shard's records through peer access (swz_mdkeys.hip).  No ghosts, no turns.
    swz::MdShardRoot sr;
    sr.shard = r;
    sr.shards = N;
    sr.views = g->views.data();
    sr.barrier = group_barrier;
    sr.barrier_arg = g;
    g->views[r] = swz::MdPeerView{};
    c->md_shard_root = &sr;
    c->md_shard_root_published = true;  // the higher shards read this root's "md_*_sr" arrays until the barrier at the end of the batch
    swz_shard_info info{global_points, nullptr, 0};
    if (m) GRP_TRY(swz_shard_begin_device(c, recv, m, a.bmin, a.bmax, a.params, &info, &taken));
    c->md_shard_root = nullptr;
    if (!g->views[r].entered) {  // no points here, or the call failed before its sweep met the others: meet them for it
      g->views[r].ncells = 0;
      g->views[r].status = ok ? SWZ_OK : SWZ_ERR_INTERNAL;
      g->views[r].entered = 1;
      g->barrier.wait();
    }
  } else {
    if (m) GRP_TRY(swz_shard_presort_device(c, recv, m, a.bmin, a.bmax, a.params, kGhostHeadroom));
    {
      std::unique_lock<std::mutex> lk(g->turn_m);
      g->turn_cv.wait(lk, [&] { return g->turn == r; });
    }
    // ghosts: what the root took on all lower shards, right in front of the received points
    uint64_t gh = 0;
    for (int s = 0; s < r; ++s) gh += g->root_taken_count[s];
    if (gh > kGhostHeadroom && ok) ok = fail(g, r, SWZ_ERR_TOO_MANY_POINTS, "more root samples on lower shards than the ghost headroom holds");
    double* gp = recv ? recv - gh * 3 : nullptr;
    uint64_t at = 0;
    for (int s = 0; s < r && ok && m; ++s) {
      if (!g->root_taken_count[s]) continue;
      GRP_HIP(hipMemcpyPeerAsync(gp + at * 3, g->devices[r], g->root_taken[s], g->devices[s], (size_t)g->root_taken_count[s] * 24, c->stream));
      at += g->root_taken_count[s];
    }
    GRP_HIP(hipStreamSynchronize(c->stream));
    swz_shard_info info{global_points, m ? gp : nullptr, m ? gh : 0};
    GRP_TRY(swz_shard_begin_device(c, recv, m, a.bmin, a.bmax, a.params, &info, &taken));
    double* mine = nullptr;
    if (ok && c->get("grp_root_taken", (size_t)std::max<uint64_t>(taken, 1) * 3, &mine) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
    if (taken) GRP_TRY(swz_shard_root_taken_device(c, mine));
    g->root_taken[r] = mine;
    g->root_taken_count[r] = ok ? taken : 0;
    {
      std::lock_guard<std::mutex> lk(g->turn_m);
      g->turn = r + 1;
    }
    g->turn_cv.notify_all();
  }

  // 4. everything below the root is local
  GRP_HIP(hipStreamSynchronize(c->stream));
  stamp(2);
  uint64_t* okeys = nullptr;
  uint32_t* operm = nullptr;
  int8_t* olevel = nullptr;
  if (ok && c->get("grp_out_keys", (size_t)m, &okeys) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  if (ok && c->get("grp_out_perm", (size_t)m, &operm) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  if (ok && c->get("grp_out_level", (size_t)m, &olevel) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
  swz_tile_stats stats{};
  GRP_TRY(swz_shard_finish_device(c, okeys, operm, olevel, &stats));
  GRP_HIP(hipStreamSynchronize(c->stream));
  stamp(3);
  if (a.result) {
    a.result->d_xyz = recv;
    a.result->d_keys = okeys;
    a.result->d_perm = operm;
    a.result->d_level = olevel;
    a.result->attrs = recv_attr;
    a.result->num_points = ok ? m : 0;
    a.result->stats = stats;
    a.result->d_dup = nullptr;
  }
  } else if (proceed) {
    // 3'. one batch of the shard's tiler (swz_tiler_shard_*): the root's take-all / sample decision uses the counts of
    // the WHOLE root (cached points anywhere force sampling, TilingAlgorithms.cpp:272-275); for MIN_DISTANCE the greedy
    // order visits the shards in turn, each with the root files of the lower ones -- as they are after this batch --
    // as ghosts.  4'. the levels below are local.
    swz_tiler* t = g->tiler[r];
    auto tiler_try = [&](int st) {
      if (ok && st != SWZ_OK) ok = fail(g, r, st, swz_last_error(c));
    };
    stamp(1);  // (the batch path stamps like the single-batch one: root step begun / done, levels done -- for FAST, which has
               // no root step per batch, "root" is the indexing of the batch and the vote on the start level)
    if (g->tparams.strategy == SWZ_FAST) {
      // TilingAlgorithmV3: no root step per batch.  The start level comes from the FIRST batch's distribution over all
      // shards (TilingAlgorithms.cpp:1473-1535) and is kept (:1230-1236).
      const bool first = g->fast_start < 0;  // (read before the first barrier below; shard 0 writes it after that barrier)
      swz_tiler_shard_info info{global_points, 0, nullptr, 0};
      uint64_t have = 0;
      if (ok && global_points < g->tparams.fast_concurrency) ok = fail(g, r, SWZ_ERR_BAD_ARG, "FAST: a batch needs at least fast_concurrency points");
      if (ok) tiler_try(swz_tiler_shard_begin_device(t, recv, m, &recv_attr, &info, &have));
      if (first) {
        g->hist[r].assign(1u << 18, 0u);
        if (ok) tiler_try(swz_tiler_shard_fast_histogram(t, g->hist[r].data()));
        g->barrier.wait();
        if (r == 0) {
          std::vector<uint64_t> sum(1u << 18, 0);
          for (int s = 0; s < N; ++s)
            for (uint32_t b = 0; b < (1u << 18); ++b) sum[b] += g->hist[s][b];
          int32_t S = -1;
          (void)swz_fast_start_level_from_counts(sum.data(), g->tparams.fast_concurrency, &S);
          g->fast_start = S;
        }
        g->barrier.wait();
      }
      if (ok) tiler_try(swz_tiler_shard_set_start_level(t, g->fast_start));
      stamp(2);
      swz_tile_stats stats{};
      if (ok) tiler_try(swz_tiler_shard_finish(t, &stats));
      if (a.stats) *a.stats = stats;
      (void)hipStreamSynchronize(c->stream);
      stamp(3);
    } else {
    uint64_t root_before = 0;
    for (int s = 0; s < N; ++s) root_before += g->root_stored[s];
    const bool sample = root_before > 0 || global_points + root_before > g->tparams.max_points_per_node;
    const bool sequential_root = g->tparams.sampler == SWZ_MIN_DISTANCE && sample && global_points > 0;
    swz_tiler_shard_info info{global_points, root_before, nullptr, 0};
    uint64_t have = 0;
    if (!sequential_root) {
      if (ok) tiler_try(swz_tiler_shard_begin_device(t, recv, m, &recv_attr, &info, &have));
    } else if (N > 1 && g->peer_access && joint_root_possible(c, g->tparams, g->tbmin, g->tbmax)) {
      // All shards sweep the root level of their tilers -- the batch's points merged with the shard's part of the root's
      // file -- at the same time; a cell at the face of a lower octant reads that shard's records through peer access, exact
      // positions through its working index (MdPeerView::aidx).  No ghosts, no turns: like swz_group_tile.
      swz::MdShardRoot sr;
      sr.shard = r;
      sr.shards = N;
      sr.views = g->views.data();
      sr.barrier = group_barrier;
      sr.barrier_arg = g;
      g->views[r] = swz::MdPeerView{};
      c->md_shard_root = &sr;
      c->md_shard_root_published = true;
      if (ok) tiler_try(swz_tiler_shard_begin_device(t, recv, m, &recv_attr, &info, &have));
      c->md_shard_root = nullptr;
      if (!g->views[r].entered) {  // nothing to sample here, or the call failed before its sweep met the others: meet them for it
        g->views[r].ncells = 0;
        g->views[r].status = ok ? SWZ_OK : SWZ_ERR_INTERNAL;
        g->views[r].entered = 1;
        g->barrier.wait();
      }
    } else {
      {
        std::unique_lock<std::mutex> lk(g->turn_m);
        g->turn_cv.wait(lk, [&] { return g->turn == r; });
      }
      uint64_t gh = 0;
      for (int s = 0; s < r; ++s) gh += g->root_taken_count[s];
      double* gp = nullptr;
      if (ok && c->get("grp_ghosts", (size_t)std::max<uint64_t>(gh, 1) * 3, &gp) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
      uint64_t at = 0;
      for (int s = 0; s < r && ok; ++s) {
        if (!g->root_taken_count[s]) continue;
        GRP_HIP(hipMemcpyPeerAsync(gp + at * 3, g->devices[r], g->root_taken[s], g->devices[s], (size_t)g->root_taken_count[s] * 24, c->stream));
        at += g->root_taken_count[s];
      }
      GRP_HIP(hipStreamSynchronize(c->stream));
      info.d_ghost_xyz = gh ? gp : nullptr;
      info.num_ghosts = gh;
      if (ok) tiler_try(swz_tiler_shard_begin_device(t, recv, m, &recv_attr, &info, &have));
      double* mine = nullptr;
      if (ok && c->get("grp_root_taken", (size_t)std::max<uint64_t>(have, 1) * 3, &mine) != SWZ_OK) ok = fail(g, r, SWZ_ERR_HIP, c->err);
      if (ok && have) tiler_try(swz_tiler_level_positions_device(t, -1, mine));
      g->root_taken[r] = mine;
      g->root_taken_count[r] = ok ? have : 0;
      {
        std::lock_guard<std::mutex> lk(g->turn_m);
        g->turn = r + 1;
      }
      g->turn_cv.notify_all();
    }
    stamp(2);
    swz_tile_stats stats{};
    if (ok) tiler_try(swz_tiler_shard_finish(t, &stats));
    if (a.stats) *a.stats = stats;
    (void)hipStreamSynchronize(c->stream);
    stamp(3);
    }
  }
  g->barrier.wait();  // ---- nobody reads a neighbour's buffers any more
  c->md_shard_root_published = false;
}

// flag of the i-th sorted element back to the position its element had before the sort
__global__ __launch_bounds__(256) void grp_unsort_flags_kernel(const uint32_t* __restrict__ perm, const uint8_t* __restrict__ taken, uint32_t n,
                                                               uint8_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[perm[i]] = taken[i];
}

}  // namespace

extern "C" {

int swz_group_create(int num_shards, const int* devices, int transport, swz_group** out) {
  if (!out || !devices || num_shards < 1 || (num_shards != 1 && num_shards != 2 && num_shards != 4 && num_shards != 8)) return SWZ_ERR_BAD_ARG;
  *out = nullptr;
  swz_group* g = new swz_group();
  g->n = num_shards;
  g->devices.assign(devices, devices + num_shards);
  g->transport = transport;
  g->barrier.parties = num_shards;
  bool distinct = true;
  for (int i = 0; i < num_shards; ++i)
    for (int j = 0; j < i; ++j) distinct &= devices[i] != devices[j];
  if (transport == 1 && !distinct) {
    delete g;
    return SWZ_ERR_BAD_ARG;  // RCCL refuses two ranks on one GPU; use peer copies (0)
  }
  for (int i = 0; i < num_shards; ++i) {
    swz_ctx* c = nullptr;
    if (swz_create(&c, devices[i]) != SWZ_OK) {
      for (swz_ctx* x : g->ctx) swz_destroy(x);
      delete g;
      return SWZ_ERR_HIP;
    }
    g->ctx.push_back(c);
  }
  if (transport == 1) {
    g->comms.assign(num_shards, nullptr);
    if (!g->rccl.load(g->err) || g->rccl.CommInitAll(g->comms.data(), num_shards, devices) != 0) {
      if (g->err.empty()) g->err = "ncclCommInitAll failed";
      for (swz_ctx* x : g->ctx) swz_destroy(x);
      delete g;
      return SWZ_ERR_HIP;
    }
  }
  // Peer access between all pairs of distinct devices, whatever the transport: peer copies want it, and the MIN_DISTANCE
  // root that all shards sweep at once has cells READ the lower shards' records in place (swz_mdkeys.hip).  Shards that
  // share a device share its memory.  Without it (no xGMI / PCIe peer path) the root is taken in turns instead.
  for (int i = 0; i < num_shards; ++i) {
    (void)hipSetDevice(devices[i]);
    for (int j = 0; j < num_shards; ++j) {
      if (devices[i] == devices[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) {
        g->peer_access = false;
        continue;
      }
      const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) g->peer_access = false;
      (void)hipGetLastError();
    }
  }
  g->send_counts.assign(num_shards, std::vector<uint64_t>(num_shards, 0));
  g->recv_rows.assign(num_shards, nullptr);
  g->recv_attr.assign(num_shards, std::vector<char*>(SWZ_ATTR_COUNT, nullptr));
  g->recv_total.assign(num_shards, 0);
  g->root_taken.assign(num_shards, nullptr);
  g->root_taken_count.assign(num_shards, 0);
  g->views.assign(num_shards, swz::MdPeerView{});
  g->hist.assign(num_shards, std::vector<uint32_t>());
  g->cand_keys.assign(num_shards, nullptr);
  g->cand_xyz.assign(num_shards, nullptr);
  g->cand_count.assign(num_shards, 0);
  g->timing.assign(num_shards, std::array<double, 4>{{0, 0, 0, 0}});
  g->status.assign(num_shards, SWZ_OK);
  g->status_msg.assign(num_shards, "");
  *out = g;
  return SWZ_OK;
}

static void group_free_tilers(swz_group* g) {
  for (size_t r = 0; r < g->tiler.size(); ++r)
    if (g->tiler[r]) (void)swz_tiler_destroy(g->tiler[r]);
  g->tiler.clear();
  for (size_t r = 0; r < g->copy_stream.size(); ++r) {
    (void)hipSetDevice(g->devices[r]);
    if (g->copy_stream[r]) {
      (void)hipStreamSynchronize(g->copy_stream[r]);
      (void)hipStreamDestroy(g->copy_stream[r]);
    }
    for (int k = 0; k < 2; ++k) {
      if (r < g->copy_done[k].size() && g->copy_done[k][r]) (void)hipEventDestroy(g->copy_done[k][r]);
      if (r < g->stage_xyz[k].size() && g->stage_xyz[k][r]) (void)hipFree(g->stage_xyz[k][r]);
      if (r < g->stage_attr[k].size())
        for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
          if (g->stage_attr[k][r].column[t]) (void)hipFree(g->stage_attr[k][r].column[t]);
    }
  }
  g->copy_stream.clear();
  for (int k = 0; k < 2; ++k) {
    g->copy_done[k].clear();
    g->stage_xyz[k].clear();
    g->stage_attr[k].clear();
    g->stage_cap[k].clear();
  }
  g->staged.clear();
  g->stage_mask = 0;
  g->next_slot = 0;
}

int swz_group_destroy(swz_group* g) {
  if (!g) return SWZ_OK;
  group_free_tilers(g);
  for (void* comm : g->comms)
    if (comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(comm);
  for (swz_ctx* c : g->ctx) swz_destroy(c);
  delete g;
  return SWZ_OK;
}

const char* swz_group_last_error(const swz_group* g) { return g ? g->err.c_str() : "swz_group_create failed"; }
int swz_group_num_shards(const swz_group* g) { return g ? g->n : 0; }
int swz_group_shard_timing(const swz_group* g, int shard, double ms_out[4]) {
  if (!g || !ms_out || shard < 0 || shard >= g->n) return SWZ_ERR_BAD_ARG;
  for (int k = 0; k < 4; ++k) ms_out[k] = g->timing[shard][k];
  return SWZ_OK;
}
swz_ctx* swz_group_ctx(swz_group* g, int shard) { return (g && shard >= 0 && shard < g->n) ? g->ctx[shard] : nullptr; }

int swz_group_tile(swz_group* g, double* const* d_xyz, const swz_attribute_columns* d_attrs, const uint64_t* n, const double bmin[3],
                   const double bmax[3], const swz_tile_params* params, swz_group_result* results) {
  if (!g || !d_xyz || !n || !bmin || !bmax || !params) return SWZ_ERR_BAD_ARG;
  if (params->strategy != SWZ_ACCURATE && params->strategy != SWZ_FAST) {
    g->err = "unknown strategy";
    return SWZ_ERR_BAD_ARG;
  }
  if (d_attrs)
    for (int r = 1; r < g->n; ++r)
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
        if ((d_attrs[r].column[t] != nullptr) != (d_attrs[0].column[t] != nullptr)) {
          g->err = "every shard must hand over the same attribute columns";
          return SWZ_ERR_BAD_ARG;
        }
  g->turn = 0;
  g->fast_tile_start = -1;  // (group state shard 0 writes: nothing of an earlier call, failed or not, may be seen by this one)
  g->cand_flags = nullptr;
  g->t_call = std::chrono::steady_clock::now();
  for (auto& t4 : g->timing) std::fill(std::begin(t4), std::end(t4), 0.0);
  std::fill(g->status.begin(), g->status.end(), SWZ_OK);
  std::vector<std::thread> threads;
  for (int r = 0; r < g->n; ++r)
    threads.emplace_back(shard_thread, ShardCall{g, r, d_xyz[r], n[r], bmin, bmax, params, d_attrs ? &d_attrs[r] : nullptr, results ? &results[r] : nullptr});
  for (auto& t : threads) t.join();
  for (int r = 0; r < g->n; ++r)
    if (g->status[r] != SWZ_OK) {
      g->err = "shard " + std::to_string(r) + ": " + g->status_msg[r];
      return g->status[r];
    }
  return SWZ_OK;
}

// ---- a data set in several batches (BASELINE config 5 from the C++ host): one swz_tiler per shard -------------------
static int group_first_failure(swz_group* g) {
  for (int r = 0; r < g->n; ++r)
    if (g->status[r] != SWZ_OK) {
      g->err = "shard " + std::to_string(r) + ": " + g->status_msg[r];
      return g->status[r];
    }
  return SWZ_OK;
}

int swz_group_tiler_open(swz_group* g, const double bmin[3], const double bmax[3], const swz_tile_params* params,
                         uint64_t capacity_hint_per_shard) {
  if (!g || !bmin || !bmax || !params) return SWZ_ERR_BAD_ARG;
  if (!g->tiler.empty()) {
    g->err = "swz_group_tiler_open: a data set is open already (swz_group_tiler_close)";
    return SWZ_ERR_BAD_ARG;
  }
  g->fast_start = -1;
  g->hist.assign(g->n, std::vector<uint32_t>());
  for (int a = 0; a < 3; ++a) {
    g->tbmin[a] = bmin[a];
    g->tbmax[a] = bmax[a];
  }
  g->tparams = *params;
  g->tiler.assign(g->n, nullptr);
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    const int st = swz_tiler_create(g->ctx[r], bmin, bmax, params, capacity_hint_per_shard, &g->tiler[r]);
    if (st != SWZ_OK) {
      g->err = "shard " + std::to_string(r) + ": " + swz_last_error(g->ctx[r]);
      group_free_tilers(g);
      return st;
    }
  }
  g->root_stored.assign(g->n, 0);
  g->copy_stream.assign(g->n, nullptr);
  for (int k = 0; k < 2; ++k) {
    g->copy_done[k].assign(g->n, nullptr);
    g->stage_xyz[k].assign(g->n, nullptr);
    g->stage_attr[k].assign(g->n, swz_attribute_columns{});
    g->stage_cap[k].assign(g->n, 0);
  }
  return SWZ_OK;
}

int swz_group_tiler_close(swz_group* g) {
  if (!g) return SWZ_ERR_BAD_ARG;
  group_free_tilers(g);
  return SWZ_OK;
}

swz_tiler* swz_group_tiler(swz_group* g, int shard) {
  return (g && shard >= 0 && shard < (int)g->tiler.size()) ? g->tiler[shard] : nullptr;
}

int swz_group_add_batch(swz_group* g, double* const* d_xyz, const swz_attribute_columns* d_attrs, const uint64_t* n,
                        swz_tile_stats* stats) {
  if (!g || !d_xyz || !n) return SWZ_ERR_BAD_ARG;
  if (g->tiler.empty()) {
    g->err = "swz_group_add_batch: no data set is open (swz_group_tiler_open)";
    return SWZ_ERR_BAD_ARG;
  }
  if (d_attrs)
    for (int r = 1; r < g->n; ++r)
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
        if ((d_attrs[r].column[t] != nullptr) != (d_attrs[0].column[t] != nullptr)) {
          g->err = "every shard must hand over the same attribute columns";
          return SWZ_ERR_BAD_ARG;
        }
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    const int st = swz_tiler_level_count(g->tiler[r], -1, &g->root_stored[r]);
    if (st != SWZ_OK) {
      g->err = "shard " + std::to_string(r) + ": " + swz_last_error(g->ctx[r]);
      return st;
    }
  }
  g->turn = 0;
  g->t_call = std::chrono::steady_clock::now();  // (swz_group_shard_timing: milliseconds since this batch's call began)
  for (auto& t4 : g->timing) std::fill(std::begin(t4), std::end(t4), 0.0);  // a stamp a path does not reach reads 0, never an older call's
  std::fill(g->status.begin(), g->status.end(), SWZ_OK);
  std::fill(g->root_taken_count.begin(), g->root_taken_count.end(), 0);
  std::vector<std::thread> threads;
  for (int r = 0; r < g->n; ++r) {
    ShardCall call{g, r, d_xyz[r], n[r], g->tbmin, g->tbmax, &g->tparams, d_attrs ? &d_attrs[r] : nullptr, nullptr};
    call.batch = true;
    call.stats = stats ? &stats[r] : nullptr;
    threads.emplace_back(shard_thread, call);
  }
  for (auto& t : threads) t.join();
  const int st = group_first_failure(g);
  if (st != SWZ_OK)  // the shards that did not fail have committed the batch and the failing one has not: nobody goes on
    for (int r = 0; r < g->n; ++r) (void)swz_tiler_poison(g->tiler[r], ("a shard of the group failed to tile a batch: " + g->err).c_str());
  return st;
}

// Staging: the batch is copied from pinned host memory into one of two device buffers per shard on the shard's copy
// stream, beside whatever its tiling stream is doing (the batch before).  At most two batches are staged.
int swz_group_stage_batch(swz_group* g, const double* const* xyz_host, const swz_attribute_columns* attrs_host, const uint64_t* n) {
  if (!g || !xyz_host || !n) return SWZ_ERR_BAD_ARG;
  if (g->tiler.empty() || g->staged.size() >= 2) {
    g->err = g->tiler.empty() ? "swz_group_stage_batch: no data set is open" : "swz_group_stage_batch: two batches are staged already";
    return SWZ_ERR_BAD_ARG;
  }
  uint32_t mask = 0;
  for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
    if (attrs_host && attrs_host[0].column[t]) mask |= 1u << t;
  const int slot = g->next_slot;
  swz_group::Staged sb;
  sb.slot = slot;
  sb.n.assign(n, n + g->n);
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    auto hip = [&](hipError_t e, const char* what) {
      if (e == hipSuccess) return true;
      g->err = "shard " + std::to_string(r) + ": " + what + ": " + hipGetErrorString(e);
      return false;
    };
    if (!g->copy_stream[r] && !hip(hipStreamCreateWithFlags(&g->copy_stream[r], hipStreamNonBlocking), "hipStreamCreate")) return SWZ_ERR_HIP;
    if (!g->copy_done[slot][r] && !hip(hipEventCreateWithFlags(&g->copy_done[slot][r], hipEventDisableTiming), "hipEventCreate")) return SWZ_ERR_HIP;
    uint32_t mine = 0;
    for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
      if (attrs_host && attrs_host[r].column[t]) mine |= 1u << t;
    if (mine != mask) {
      g->err = "every shard must hand over the same attribute columns";
      return SWZ_ERR_BAD_ARG;
    }
    if (g->stage_cap[slot][r] < std::max<uint64_t>(n[r], 1) || (mask & ~g->stage_mask)) {
      // (the buffer is free: the batch it held has been tiled -- swz_group_tile_staged returns after the shard's stream is idle)
      const uint64_t cap = std::max<uint64_t>(n[r] + n[r] / 8, 1);
      if (g->stage_xyz[slot][r]) (void)hipFree(g->stage_xyz[slot][r]);
      g->stage_xyz[slot][r] = nullptr;
      if (!hip(hipMalloc(&g->stage_xyz[slot][r], cap * 24), "hipMalloc(stage)")) return SWZ_ERR_HIP;
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t) {
        if (g->stage_attr[slot][r].column[t]) (void)hipFree(g->stage_attr[slot][r].column[t]);
        g->stage_attr[slot][r].column[t] = nullptr;
        if ((mask >> t) & 1u)
          if (!hip(hipMalloc(&g->stage_attr[slot][r].column[t], cap * swz_attribute_row_bytes(t)), "hipMalloc(stage column)")) return SWZ_ERR_HIP;
      }
      g->stage_cap[slot][r] = cap;
    }
    if (n[r]) {
      if (!hip(hipMemcpyAsync(g->stage_xyz[slot][r], xyz_host[r], n[r] * 24, hipMemcpyHostToDevice, g->copy_stream[r]), "hipMemcpyAsync")) return SWZ_ERR_HIP;
      for (int t = 0; t < SWZ_ATTR_COUNT; ++t)
        if ((mask >> t) & 1u)
          if (!hip(hipMemcpyAsync(g->stage_attr[slot][r].column[t], attrs_host[r].column[t], n[r] * swz_attribute_row_bytes(t),
                                  hipMemcpyHostToDevice, g->copy_stream[r]), "hipMemcpyAsync")) return SWZ_ERR_HIP;
    }
    if (!hip(hipEventRecord(g->copy_done[slot][r], g->copy_stream[r]), "hipEventRecord")) return SWZ_ERR_HIP;
  }
  g->stage_mask |= mask;
  g->staged.push_back(sb);
  g->next_slot ^= 1;
  return SWZ_OK;
}

int swz_group_tile_staged(swz_group* g, swz_tile_stats* stats) {
  if (!g) return SWZ_ERR_BAD_ARG;
  if (g->staged.empty()) {
    g->err = "swz_group_tile_staged: nothing is staged";
    return SWZ_ERR_BAD_ARG;
  }
  const swz_group::Staged sb = g->staged.front();
  g->staged.erase(g->staged.begin());
  std::vector<double*> xyz(g->n);
  std::vector<swz_attribute_columns> attrs(g->n);
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    // the shard's kernels wait for its copy on the device; the host goes on
    const hipError_t e = hipStreamWaitEvent(g->ctx[r]->stream, g->copy_done[sb.slot][r], 0);
    if (e != hipSuccess) {
      g->err = "shard " + std::to_string(r) + ": hipStreamWaitEvent: " + hipGetErrorString(e);
      return SWZ_ERR_HIP;
    }
    xyz[r] = g->stage_xyz[sb.slot][r];
    attrs[r] = g->stage_attr[sb.slot][r];
  }
  return swz_group_add_batch(g, xyz.data(), g->stage_mask ? attrs.data() : nullptr, sb.n.data(), stats);
}

int swz_group_finalize(swz_group* g, swz_tile_stats* stats) {
  if (!g) return SWZ_ERR_BAD_ARG;
  if (g->tiler.empty() || !g->staged.empty()) {
    g->err = g->tiler.empty() ? "swz_group_finalize: no data set is open" : "swz_group_finalize: staged batches have not been tiled";
    return SWZ_ERR_BAD_ARG;
  }
  const bool fast = g->tparams.strategy == SWZ_FAST && g->fast_start > 0;
  auto shard_fail = [&](int r, int rc) {
    g->err = "shard " + std::to_string(r) + ": " + swz_last_error(g->ctx[r]);
    for (int s = 0; s < g->n; ++s) (void)swz_tiler_poison(g->tiler[s], ("the group failed to finalize: " + g->err).c_str());
    return rc;
  };
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    swz_tile_stats st{};
    const int rc = fast ? swz_tiler_shard_fast_finalize_local(g->tiler[r], &st) : swz_tiler_finalize(g->tiler[r], &st);
    if (rc != SWZ_OK) return shard_fail(r, rc);
    if (stats) stats[r] = st;
  }
  if (!fast) return SWZ_OK;
  // The root of TilingAlgorithmV3::finalize (reconstruct_single_node, :1661-1715) samples what its eight children hold,
  // and they lie on different shards: their files come together on shard 0 in shard order (= octant order: the order
  // the reference appends them in), are indexed against the root bounds and sampled with AlwaysAdhereToMinSpacing there;
  // every shard keeps the part of the root's file that comes from its own points.
  std::vector<uint64_t> cnt(g->n, 0), off(g->n + 1, 0);
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    const int rc = swz_tiler_level_count(g->tiler[r], 0, &cnt[r]);
    if (rc != SWZ_OK) return shard_fail(r, rc);
    off[r + 1] = off[r] + cnt[r];
  }
  const uint64_t total = off[g->n];
  swz_ctx* c0 = g->ctx[0];
  uint8_t* d_flags0 = nullptr;
  if (total) {
    if (total > 0xFFFF0000ull) {
      g->err = "more than 2^32-65536 points in the level-0 files";
      return SWZ_ERR_TOO_MANY_POINTS;
    }
    (void)hipSetDevice(g->devices[0]);
    double* all = nullptr;
    uint64_t *keys = nullptr, *skeys = nullptr;
    uint32_t* perm = nullptr;
    uint8_t* taken = nullptr;
    if (c0->get("grp_root_xyz", (size_t)total * 3, &all) != SWZ_OK || c0->get("grp_root_keys", (size_t)total, &keys) != SWZ_OK ||
        c0->get("grp_root_skeys", (size_t)total, &skeys) != SWZ_OK || c0->get("grp_root_perm", (size_t)total, &perm) != SWZ_OK ||
        c0->get("grp_root_taken", (size_t)total, &taken) != SWZ_OK || c0->get("grp_root_flags", (size_t)total, &d_flags0) != SWZ_OK)
      return shard_fail(0, SWZ_ERR_HIP);
    // (the peer copies below run on the default stream: nothing queued on shard 0's own stream -- the SWZ_POISON fill of a
    // buffer that has just been allocated -- may still be writing the destination)
    (void)hipStreamSynchronize(c0->stream);
    for (int r = 0; r < g->n; ++r) {
      if (!cnt[r]) continue;
      (void)hipSetDevice(g->devices[r]);
      swz_ctx* c = g->ctx[r];
      double* mine = nullptr;
      if (c->get("grp_l0_xyz", (size_t)cnt[r] * 3, &mine) != SWZ_OK) return shard_fail(r, SWZ_ERR_HIP);
      const int rc = swz_tiler_level_positions_device(g->tiler[r], 0, mine);
      if (rc != SWZ_OK) return shard_fail(r, rc);
      if (hipStreamSynchronize(c->stream) != hipSuccess ||
          hipMemcpyPeer(all + off[r] * 3, g->devices[0], mine, g->devices[r], (size_t)cnt[r] * 24) != hipSuccess) {
        c->err = "copy of the level-0 file to shard 0 failed";
        return shard_fail(r, SWZ_ERR_HIP);
      }
    }
    (void)hipSetDevice(g->devices[0]);
    (void)hipDeviceSynchronize();  // (a device-to-device hipMemcpyPeer may return before it is done; shard 0's stream does not wait for it)
    int rc = swz_morton_encode_device(c0, all, total, g->tbmin, g->tbmax, keys);
    if (rc == SWZ_OK) rc = swz_sort_by_key_device(c0, keys, total, perm, skeys);
    if (rc == SWZ_OK)
      rc = swz::sample_points_device(c0, g->tparams.sampler, g->tparams.max_points_per_node, skeys, perm, (uint32_t)total, all, 0, -1, g->tbmin,
                                     g->tbmax, g->tparams.spacing_at_root, SWZ_ALWAYS_ADHERE_TO_MIN_SPACING, taken, nullptr);
    if (rc != SWZ_OK) return shard_fail(0, rc);
    hipLaunchKernelGGL(grp_unsort_flags_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c0->stream, perm, taken, (uint32_t)total, d_flags0);
    if (hipStreamSynchronize(c0->stream) != hipSuccess) {
      c0->err = "root reconstruction failed on the device";
      return shard_fail(0, SWZ_ERR_HIP);
    }
  }
  for (int r = 0; r < g->n; ++r) {
    (void)hipSetDevice(g->devices[r]);
    swz_ctx* c = g->ctx[r];
    uint8_t* mine = nullptr;
    if (cnt[r]) {
      if (r == 0) {
        mine = d_flags0;
      } else {
        if (c->get("grp_l0_flags", (size_t)cnt[r], &mine) != SWZ_OK) return shard_fail(r, SWZ_ERR_HIP);
        (void)hipStreamSynchronize(c->stream);  // (as above: the copy is not ordered with this shard's stream)
        if (hipMemcpyPeer(mine, g->devices[r], d_flags0 + off[r], g->devices[0], (size_t)cnt[r]) != hipSuccess) {
          c->err = "copy of the root flags failed";
          return shard_fail(r, SWZ_ERR_HIP);
        }
        (void)hipDeviceSynchronize();
      }
    }
    const int rc = swz_tiler_shard_fast_set_root(g->tiler[r], mine);
    if (rc != SWZ_OK) return shard_fail(r, rc);
  }
  return SWZ_OK;
}


// ------------------------------------------------------------------------- the joint root for one PROCESS per GPU
// swz_group sweeps the MIN_DISTANCE root with all shards at once because its shards share an address space.  A driver
// with one process per GPU (torch.distributed) gets the same through IPC: inside swz_shard_begin_device every rank
// exports the arrays of its root level (hipIpcGetMemHandle on the allocations they live in + offsets), the driver's
// all-gather callback passes the blobs round, and every rank maps the arrays of the LOWER ranks (hipIpcOpenMemHandle).
// From there on the sweep is the one of swz_mdkeys.hip: remote records bracketed by the owner's round word, system-scope
// loads, cells at the faces polling.  swz_shard_joint_root_end -- after the driver's barrier, when every rank has
// finished its root -- unmaps.
}  // extern "C"
namespace {
struct JointBlob {
  hipIpcMemHandle_t handle[9];
  uint64_t offset[9];
  uint8_t has[9];
  uint32_t ncells, rg, cell_shift, npoints;
  int32_t status, entered;
};
struct JointState {
  swz::MdShardRoot sr;
  std::vector<swz::MdPeerView> views;
  swz_exchange_fn exchange = nullptr;
  void* exchange_arg = nullptr;
  std::vector<std::pair<hipIpcMemHandle_t, void*>> opened;  // one mapping per allocation
  std::string err;
  swz_ctx* c = nullptr;
};
std::map<swz_ctx*, JointState*>& joint_states() {
  static std::map<swz_ctx*, JointState*> m;
  return m;
}
std::mutex joint_m;

void* joint_open(JointState* js, const hipIpcMemHandle_t& h) {
  for (auto& kv : js->opened)
    if (std::memcmp(&kv.first, &h, sizeof(h)) == 0) return kv.second;
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  js->opened.emplace_back(h, p);
  return p;
}

// the "barrier" of MdShardRoot: every rank has published its view in views[shard] -- exchange them
void joint_exchange(void* arg) {
  JointState* js = static_cast<JointState*>(arg);
  const int N = js->sr.shards, r = js->sr.shard;
  swz::MdPeerView& mine = js->views[r];
  JointBlob blob{};
  const void* ptrs[9] = {mine.rec, mine.qpos, mine.state, mine.ovf, mine.gridmap, mine.round_word, mine.perm, mine.xyz, mine.aidx};
  blob.ncells = mine.ncells;
  blob.npoints = mine.npoints;
  blob.rg = mine.rg;
  blob.cell_shift = mine.cell_shift;
  blob.status = mine.status;
  blob.entered = 1;
  for (int k = 0; k < 9 && mine.ncells; ++k) {
    if (!ptrs[k]) continue;
    void* base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(ptrs[k])) != hipSuccess ||
        hipIpcGetMemHandle(&blob.handle[k], base) != hipSuccess) {
      (void)hipGetLastError();
      blob.status = SWZ_ERR_HIP;  // (e.g. memory that does not come from hipMalloc)
      continue;
    }
    blob.offset[k] = (uint64_t)((const char*)ptrs[k] - (const char*)base);
    blob.has[k] = 1;
  }
  std::vector<JointBlob> all((size_t)N);
  if (js->exchange(js->exchange_arg, &blob, sizeof(JointBlob), all.data()) != 0) {
    // Nobody sweeps.  The other ranks still run the vote below: take part in it (with an error), or this rank would be one
    // collective behind for everything that follows.
    for (int p = 0; p < N; ++p) js->views[p].status = SWZ_ERR_INTERNAL;
    JointBlob vote{};
    vote.status = SWZ_ERR_INTERNAL;
    std::vector<JointBlob> votes((size_t)N);
    (void)js->exchange(js->exchange_arg, &vote, sizeof(JointBlob), votes.data());
    return;
  }
  for (int p = 0; p < N; ++p) {
    if (p == r) {
      js->views[p].status = blob.status;
      continue;
    }
    swz::MdPeerView v{};
    v.ncells = all[p].ncells;
    v.rg = all[p].rg;
    v.npoints = all[p].npoints;
    v.cell_shift = all[p].cell_shift;
    v.status = all[p].status;
    v.entered = all[p].entered;
    if (p < r && v.ncells && v.status == SWZ_OK) {  // only the lower ranks' arrays are read
      const void* got[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      for (int k = 0; k < 9; ++k) {
        if (!all[p].has[k]) continue;
        char* base = static_cast<char*>(joint_open(js, all[p].handle[k]));
        if (!base) {
          v.status = SWZ_ERR_HIP;
          break;
        }
        got[k] = base + all[p].offset[k];
      }
      v.rec = static_cast<const uint4*>(got[0]);
      v.qpos = static_cast<const uint64_t*>(got[1]);
      v.state = static_cast<const uint8_t*>(got[2]);
      v.ovf = static_cast<const float4*>(got[3]);
      v.gridmap = static_cast<const uint32_t*>(got[4]);
      v.round_word = static_cast<const uint32_t*>(got[5]);
      v.perm = static_cast<const uint32_t*>(got[6]);
      v.xyz = static_cast<const double*>(got[7]);
      v.aidx = static_cast<const uint32_t*>(got[8]);
    }
    js->views[p] = v;
  }
  // (a mapping that failed on ONE rank must stop all of them, or the others sweep against a rank that has given up: the
  // statuses are voted on)
  int32_t my_ok = 1;
  for (int p = 0; p < N; ++p) my_ok &= js->views[p].status == SWZ_OK ? 1 : 0;
  JointBlob vote{};
  vote.status = my_ok ? SWZ_OK : SWZ_ERR_INTERNAL;
  std::vector<JointBlob> votes((size_t)N);
  if (js->exchange(js->exchange_arg, &vote, sizeof(JointBlob), votes.data()) != 0) vote.status = SWZ_ERR_INTERNAL;
  for (int p = 0; p < N; ++p)
    if (votes[p].status != SWZ_OK || vote.status != SWZ_OK) js->views[p].status = SWZ_ERR_INTERNAL;
}
}  // namespace
extern "C" {

int swz_shard_joint_root_possible(swz_ctx* c, const swz_tile_params* p, const double bmin[3], const double bmax[3]) {
  if (!c || !p) return 0;
  return joint_root_possible(c, *p, bmin, bmax) ? 1 : 0;
}

// Can the ranks of this run map each other's device memory at all?  Every rank exports a small buffer of its workspace that
// holds a pattern, the lower ranks' buffers are opened and read back, and the outcome is voted on -- two exchanges, like the
// real thing -- so that a driver can settle for the chain of ghosts BEFORE a batch depends on the mappings (containers
// without a shared /dev/shm, devices without a peer path, IPC switched off).  *usable = 1 only when every rank could map
// and read every lower rank's buffer.
int swz_shard_joint_root_probe(swz_ctx* c, int shard, int shards, swz_exchange_fn exchange, void* arg, int* usable) {
  if (!c || !exchange || !usable || shards < 1 || shards > 8 || shard < 0 || shard >= shards) return SWZ_ERR_BAD_ARG;
  *usable = 0;
  (void)hipSetDevice(c->device);
  struct ProbeBlob {
    hipIpcMemHandle_t handle;
    uint64_t offset;
    int32_t status;
    uint32_t pattern;
  };
  ProbeBlob mine{};
  mine.pattern = 0x5C4A0000u + (uint32_t)shard;
  uint32_t* buf = nullptr;
  mine.status = c->get("grp_ipc_probe", (size_t)64, &buf);
  if (mine.status == SWZ_OK) {
    void* base = nullptr;
    size_t size = 0;
    if (hipMemcpy(buf, &mine.pattern, 4, hipMemcpyHostToDevice) != hipSuccess || hipMemGetAddressRange(&base, &size, buf) != hipSuccess ||
        hipIpcGetMemHandle(&mine.handle, base) != hipSuccess) {
      (void)hipGetLastError();
      mine.status = SWZ_ERR_HIP;
    } else {
      mine.offset = (uint64_t)((const char*)buf - (const char*)base);
    }
  }
  std::vector<ProbeBlob> all((size_t)shards);
  int32_t ok = mine.status == SWZ_OK ? 1 : 0;
  std::vector<void*> opened;
  if (exchange(arg, &mine, sizeof(ProbeBlob), all.data()) != 0) {
    ok = 0;
  } else {
    for (int p = 0; p < shards && ok; ++p) ok &= all[p].status == SWZ_OK ? 1 : 0;
    for (int p = 0; p < shard && ok; ++p) {  // only the lower ranks' arrays are ever read
      void* base = nullptr;
      uint32_t got = 0;
      if (hipIpcOpenMemHandle(&base, all[p].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        ok = 0;
        break;
      }
      opened.push_back(base);
      if (hipMemcpy(&got, (const char*)base + all[p].offset, 4, hipMemcpyDeviceToHost) != hipSuccess || got != all[p].pattern) {
        (void)hipGetLastError();
        ok = 0;
      }
    }
  }
  ProbeBlob vote{};
  vote.status = ok ? SWZ_OK : SWZ_ERR_INTERNAL;
  std::vector<ProbeBlob> votes((size_t)shards);
  int all_ok = ok;
  if (exchange(arg, &vote, sizeof(ProbeBlob), votes.data()) != 0) all_ok = 0;
  for (int p = 0; p < shards && all_ok; ++p) all_ok &= votes[p].status == SWZ_OK ? 1 : 0;
  // Unmap, one rank at a time (the exchange is the barrier between them): with several processes closing mappings of one
  // allocation at the same moment the runtime was seen to abort ("Memobj map does not have ptr", one run in four with
  // four processes on one GPU) -- never with the closes taken in turns.
  for (int p = 0; p < shards; ++p) {
    if (p == shard)
      for (void* b : opened) (void)hipIpcCloseMemHandle(b);
    ProbeBlob turn{};
    std::vector<ProbeBlob> turns((size_t)shards);
    if (exchange(arg, &turn, sizeof(ProbeBlob), turns.data()) != 0) all_ok = 0;
  }
  *usable = all_ok;
  return SWZ_OK;
}

int swz_shard_joint_root_begin(swz_ctx* c, int shard, int shards, swz_exchange_fn exchange, void* arg) {
  if (!c || !exchange || shards < 2 || shards > 8 || shard < 0 || shard >= shards) return SWZ_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(joint_m);
  JointState*& js = joint_states()[c];
  if (!js) js = new JointState;
  if (!js->opened.empty()) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_joint_root_begin: the previous joint root was not ended");
  js->c = c;
  js->views.assign((size_t)shards, swz::MdPeerView{});
  js->sr.shard = shard;
  js->sr.shards = shards;
  js->sr.views = js->views.data();
  js->sr.barrier = joint_exchange;
  js->sr.barrier_arg = js;
  js->exchange = exchange;
  js->exchange_arg = arg;
  c->md_shard_root = &js->sr;
  c->md_shard_root_published = true;  // (until swz_shard_joint_root_end, which the driver calls after its barrier)
  return SWZ_OK;
}

// A rank whose swz_shard_begin_device did not reach the sweep (no points, or an error before it) meets the others here.
int swz_shard_joint_root_meet(swz_ctx* c, int ok) {
  if (!c) return SWZ_ERR_BAD_ARG;
  JointState* js = nullptr;
  {
    std::lock_guard<std::mutex> lk(joint_m);
    auto it = joint_states().find(c);
    if (it != joint_states().end()) js = it->second;
  }
  if (!js || !c->md_shard_root) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_joint_root_meet: no joint root is open");
  swz::MdPeerView& mine = js->views[js->sr.shard];
  if (mine.entered) return SWZ_OK;
  mine = swz::MdPeerView{};
  mine.status = ok ? SWZ_OK : SWZ_ERR_INTERNAL;
  mine.entered = 1;
  joint_exchange(js);
  return SWZ_OK;
}

int swz_shard_joint_root_end(swz_ctx* c) {
  if (!c) return SWZ_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(joint_m);
  auto it = joint_states().find(c);
  c->md_shard_root = nullptr;
  c->md_shard_root_published = false;
  if (it == joint_states().end()) return SWZ_OK;
  JointState* js = it->second;
  (void)hipSetDevice(c->device);
  for (auto& kv : js->opened) (void)hipIpcCloseMemHandle(kv.second);
  js->opened.clear();
  delete js;
  joint_states().erase(it);
  return SWZ_OK;
}

}  // extern "C"
