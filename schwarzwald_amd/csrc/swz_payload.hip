// swz_payload.hip -- node lists on the device, the permuted payload gather, and BinaryPersistence node files
// (reference: core/io/BinaryPersistence.h:45-193, BinaryPersistence.cpp:200-375,
// core/tiling/TilingAlgorithms.cpp:139, 232-236).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "swz_internal.h"
#include "swz_device.h"

namespace swz {

static const uint32_t ATTR_BYTES[SWZ_ATTR_COUNT] = {3, 12, 2, 1, 1, 8, 1, 1, 2, 1, 1, 1};
// order of the attribute arrays in a node file (BinaryPersistence.h:120-190: bit 10 before bit 9)
static const int FILE_ORDER[SWZ_ATTR_COUNT] = {SWZ_ATTR_RGB, SWZ_ATTR_NORMAL, SWZ_ATTR_INTENSITY, SWZ_ATTR_CLASSIFICATION,
                                               SWZ_ATTR_EDGE_OF_FLIGHT_LINE, SWZ_ATTR_GPS_TIME, SWZ_ATTR_NUMBER_OF_RETURNS,
                                               SWZ_ATTR_RETURN_NUMBER, SWZ_ATTR_POINT_SOURCE_ID, SWZ_ATTR_SCAN_ANGLE_RANK,
                                               SWZ_ATTR_SCAN_DIRECTION_FLAG, SWZ_ATTR_USER_DATA};

// ---------------------------------------------------------------------------------- node lists
__global__ __launch_bounds__(256) void level_key_kernel(const int8_t* __restrict__ level, uint32_t n, uint64_t* __restrict__ out,
                                                        uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int l = level[i];
  if (l < -1 || l > 20) atomicAdd(bad, 1u);
  out[i] = (uint64_t)(uint8_t)(l + 1) << 56;  // the stable partition looks at the top byte only
}

// position i of the level-grouped order starts a node: first of its level, or another key prefix
__global__ __launch_bounds__(256) void node_head_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ keys,
                                                        const int8_t* __restrict__ level, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = order[i];
  const int l = level[p];
  bool head = i == 0;
  if (!head) {
    const uint32_t q = order[i - 1];
    head = level[q] != l || (l >= 0 && (keys[q] >> level_shift(l)) != (keys[p] >> level_shift(l)));
  }
  flags[i] = head ? 1u : 0u;
}

__global__ __launch_bounds__(256) void node_table_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ keys,
                                                         const int8_t* __restrict__ level, uint32_t n,
                                                         const uint32_t* __restrict__ excl, uint32_t max_nodes,
                                                         int8_t* __restrict__ nlevel, uint64_t* __restrict__ nkey,
                                                         uint64_t* __restrict__ noffset) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = order[i];
  const int l = level[p];
  bool head = i == 0;
  if (!head) {
    const uint32_t q = order[i - 1];
    head = level[q] != l || (l >= 0 && (keys[q] >> level_shift(l)) != (keys[p] >> level_shift(l)));
  }
  if (!head) return;
  const uint32_t k = excl[i];
  if (k >= max_nodes) return;
  nlevel[k] = (int8_t)l;
  nkey[k] = l < 0 ? 0ull : ((keys[p] >> level_shift(l)) << level_shift(l));
  noffset[k] = i;
}

// ---------------------------------------------------------------------------------- payload gather
__global__ __launch_bounds__(256) void compose_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ order,
                                                      uint32_t n, uint32_t* __restrict__ src) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) src[i] = perm[order ? order[i] : i];
}

// rows of K elements of T; one thread per row (rows are at most 24 bytes)
template <typename T, int K>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ in, const uint32_t* __restrict__ src, uint32_t n,
                                                          T* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T* r = in + (size_t)src[i] * K;
  T v[K];
#pragma unroll
  for (int j = 0; j < K; ++j) v[j] = r[j];
#pragma unroll
  for (int j = 0; j < K; ++j) out[(size_t)i * K + j] = v[j];
}

template <typename T, int K>
static void launch_gather(swz_ctx* c, const void* in, const uint32_t* src, uint32_t n, void* out) {
  hipLaunchKernelGGL((gather_rows_kernel<T, K>), dim3(div_up(n, 256)), dim3(256), 0, c->stream, (const T*)in, src, n, (T*)out);
}

// ---------------------------------------------------------------------------------- node files
struct ByteSink {
  std::vector<unsigned char> buf;
  void put(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    buf.insert(buf.end(), b, b + n);
  }
};

static int fail(swz_ctx* c, int code, const std::string& msg) {
  if (c) return c->fail(code, msg.c_str());
  return code;
}

static int read_file(swz_ctx* c, const char* path, int compressed, std::vector<unsigned char>& out, size_t want_at_most) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(c, SWZ_ERR_BAD_ARG, std::string("cannot open ") + path);
  std::vector<unsigned char> raw;
  unsigned char tmp[1 << 16];
  size_t got;
  while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) raw.insert(raw.end(), tmp, tmp + got);
  fclose(f);
  if (!compressed) {
    out.swap(raw);
    return SWZ_OK;
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) return fail(c, SWZ_ERR_INTERNAL, "inflateInit failed");
  zs.next_in = raw.data();
  zs.avail_in = (uInt)raw.size();
  int rc = Z_OK;
  out.clear();
  while (rc != Z_STREAM_END && out.size() < want_at_most) {
    zs.next_out = tmp;
    zs.avail_out = sizeof(tmp);
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END) {
      inflateEnd(&zs);
      return fail(c, SWZ_ERR_BAD_ARG, std::string("corrupt zlib stream in ") + path);
    }
    out.insert(out.end(), tmp, tmp + (sizeof(tmp) - zs.avail_out));
  }
  inflateEnd(&zs);
  return SWZ_OK;
}

}  // namespace swz

using namespace swz;

extern "C" {

uint32_t swz_attribute_row_bytes(int attribute) {
  return (attribute >= 0 && attribute < SWZ_ATTR_COUNT) ? ATTR_BYTES[attribute] : 0u;
}

int swz_build_node_lists_device(swz_ctx* c, const uint64_t* d_keys_sorted, const int8_t* d_level, uint64_t n,
                                uint32_t* d_order_out, uint64_t max_nodes, int8_t* node_level_out, uint64_t* node_key_out,
                                uint64_t* node_offset_out, uint64_t* node_count_out, uint64_t* num_nodes_out) {
  if (!c || !num_nodes_out) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  *num_nodes_out = 0;
  if (n == 0) return SWZ_OK;
  if (n > 0xFFFFFFFFull) return c->fail(SWZ_ERR_BAD_ARG, "more than 2^32-1 points in one batch");
  if (!d_keys_sorted || !d_level || !d_order_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_build_node_lists_device: NULL buffer");
  const uint32_t m = (uint32_t)n;
  const uint32_t nb = div_up(m, 256);
  uint64_t* lkeys = nullptr;
  uint32_t *flags = nullptr, *cnt = nullptr;
  SWZ_TRY(c->get("nl_keys", (size_t)m, &lkeys));
  SWZ_TRY(c->get("nl_flags", (size_t)m, &flags));
  SWZ_TRY(c->get("nl_cnt", (size_t)2, &cnt));
  SWZ_HIP(c, hipMemsetAsync(cnt, 0, 8, c->stream));
  hipLaunchKernelGGL(level_key_kernel, dim3(nb), dim3(256), 0, c->stream, d_level, m, lkeys, cnt);
  SWZ_LAUNCH_CHECK(c);
  uint32_t starts[256];
  SWZ_TRY(partition_by_top_byte(c, lkeys, m, d_order_out, starts));
  hipLaunchKernelGGL(node_head_kernel, dim3(nb), dim3(256), 0, c->stream, d_order_out, d_keys_sorted, d_level, m, flags);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, flags, flags, m, cnt + 1, "nl"));
  uint32_t h[2] = {0, 0};
  SWZ_HIP(c, hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  if (h[0]) return c->fail(SWZ_ERR_BAD_ARG, "level out of range");
  const uint64_t nn = h[1];
  *num_nodes_out = nn;
  if (nn > max_nodes) return c->fail(SWZ_ERR_BAD_ARG, "max_nodes too small");
  if (!node_level_out || !node_key_out || !node_offset_out || !node_count_out) return SWZ_OK;
  int8_t* dl = nullptr;
  uint64_t *dk = nullptr, *doff = nullptr;
  SWZ_TRY(c->get("nl_level", (size_t)nn, &dl));
  SWZ_TRY(c->get("nl_key", (size_t)nn, &dk));
  SWZ_TRY(c->get("nl_off", (size_t)nn, &doff));
  hipLaunchKernelGGL(node_table_kernel, dim3(nb), dim3(256), 0, c->stream, d_order_out, d_keys_sorted, d_level, m, flags,
                     (uint32_t)nn, dl, dk, doff);
  SWZ_LAUNCH_CHECK(c);
  SWZ_HIP(c, hipMemcpyAsync(node_level_out, dl, nn, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipMemcpyAsync(node_key_out, dk, nn * 8, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipMemcpyAsync(node_offset_out, doff, nn * 8, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  for (uint64_t k = 0; k < nn; ++k) node_count_out[k] = (k + 1 < nn ? node_offset_out[k + 1] : n) - node_offset_out[k];
  return SWZ_OK;
}

int swz_gather_payload_device(swz_ctx* c, const uint32_t* d_perm, const uint32_t* d_order, uint64_t n, const double* d_xyz,
                              const swz_attribute_columns* d_in, double* d_xyz_out, const swz_attribute_columns* d_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (n == 0) return SWZ_OK;
  if (n > 0xFFFFFFFFull) return c->fail(SWZ_ERR_BAD_ARG, "more than 2^32-1 points in one batch");
  if (!d_perm) return c->fail(SWZ_ERR_BAD_ARG, "swz_gather_payload_device: NULL perm");
  if ((d_xyz == nullptr) != (d_xyz_out == nullptr)) return c->fail(SWZ_ERR_BAD_ARG, "positions need input and output");
  const uint32_t m = (uint32_t)n;
  uint32_t* src = nullptr;
  SWZ_TRY(c->get("payload_src", (size_t)m, &src));
  uint64_t bytes = 12ull * n;  // perm + order read, composed index written
  hipLaunchKernelGGL(compose_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, d_perm, d_order, m, src);
  SWZ_LAUNCH_CHECK(c);
  {
    ProfScope ps(c, "payload_gather", 0, 1);
    if (d_xyz) {
      launch_gather<double, 3>(c, d_xyz, src, m, d_xyz_out);
      bytes += 52ull * n;
    }
    for (int a = 0; a < SWZ_ATTR_COUNT; ++a) {
      const void* in = d_in ? d_in->column[a] : nullptr;
      void* out = d_out ? d_out->column[a] : nullptr;
      if (!in || !out) continue;
      switch (ATTR_BYTES[a]) {
        case 3: launch_gather<uint8_t, 3>(c, in, src, m, out); break;
        case 12: launch_gather<float, 3>(c, in, src, m, out); break;
        case 2: launch_gather<uint16_t, 1>(c, in, src, m, out); break;
        case 8: launch_gather<double, 1>(c, in, src, m, out); break;
        default: launch_gather<uint8_t, 1>(c, in, src, m, out); break;
      }
      bytes += (4ull + 2ull * ATTR_BYTES[a]) * n;
    }
    SWZ_LAUNCH_CHECK(c);
    ps.bytes = bytes;
  }
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}

int swz_node_name(int8_t node_level, uint64_t node_key, char* name_out) {
  if (!name_out || node_level < -1 || node_level > 20) return SWZ_ERR_BAD_ARG;
  int k = 0;
  name_out[k++] = 'r';
  for (int l = 0; l <= node_level; ++l) name_out[k++] = (char)('0' + ((node_key >> level_shift(l)) & 7u));
  name_out[k] = 0;
  return SWZ_OK;
}

int swz_node_name_entwine(int8_t node_level, uint64_t node_key, char* name_out) {
  if (!name_out || node_level < -1 || node_level > 20) return SWZ_ERR_BAD_ARG;
  uint64_t x = 0, y = 0, z = 0;
  for (int l = 0; l <= node_level; ++l) {
    const uint32_t o = (uint32_t)(node_key >> level_shift(l)) & 7u;
    x = (x << 1) | ((o >> 2) & 1u);
    y = (y << 1) | ((o >> 1) & 1u);
    z = (z << 1) | (o & 1u);
  }
  snprintf(name_out, 72, "%d-%llu-%llu-%llu", node_level + 1, (unsigned long long)x, (unsigned long long)y,
           (unsigned long long)z);
  return SWZ_OK;
}

int swz_node_from_entwine_name(const char* name, int8_t* node_level_out, uint64_t* node_key_out) {
  if (!name || !node_level_out || !node_key_out) return SWZ_ERR_BAD_ARG;
  unsigned long long d = 0, x = 0, y = 0, z = 0;
  char tail = 0;
  if (sscanf(name, "%llu-%llu-%llu-%llu%c", &d, &x, &y, &z, &tail) != 4) return SWZ_ERR_BAD_ARG;
  if (d > 21) return SWZ_ERR_BAD_ARG;  // more levels than a 64-bit index holds (OctreeNodeIndex64)
  if (d < 64 && ((x >> d) || (y >> d) || (z >> d))) return SWZ_ERR_BAD_ARG;
  uint64_t key = 0;
  for (unsigned l = 0; l < d; ++l) {
    const unsigned b = (unsigned)d - 1u - l;
    const uint64_t o = (((x >> b) & 1ull) << 2) | (((y >> b) & 1ull) << 1) | ((z >> b) & 1ull);
    key |= o << level_shift((int)l);
  }
  *node_level_out = (int8_t)((int)d - 1);
  *node_key_out = key;
  return SWZ_OK;
}

int swz_node_bounds(int8_t node_level, uint64_t node_key, const double root_min[3], const double root_max[3],
                    double min_out[3], double max_out[3]) {
  if (!root_min || !root_max || !min_out || !max_out || node_level < -1 || node_level > 20) return SWZ_ERR_BAD_ARG;
  double mn[3] = {root_min[0], root_min[1], root_min[2]}, mx[3] = {root_max[0], root_max[1], root_max[2]};
  for (int l = 0; l <= node_level; ++l) {
    const uint32_t o = (uint32_t)(node_key >> level_shift(l)) & 7u;
    const uint32_t bit[3] = {(o >> 2) & 1u, (o >> 1) & 1u, o & 1u};
    for (int ax = 0; ax < 3; ++ax) {  // get_octant_bounds: min (+ extent / 2), max = min + extent / 2
      const double e = mx[ax] - mn[ax];
      if (bit[ax]) mn[ax] = mn[ax] + e / 2;
      mx[ax] = mn[ax] + e / 2;
    }
  }
  for (int ax = 0; ax < 3; ++ax) {
    min_out[ax] = mn[ax];
    max_out[ax] = mx[ax];
  }
  return SWZ_OK;
}

double swz_node_geometric_error(int8_t node_level, float spacing_at_root) {
  return spacing_at_root / std::pow(2.0, (double)(node_level + 1));
}

// Cesium3DTilesPersistence::on_write_node (core/io/Cesium3DTilesPersistence.cpp:80-156): every written node makes a
// Tileset for itself and for every ancestor that has none yet; name, geometricError = spacing_at_root / 2^depth
// (:91-92), bounds by descending from the root box octant by octant (:100-112, 136) translated by the global offset
// (:88); write_tilesets (:175-199) starts a new tileset.json every MAX_DEPTH + 1 = 3 levels.  The reference keeps
// children in the order its worker threads happened to write them; here they are ordered by octant.
int swz_tileset_build(uint64_t num_nodes, const int8_t* node_level, const uint64_t* node_key, const double root_min[3],
                      const double root_max[3], float spacing_at_root, const double global_offset[3], uint64_t max_out,
                      swz_tileset_node* out, uint64_t* num_out) {
  if (!num_out || (num_nodes && (!node_level || !node_key)) || !root_min || !root_max) return SWZ_ERR_BAD_ARG;
  *num_out = 0;
  // all nodes and their ancestors, keyed by (level, prefix): std::map iterates in (level, key) order
  std::map<std::pair<int, uint64_t>, uint8_t> all;
  for (uint64_t j = 0; j < num_nodes; ++j) {
    const int lv = node_level[j];
    if (lv < -1 || lv > 20) return SWZ_ERR_BAD_ARG;
    for (int l = lv; l >= -1; --l) {
      const uint64_t key = l < 0 ? 0ull : ((node_key[j] >> level_shift(l)) << level_shift(l));
      uint8_t& flag = all[{l, key}];
      if (l == lv) {
        flag = 1;
      } else if (all.size() == 0) {
        break;
      }
    }
  }
  const uint64_t total = all.size();
  *num_out = total;
  if (!out) return SWZ_OK;
  if (total > max_out) return SWZ_ERR_BAD_ARG;
  std::map<std::pair<int, uint64_t>, uint64_t> index;
  uint64_t k = 0;
  for (const auto& kv : all) index[kv.first] = k++;
  k = 0;
  for (const auto& kv : all) {
    swz_tileset_node& t = out[k];
    std::memset(&t, 0, sizeof(t));
    const int lv = kv.first.first;
    t.level = (int8_t)lv;
    t.key = kv.first.second;
    t.has_content = kv.second;
    t.parent = -1;
    t.first_child = -1;
    if (lv >= 0) {
      const uint64_t pkey = lv == 0 ? 0ull : ((t.key >> level_shift(lv - 1)) << level_shift(lv - 1));
      const uint64_t p = index[{lv - 1, pkey}];
      t.parent = (int64_t)p;
      if (out[p].first_child < 0) out[p].first_child = (int64_t)k;  // (level, key) order: children are contiguous
      out[p].num_children += 1;
    }
    t.is_tileset_root = ((lv + 1) % 3 == 0) ? 1 : 0;
    t.geometric_error = swz_node_geometric_error((int8_t)lv, spacing_at_root);
    swz_node_bounds((int8_t)lv, t.key, root_min, root_max, t.bounds_min, t.bounds_max);
    for (int ax = 0; ax < 3; ++ax) {  // AABB::translate(_global_offset), :88
      const double off = global_offset ? global_offset[ax] : 0.0;
      t.bounds_min[ax] += off;
      t.bounds_max[ax] += off;
    }
    ++k;
  }
  return SWZ_OK;
}

}  // extern "C"

// one node file (BinaryPersistence::persist_points, BinaryPersistence.h:46-57 / .cpp): bitmask, count, positions, then the
// attribute columns in file order.  No context: the files of a batch are written by several threads (swz_bin_persist_nodes).
static int bin_write_node_impl(const char* path, uint64_t count, const double* xyz, const swz_attribute_columns* columns, int compressed,
                               std::string* err) {
  uint32_t bitmask = 0;
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
    if (columns && columns->column[a]) bitmask |= 1u << a;
  FILE* f = fopen(path, "wb");
  if (!f) {
    *err = std::string("cannot write ") + path;
    return SWZ_ERR_BAD_ARG;
  }
  bool ok = true;
  if (!compressed) {
    // (the pieces go out as they lie in the caller's arrays: a node file is its header and slices of the columns)
    ok = fwrite(&bitmask, 1, 4, f) == 4 && fwrite(&count, 1, 8, f) == 8 && fwrite(xyz, 24, (size_t)count, f) == (size_t)count;
    for (int k = 0; k < SWZ_ATTR_COUNT && ok; ++k) {
      const int a = FILE_ORDER[k];
      if (bitmask & (1u << a)) ok = fwrite(columns->column[a], ATTR_BYTES[a], (size_t)count, f) == (size_t)count;
    }
  } else {
    ByteSink sink;
    sink.put(&bitmask, 4);
    sink.put(&count, 8);
    sink.put(xyz, (size_t)count * 24);
    for (int k = 0; k < SWZ_ATTR_COUNT; ++k) {
      const int a = FILE_ORDER[k];
      if (bitmask & (1u << a)) sink.put(columns->column[a], (size_t)count * ATTR_BYTES[a]);
    }
    uLongf cap = compressBound((uLong)sink.buf.size());
    std::vector<unsigned char> z(cap);
    ok = compress2(z.data(), &cap, sink.buf.data(), (uLong)sink.buf.size(), Z_BEST_SPEED) == Z_OK &&
         fwrite(z.data(), 1, cap, f) == cap;
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) {
    *err = std::string("short write to ") + path;
    return SWZ_ERR_INTERNAL;
  }
  return SWZ_OK;
}

extern "C" {

int swz_bin_write_node(swz_ctx* c, const char* path, uint64_t count, const double* xyz, const swz_attribute_columns* columns,
                       int compressed) {
  if (!path) return fail(c, SWZ_ERR_BAD_ARG, "swz_bin_write_node: NULL path");
  if (count == 0) return SWZ_OK;  // persist_points returns before opening the file
  if (!xyz) return fail(c, SWZ_ERR_BAD_ARG, "swz_bin_write_node: NULL positions");
  std::string err;
  const int st = bin_write_node_impl(path, count, xyz, columns, compressed, &err);
  return st == SWZ_OK ? SWZ_OK : fail(c, st, err);
}

int swz_bin_read_header(swz_ctx* c, const char* path, int compressed, uint32_t* bitmask_out, uint64_t* count_out) {
  if (!path || !bitmask_out || !count_out) return fail(c, SWZ_ERR_BAD_ARG, "swz_bin_read_header: NULL argument");
  std::vector<unsigned char> data;
  const int st = read_file(c, path, compressed, data, compressed ? 12 : (size_t)-1);
  if (st != SWZ_OK) return st;
  if (data.size() < 12) return fail(c, SWZ_ERR_BAD_ARG, std::string("truncated node file ") + path);
  memcpy(bitmask_out, data.data(), 4);
  memcpy(count_out, data.data() + 4, 8);
  return SWZ_OK;
}

int swz_bin_read_node(swz_ctx* c, const char* path, int compressed, double* xyz_out, const swz_attribute_columns* columns_out) {
  if (!path) return fail(c, SWZ_ERR_BAD_ARG, "swz_bin_read_node: NULL path");
  std::vector<unsigned char> data;
  const int st = read_file(c, path, compressed, data, (size_t)-1);
  if (st != SWZ_OK) return st;
  if (data.size() < 12) return fail(c, SWZ_ERR_BAD_ARG, std::string("truncated node file ") + path);
  uint32_t bitmask;
  uint64_t count;
  memcpy(&bitmask, data.data(), 4);
  memcpy(&count, data.data() + 4, 8);
  size_t need = 12 + (size_t)count * 24;
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
    if (bitmask & (1u << a)) need += (size_t)count * ATTR_BYTES[a];
  if (data.size() < need) return fail(c, SWZ_ERR_BAD_ARG, std::string("truncated node file ") + path);
  size_t at = 12;
  if (xyz_out) memcpy(xyz_out, data.data() + at, (size_t)count * 24);
  at += (size_t)count * 24;
  for (int k = 0; k < SWZ_ATTR_COUNT; ++k) {
    const int a = FILE_ORDER[k];
    if (!(bitmask & (1u << a))) continue;
    if (columns_out && columns_out->column[a]) memcpy(columns_out->column[a], data.data() + at, (size_t)count * ATTR_BYTES[a]);
    at += (size_t)count * ATTR_BYTES[a];
  }
  return SWZ_OK;
}

int swz_bin_persist_nodes(swz_ctx* c, const char* dir, uint64_t num_nodes, const int8_t* node_level, const uint64_t* node_key,
                          const uint64_t* node_offset, const uint64_t* node_count, const double* xyz,
                          const swz_attribute_columns* columns, int compressed) {
  if (!dir || (num_nodes && (!node_level || !node_key || !node_offset || !node_count || !xyz)))
    return fail(c, SWZ_ERR_BAD_ARG, "swz_bin_persist_nodes: NULL argument");
  for (uint64_t k = 0; k < num_nodes; ++k) {
    char name[24];
    if (swz_node_name(node_level[k], node_key[k], name) != SWZ_OK) return fail(c, SWZ_ERR_BAD_ARG, "bad node level");
  }
  // The files are independent: a few host threads take the nodes by ticket (the reference persists its nodes from the
  // tasks of its tiling graph, TilingAlgorithms.cpp:330-334).  One thread wrote 2.7 GB/s -- a hundredth of what the device
  // hands over.  SWZ_BIN_WRITER_THREADS: the number of threads (default: the host's, at most 32).
  unsigned threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  if (c)
    if (const char* e = c->opt("SWZ_BIN_WRITER_THREADS")) threads = (unsigned)std::max(1, atoi(e));
  threads = (unsigned)std::min<uint64_t>(threads, std::max<uint64_t>(num_nodes, 1));
  std::atomic<uint64_t> next{0};
  std::atomic<int> status{SWZ_OK};
  std::mutex err_m;
  std::string first_err;
  auto work = [&]() {
    for (;;) {
      const uint64_t k = next.fetch_add(1);
      if (k >= num_nodes || status.load() != SWZ_OK) return;
      if (node_count[k] == 0) continue;  // persist_points returns before opening the file
      char name[24];
      (void)swz_node_name(node_level[k], node_key[k], name);
      const std::string path = std::string(dir) + "/" + name + (compressed ? ".binz" : ".bin");
      swz_attribute_columns cols;
      for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
        cols.column[a] = (columns && columns->column[a])
                           ? (void*)((unsigned char*)columns->column[a] + (size_t)node_offset[k] * ATTR_BYTES[a])
                           : nullptr;
      std::string err;
      const int st = bin_write_node_impl(path.c_str(), node_count[k], xyz + (size_t)node_offset[k] * 3, &cols, compressed, &err);
      if (st != SWZ_OK) {
        std::lock_guard<std::mutex> lk(err_m);
        if (status.load() == SWZ_OK) {
          first_err = err;
          status.store(st);
        }
        return;
      }
    }
  };
  if (threads <= 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
  return status.load() == SWZ_OK ? SWZ_OK : fail(c, status.load(), first_err);
}

}  // extern "C"

// ---------------------------------------------------------------------------------- LAS point records (F2)
namespace swz {

struct LasArgs {
  double scale[3], offset[3], mn[3], mx[3];
  uint32_t format, record_bytes;
  double* xyz;
  void* col[SWZ_ATTR_COUNT];
};

constexpr int LAS_POINTS_PER_BLOCK = 256;
constexpr int LAS_MAX_RECORD = 96;  // staged through LDS; longer records (many extra bytes) are read directly

__device__ __forceinline__ uint32_t las_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ int32_t las_i32(const uint8_t* p) {
  return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

__device__ __forceinline__ void las_unpack(const LasArgs& a, const uint8_t* r, uint32_t i) {
  // position_from_las_point (LASFile.cpp:79-94): offset + X * scale, then min(max, max(min, p)) per axis
  if (a.xyz) {
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      double p = a.offset[ax] + (double)las_i32(r + 4 * ax) * a.scale[ax];
      p = fmin(a.mx[ax], fmax(a.mn[ax], p));
      a.xyz[(size_t)i * 3 + ax] = p;
    }
  }
  // the fields of a laszip_point the reference copies (las_read_points_into, LASFile.cpp:578-632), from the legacy record
  // (formats 0-5, LAS 1.2 / 1.3) or from the LAS 1.4 record (formats 6-10) the way LASzip's raw reader maps it onto them
  uint32_t ret, nret, dir, edge, cls, user, src_at, gps_at;
  int32_t angle;
  bool has_gps, has_rgb;
  uint32_t rgb_at;
  if (a.format >= 6u) {
    const uint32_t rn = r[14] & 15u, nor = (r[14] >> 4) & 15u;
    if (nor > 7u) {  // returns above 7 saturate (LASreadItemRaw_POINT14_LE::read)
      ret = rn > 6u ? (rn >= nor ? 7u : 6u) : rn;
      nret = 7u;
    } else {
      ret = rn & 7u;
      nret = nor;
    }
    dir = (r[15] >> 6) & 1u;
    edge = (r[15] >> 7) & 1u;
    cls = r[16] < 32u ? r[16] : 0u;  // the 5-bit classification field holds the LAS 1.4 classes below 32 only
    user = r[17];
    const int32_t raw = (int32_t)(int16_t)las_u16(r + 18);
    const float deg = 0.006f * (float)raw;                                     // I8_CLAMP(I16_QUANTIZE(0.006f * scan_angle))
    const int32_t q = deg >= 0.f ? (int32_t)(int16_t)(deg + 0.5f) : (int32_t)(int16_t)(deg - 0.5f);
    angle = q <= -128 ? -128 : (q >= 127 ? 127 : q);
    src_at = 20u;
    gps_at = 22u;
    has_gps = true;
    has_rgb = a.format == 7u || a.format == 8u || a.format == 10u;
    rgb_at = 30u;
  } else {
    const uint32_t bits = r[14];
    ret = bits & 7u;
    nret = (bits >> 3) & 7u;
    dir = (bits >> 6) & 1u;
    edge = (bits >> 7) & 1u;
    cls = r[15] & 31u;
    angle = (int32_t)(int8_t)r[16];
    user = r[17];
    src_at = 18u;
    gps_at = 20u;
    has_gps = a.format == 1u || a.format == 3u || a.format == 4u || a.format == 5u;
    has_rgb = a.format == 2u || a.format == 3u || a.format == 5u;
    rgb_at = has_gps ? 28u : 20u;
  }
  if (a.col[SWZ_ATTR_INTENSITY]) ((uint16_t*)a.col[SWZ_ATTR_INTENSITY])[i] = (uint16_t)las_u16(r + 12);
  if (a.col[SWZ_ATTR_RETURN_NUMBER]) ((uint8_t*)a.col[SWZ_ATTR_RETURN_NUMBER])[i] = (uint8_t)ret;
  if (a.col[SWZ_ATTR_NUMBER_OF_RETURNS]) ((uint8_t*)a.col[SWZ_ATTR_NUMBER_OF_RETURNS])[i] = (uint8_t)nret;
  if (a.col[SWZ_ATTR_SCAN_DIRECTION_FLAG]) ((uint8_t*)a.col[SWZ_ATTR_SCAN_DIRECTION_FLAG])[i] = (uint8_t)dir;
  if (a.col[SWZ_ATTR_EDGE_OF_FLIGHT_LINE]) ((uint8_t*)a.col[SWZ_ATTR_EDGE_OF_FLIGHT_LINE])[i] = (uint8_t)edge;
  if (a.col[SWZ_ATTR_CLASSIFICATION]) ((uint8_t*)a.col[SWZ_ATTR_CLASSIFICATION])[i] = (uint8_t)cls;
  if (a.col[SWZ_ATTR_SCAN_ANGLE_RANK]) ((int8_t*)a.col[SWZ_ATTR_SCAN_ANGLE_RANK])[i] = (int8_t)angle;
  if (a.col[SWZ_ATTR_USER_DATA]) ((uint8_t*)a.col[SWZ_ATTR_USER_DATA])[i] = (uint8_t)user;
  if (a.col[SWZ_ATTR_POINT_SOURCE_ID]) ((uint16_t*)a.col[SWZ_ATTR_POINT_SOURCE_ID])[i] = (uint16_t)las_u16(r + src_at);
  if (a.col[SWZ_ATTR_GPS_TIME]) {
    uint64_t v = 0;
    if (has_gps)
      for (int b = 7; b >= 0; --b) v = (v << 8) | r[gps_at + b];
    ((double*)a.col[SWZ_ATTR_GPS_TIME])[i] = __longlong_as_double((long long)v);
  }
  if (a.col[SWZ_ATTR_RGB]) {
    const uint8_t* c = r + rgb_at;
    uint8_t* o = (uint8_t*)a.col[SWZ_ATTR_RGB] + (size_t)i * 3;
    // las_read_points_into (LASFile.cpp:592-597): static_cast<uint8_t>(rgb[k] >> 8)
    for (int k = 0; k < 3; ++k) o[k] = has_rgb ? c[2 * k + 1] : (uint8_t)0;
  }
}

__global__ __launch_bounds__(LAS_POINTS_PER_BLOCK) void las_decode_kernel(const uint8_t* __restrict__ rec, uint32_t n, LasArgs a) {
  __shared__ uint32_t stage[LAS_POINTS_PER_BLOCK * LAS_MAX_RECORD / 4];
  const uint32_t first = blockIdx.x * LAS_POINTS_PER_BLOCK;
  const uint32_t cnt = min((uint32_t)LAS_POINTS_PER_BLOCK, n - first);
  const uint32_t i = first + threadIdx.x;
  if (a.record_bytes <= (uint32_t)LAS_MAX_RECORD && (a.record_bytes & 1u) == 0) {
    // the block's records are one contiguous, 4-byte aligned run (256 x an even length): coalesced word loads
    const size_t byte0 = (size_t)first * a.record_bytes;
    const uint32_t bytes = cnt * a.record_bytes, words = bytes / 4u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(rec + byte0);
    for (uint32_t w = threadIdx.x; w < words; w += LAS_POINTS_PER_BLOCK) stage[w] = src[w];
    if (threadIdx.x < (bytes & 3u))  // the last block may end in a partial word: no read past the buffer
      reinterpret_cast<uint8_t*>(stage)[words * 4u + threadIdx.x] = rec[byte0 + words * 4u + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < cnt) las_unpack(a, reinterpret_cast<const uint8_t*>(stage) + (size_t)threadIdx.x * a.record_bytes, i);
  } else if (threadIdx.x < cnt) {
    las_unpack(a, rec + (size_t)i * a.record_bytes, i);
  }
}

}  // namespace swz

extern "C" int swz_las_decode_device(swz_ctx* c, const uint8_t* d_records, uint64_t n, const swz_las_layout* layout,
                                     double* d_xyz_out, const swz_attribute_columns* d_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (!layout) return c->fail(SWZ_ERR_BAD_ARG, "swz_las_decode_device: NULL layout");
  static const uint32_t kMinBytes[11] = {20, 28, 26, 34, 57, 63, 30, 36, 38, 59, 67};
  if (layout->point_format > 10) return c->fail(SWZ_ERR_BAD_ARG, "LAS point data record formats 0-10 are decoded");
  if (layout->record_bytes < kMinBytes[layout->point_format])
    return c->fail(SWZ_ERR_BAD_ARG, "record length shorter than the point format");
  if (n == 0) return SWZ_OK;
  if (n > 0xFFFFFFFFull) return c->fail(SWZ_ERR_BAD_ARG, "more than 2^32-1 points in one batch");
  if (!d_records) return c->fail(SWZ_ERR_BAD_ARG, "swz_las_decode_device: NULL records");
  if (((uintptr_t)d_records & 3u) != 0) return c->fail(SWZ_ERR_BAD_ARG, "records must be 4-byte aligned");
  swz::LasArgs a{};
  uint64_t out_bytes = d_xyz_out ? 24 : 0;
  for (int k = 0; k < 3; ++k) {
    a.scale[k] = layout->scale[k];
    a.offset[k] = layout->offset[k];
    a.mn[k] = layout->min[k];
    a.mx[k] = layout->max[k];
  }
  a.format = layout->point_format;
  a.record_bytes = layout->record_bytes;
  a.xyz = d_xyz_out;
  for (int k = 0; k < SWZ_ATTR_COUNT; ++k) {
    a.col[k] = (d_out && k != SWZ_ATTR_NORMAL) ? d_out->column[k] : nullptr;  // LAS points carry no normals
    if (a.col[k]) out_bytes += swz::ATTR_BYTES[k];
  }
  swz::ProfScope ps(c, "las_decode", (uint64_t)n * (layout->record_bytes + out_bytes), 1);
  hipLaunchKernelGGL(swz::las_decode_kernel, dim3(swz::div_up((uint32_t)n, swz::LAS_POINTS_PER_BLOCK)),
                     dim3(swz::LAS_POINTS_PER_BLOCK), 0, c->stream, d_records, (uint32_t)n, a);
  SWZ_LAUNCH_CHECK(c);
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}
