// swz_mindist.hip -- MIN_DISTANCE (K4c): exact greedy minimum-distance sampling on the GPU.
//
// Reference: PoissonDiskSampling::sample_points (core/tiling/Sampling.h:421-471) feeds the node's
// Morton-sorted points one by one into SparseGrid::add (core/datastructures/SparseGrid.cpp:116-146),
// which accepts a point iff no previously accepted point is closer than the spacing
// (GridCell::isDistant, GridCell.cpp:43-58: squared double distance < float-squared spacing).  The
// hash grid there is only an accelerator; the result is the lexicographically-first maximal
// independent set in Morton order:  accept(i) <=> for all accepted j < i : d2(i, j) >= s2.
//
// Exact parallel form used here.  Cut every sampled node into octree cells at least one spacing
// wide, so a point can only conflict with points of its own and the 26 adjacent cells.  Cells are
// runs of the sorted keys, and every point of a cell with a smaller Morton code precedes every point
// of a cell with a larger one.  Hence a cell's outcome depends only on the accepted points of its
// adjacent cells with smaller code: a dependency DAG, processed in topological rounds (Kahn): a
// cell becomes ready when all its earlier neighbours are final; one wavefront then runs the greedy
// for the cell (all points against the neighbours' accepted points in parallel, then the in-cell
// sequential part with wave ballots) and releases its later neighbours.  Every distance compare is
// the reference's, so the accepted set is bit-identical.
#include <algorithm>
#include <cmath>

#include "swz_level.h"

namespace swz {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr int MD_THREADS = 256;
constexpr int MD_WAVES = MD_THREADS / WAVE;
constexpr int MD_OWN_CAP = 128;  // accepted points of the current cell cached in LDS per wave

struct MdArgs {
  const uint64_t* akey;
  const uint32_t* aidx;
  uint32_t m;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* nstart;
  const double* X;
  const double* Y;
  const double* Z;
  uint8_t* taken;
  uint32_t* counters;
  // cells
  uint32_t* cstart;
  uint32_t* cend;
  uint32_t* crel;
  uint32_t* csnode;
  uint32_t* ndeps;
  uint32_t* acc_cnt;
  uint32_t* acc_list;   // per cell: sorted positions of its accepted points, at [cstart, cstart+acc_cnt)
  uint32_t* gridmap;    // [sample node][cell code] -> cell index
  uint32_t* queue[2];
  const uint32_t* snode_of;  // node -> compact index among sampled nodes
  uint32_t cell_shift;
  uint32_t cell_levels;      // octree levels between node and cell (cells per axis = 2^cell_levels)
  uint64_t cells_per_node;   // 8^cell_levels
  double sq_spacing;
};

__device__ __forceinline__ uint32_t md_spos(const uint32_t* aidx, uint32_t i) { return aidx ? aidx[i] : i; }

__global__ __launch_bounds__(256) void md_node_flag_kernel(const uint8_t* __restrict__ nmode, uint32_t nnodes,
                                                           uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < nnodes) out[j] = nmode[j] == MODE_SAMPLE ? 1u : 0u;
}

__device__ __forceinline__ bool md_is_head(const MdArgs& a, uint32_t i) {
  if (a.nmode[a.nid[i]] != MODE_SAMPLE) return false;
  return i == 0 || ((a.akey[i] >> a.cell_shift) != (a.akey[i - 1] >> a.cell_shift));
}

__global__ __launch_bounds__(256) void md_cell_head_kernel(MdArgs a, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.m) flags[i] = md_is_head(a, i) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void md_cell_build_kernel(MdArgs a, const uint32_t* __restrict__ excl) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m || !md_is_head(a, i)) return;
  const uint32_t c = excl[i];
  a.cstart[c] = i;
  a.crel[c] = (uint32_t)((a.akey[i] >> a.cell_shift) & (a.cells_per_node - 1ull));
  a.csnode[c] = a.snode_of[a.nid[i]];
  a.acc_cnt[c] = 0;
}

__global__ __launch_bounds__(256) void md_cell_end_kernel(MdArgs a, uint32_t ncells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint32_t s = a.cstart[c];
  const uint32_t node_end = a.nstart[a.nid[s] + 1];
  const uint32_t next = (c + 1 < ncells) ? a.cstart[c + 1] : a.m;
  a.cend[c] = next < node_end ? next : node_end;
  a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + a.crel[c]] = c;
}

// neighbour k (0..26, 13 = self) of the cell with Morton code rel; returns false when outside the node
__device__ __forceinline__ bool md_neighbour_code(uint32_t rel, uint32_t cell_levels, int k, uint32_t& nrel) {
  const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
  const int lim = 1 << cell_levels;
  const int x = (int)contract_bits_by_3((uint64_t)rel >> 2) + dx;
  const int y = (int)contract_bits_by_3((uint64_t)rel >> 1) + dy;
  const int z = (int)contract_bits_by_3((uint64_t)rel) + dz;
  if (x < 0 || y < 0 || z < 0 || x >= lim || y >= lim || z >= lim) return false;
  nrel = (uint32_t)(expand_bits_by_3((uint64_t)z) | (expand_bits_by_3((uint64_t)y) << 1) |
                    (expand_bits_by_3((uint64_t)x) << 2));
  return true;
}

// dependency count = existing adjacent cells with a smaller code; cells without any start round 0
__global__ __launch_bounds__(256) void md_deps_kernel(MdArgs a, uint32_t ncells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint32_t rel = a.crel[c];
  const uint64_t base = (uint64_t)a.csnode[c] * a.cells_per_node;
  uint32_t deps = 0;
  for (int k = 0; k < 27; ++k) {
    if (k == 13) continue;
    uint32_t nrel;
    if (!md_neighbour_code(rel, a.cell_levels, k, nrel)) continue;
    if (nrel < rel && a.gridmap[base + nrel] != NONE32) ++deps;
  }
  a.ndeps[c] = deps;
  if (deps == 0) {
    const uint32_t slot = atomicAdd(&a.counters[CTR_Q0], 1u);
    a.queue[0][slot] = c;
  }
}

__device__ __forceinline__ double bcast_f64(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

// One wavefront runs the reference's greedy for one ready cell.
__device__ void md_process_cell(const MdArgs& a, uint32_t c, uint32_t* qout, uint32_t* qout_count, double* own_x,
                                double* own_y, double* own_z) {
  const uint32_t l = lane_id();
  const uint32_t s = a.cstart[c], e = a.cend[c], rel = a.crel[c];
  const uint64_t gbase = (uint64_t)a.csnode[c] * a.cells_per_node;
  const double t = a.sq_spacing;

  // lanes 0..26 look up the adjacent cells
  uint32_t nb = NONE32;
  bool earlier = false, later = false;
  if (l < 27 && l != 13) {
    uint32_t nrel;
    if (md_neighbour_code(rel, a.cell_levels, (int)l, nrel)) {
      nb = a.gridmap[gbase + nrel];
      earlier = nb != NONE32 && nrel < rel;
      later = nb != NONE32 && nrel > rel;
    }
  }
  const uint64_t emask = __ballot(earlier);
  uint32_t own_cnt = 0;

  for (uint32_t p0 = s; p0 < e; p0 += WAVE) {
    const uint32_t p = p0 + l;
    const bool valid = p < e;
    uint32_t sp = 0;
    double px = 0, py = 0, pz = 0;
    if (valid) {
      sp = md_spos(a.aidx, p);
      px = a.X[sp];
      py = a.Y[sp];
      pz = a.Z[sp];
    }
    bool rej = !valid;
    // accepted points of the earlier adjacent cells (final: those cells finished in earlier rounds)
    uint64_t mm = emask;
    while (mm) {
      const int k = __ffsll((unsigned long long)mm) - 1;
      mm &= mm - 1;
      const uint32_t nbc = (uint32_t)__builtin_amdgcn_readlane((int)nb, k);
      const uint32_t cnt = a.acc_cnt[nbc];
      const uint32_t st = a.cstart[nbc];
      for (uint32_t a0 = 0; a0 < cnt; a0 += WAVE) {
        double qx = 0, qy = 0, qz = 0;
        if (a0 + l < cnt) {
          const uint32_t q = a.acc_list[st + a0 + l];
          qx = a.X[q];
          qy = a.Y[q];
          qz = a.Z[q];
        }
        const int nchunk = (int)((cnt - a0) < (uint32_t)WAVE ? (cnt - a0) : WAVE);
        for (int j = 0; j < nchunk; ++j) {
          const double bx = bcast_f64(qx, j), by = bcast_f64(qy, j), bz = bcast_f64(qz, j);
          if (sq_dist(px, py, pz, bx, by, bz) < t) rej = true;
        }
      }
      if (!__ballot(!rej)) break;
    }
    // accepted points of this cell from earlier chunks
    if (__ballot(!rej)) {
      const uint32_t in_lds = own_cnt < (uint32_t)MD_OWN_CAP ? own_cnt : (uint32_t)MD_OWN_CAP;
      for (uint32_t j = 0; j < in_lds; ++j)
        if (sq_dist(px, py, pz, own_x[j], own_y[j], own_z[j]) < t) rej = true;
      for (uint32_t j = MD_OWN_CAP; j < own_cnt; ++j) {  // overflow: read back what this wave published
        const uint32_t q = __hip_atomic_load(&a.acc_list[s + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sq_dist(px, py, pz, a.X[q], a.Y[q], a.Z[q]) < t) rej = true;
      }
    }
    // sequential greedy inside the chunk: the first surviving lane is accepted and rejects the
    // later lanes closer than the spacing
    uint64_t alive = __ballot(!rej);
    while (alive) {
      const int j = __ffsll((unsigned long long)alive) - 1;
      const double bx = bcast_f64(px, j), by = bcast_f64(py, j), bz = bcast_f64(pz, j);
      if ((int)l == j) {
        a.taken[p] = 1;
        __hip_atomic_store(&a.acc_list[s + own_cnt], sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (own_cnt < (uint32_t)MD_OWN_CAP) {
          own_x[own_cnt] = bx;
          own_y[own_cnt] = by;
          own_z[own_cnt] = bz;
        }
      }
      ++own_cnt;
      if ((int)l > j && !rej && sq_dist(px, py, pz, bx, by, bz) < t) rej = true;
      alive = __ballot(!rej && (int)l > j);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (l == 0) {
    a.acc_cnt[c] = own_cnt;
    atomicAdd(&a.counters[CTR_DONE_CELLS], 1u);
  }
  // release the later adjacent cells
  if (later) {
    const uint32_t old = atomicSub(&a.ndeps[nb], 1u);
    if (old == 1u) {
      const uint32_t slot = atomicAdd(qout_count, 1u);
      qout[slot] = nb;
    }
  }
}

__global__ __launch_bounds__(MD_THREADS) void md_round_kernel(MdArgs a, uint32_t round) {
  __shared__ double own[MD_WAVES][3][MD_OWN_CAP];
  const uint32_t w = threadIdx.x / WAVE;
  uint32_t* cin = &a.counters[CTR_Q0 + round % 3];
  uint32_t* cout = &a.counters[CTR_Q0 + (round + 1) % 3];
  if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[CTR_Q0 + (round + 2) % 3] = 0;
  const uint32_t nq = *cin;
  const uint32_t* qin = a.queue[round & 1];
  uint32_t* qout = a.queue[(round + 1) & 1];
  for (uint32_t e = blockIdx.x * MD_WAVES + w; e < nq; e += gridDim.x * MD_WAVES)
    md_process_cell(a, qin[e], qout, cout, own[w][0], own[w][1], own[w][2]);
}

int min_distance_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                       const LevelBuffers& lb, uint32_t nnodes, uint32_t sample_nodes, uint32_t sample_points,
                       uint32_t* rounds_out) {
  const uint32_t m = as.m;

  // cell size: as fine as the spacing allows, but coarse enough that an average cell holds >= 8
  // points (the dense [node][cell] lookup table then stays smaller than the point arrays)
  int cl = plan.cell_levels_geo;
  const double avg = (double)sample_points / (double)sample_nodes;
  int cl_density = 0;
  while (cl_density < 10 && std::pow(8.0, cl_density + 1) * 8.0 <= avg) ++cl_density;
  cl = std::max(0, std::min(cl, cl_density));
  const uint64_t cells_per_node = 1ull << (3 * cl);

  MdArgs a{};
  a.akey = as.akey;
  a.aidx = as.aidx;
  a.m = m;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.nstart = lb.nstart;
  a.X = sp.X;
  a.Y = sp.Y;
  a.Z = sp.Z;
  a.taken = lb.taken;
  a.counters = lb.counters;
  a.cell_levels = (uint32_t)cl;
  a.cells_per_node = cells_per_node;
  a.cell_shift = plan.node_shift == 63u ? level_shift(cl - 1) : plan.node_shift - 3u * (uint32_t)cl;
  if (plan.node_shift == 63u && cl == 0) a.cell_shift = 63u;
  a.sq_spacing = plan.sq_spacing;

  uint32_t* snode = nullptr;
  SWZ_TRY(c->get("md_snode", (size_t)nnodes, &snode));
  a.snode_of = snode;
  const uint32_t nb = div_up(m, 256);
  ProfScope ps(c, "sample_min_distance", (uint64_t)sample_points * 33ull, 1);
  hipLaunchKernelGGL(md_node_flag_kernel, dim3(div_up(nnodes, 256)), dim3(256), 0, c->stream, lb.nmode, nnodes, snode);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, snode, snode, nnodes, nullptr, "mdn"));

  // cells = runs of the cell prefix inside sampled nodes
  hipLaunchKernelGGL(md_cell_head_kernel, dim3(nb), dim3(256), 0, c->stream, a, lb.flags);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, lb.flags, lb.flags, m, lb.counters + CTR_NUM_CELLS, "mdc"));
  uint32_t ncells = 0;
  SWZ_HIP(c, hipMemcpyAsync(&ncells, lb.counters + CTR_NUM_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  if (ncells == 0) return SWZ_OK;

  SWZ_TRY(c->get("md_cstart", (size_t)ncells, &a.cstart));
  SWZ_TRY(c->get("md_cend", (size_t)ncells, &a.cend));
  SWZ_TRY(c->get("md_crel", (size_t)ncells, &a.crel));
  SWZ_TRY(c->get("md_csnode", (size_t)ncells, &a.csnode));
  SWZ_TRY(c->get("md_ndeps", (size_t)ncells, &a.ndeps));
  SWZ_TRY(c->get("md_acc_cnt", (size_t)ncells, &a.acc_cnt));
  SWZ_TRY(c->get("md_acc_list", (size_t)m, &a.acc_list));
  SWZ_TRY(c->get("md_queue0", (size_t)ncells, &a.queue[0]));
  SWZ_TRY(c->get("md_queue1", (size_t)ncells, &a.queue[1]));
  const uint64_t grid_entries = (uint64_t)sample_nodes * cells_per_node;
  SWZ_TRY(c->get("md_gridmap", (size_t)grid_entries, &a.gridmap));
  SWZ_HIP(c, hipMemsetAsync(a.gridmap, 0xFF, (size_t)grid_entries * 4, c->stream));

  hipLaunchKernelGGL(md_cell_build_kernel, dim3(nb), dim3(256), 0, c->stream, a, lb.flags);
  SWZ_LAUNCH_CHECK(c);
  const uint32_t cb = div_up(ncells, 256);
  hipLaunchKernelGGL(md_cell_end_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);
  hipLaunchKernelGGL(md_deps_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);

  // topological rounds; the host only looks at the done counter every `batch` launches
  const uint32_t grid = std::min<uint32_t>(2048u, std::max<uint32_t>(1u, div_up(ncells, MD_WAVES)));
  uint32_t round = 0, done = 0;
  const uint32_t batch = 256;
  const uint64_t max_rounds = (uint64_t)ncells + batch;
  while (done < ncells) {
    for (uint32_t b = 0; b < batch; ++b, ++round) {
      hipLaunchKernelGGL(md_round_kernel, dim3(grid), dim3(MD_THREADS), 0, c->stream, a, round);
    }
    SWZ_LAUNCH_CHECK(c);
    SWZ_HIP(c, hipMemcpyAsync(&done, lb.counters + CTR_DONE_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    if (round > max_rounds) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE dependency sweep did not terminate");
  }
  if (rounds_out) *rounds_out += round;
  return SWZ_OK;
}

}  // namespace swz
