// Times one sharded batch through swz_group_tile (C++ host, one process): N shards -- on N devices, or all on device 0
// when the box has fewer -- each starting with P uniform points of the unit cube, exact MIN_DISTANCE, d = 250.
// Prints the wall time of the call and every shard's root interval, with the root swept by all shards at once
// (SWZ_GROUP_JOINT_ROOT unset) and in turns (SWZ_GROUP_JOINT_ROOT=0).
//   group_bench SHARDS POINTS_PER_SHARD REPS DEVICES [TRANSPORT 0 = peer copies | 1 = RCCL] [BATCHES] [SAMPLER 0..3] [STRATEGY 0 | 1] [WARMUP] [STAGED 0 | 1]
// STAGED = 1 (with BATCHES > 1; BASELINE config 5's shape): every shard's batches come from PINNED HOST memory with RGB +
// intensity columns along (29 B per point), batch k + 1 copied on the shards' copy streams under the kernels of batch k
// (swz_group_stage_batch / swz_group_tile_staged); the wall time then includes the host link.
// BATCHES > 1: the shard's points in that many batches through swz_group_add_batch (one swz_tiler per shard) and
// swz_group_finalize.  `bench.py --driver group` runs this program and reports its timings.
//   g++ -std=c++17 -O2 tools/group_bench.cpp -o /tmp/group_bench -Lschwarzwald_amd/lib -lswz_gpu -Wl,-rpath,$PWD/schwarzwald_amd/lib
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/swz_gpu.h"

int main(int argc, char** argv) {
  const int shards = argc > 1 ? std::atoi(argv[1]) : 8;
  const uint64_t per = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 25000000ull;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 3;
  const int ndev = argc > 4 ? std::atoi(argv[4]) : 1;
  const int transport = argc > 5 ? std::atoi(argv[5]) : 0;
  const int batches = argc > 6 ? std::max(1, std::atoi(argv[6])) : 1;
  const int sampler = argc > 7 ? std::atoi(argv[7]) : (int)SWZ_MIN_DISTANCE;
  const int strategy = argc > 8 ? std::atoi(argv[8]) : (int)SWZ_ACCURATE;
  const int warmup = argc > 9 ? std::atoi(argv[9]) : 0;  // reps before the kernel classes of shard 0 are timed (HIP events)
  const bool staged = argc > 10 && std::atoi(argv[10]) != 0 && batches > 1;
  const double mn[3] = {0, 0, 0}, mx[3] = {1, 1, 1};
  std::vector<int> dev(shards);
  for (int s = 0; s < shards; ++s) dev[s] = s % ndev;
  swz_group* g = nullptr;
  if (swz_group_create(shards, dev.data(), transport, &g) != SWZ_OK) {
    std::fprintf(stderr, "swz_group_create: %s\n", swz_group_last_error(nullptr));
    return 1;
  }
  std::vector<double*> d_xyz(shards);
  std::vector<uint64_t> n(shards, per);
  for (int s = 0; s < shards; ++s) {
    swz_ctx* c = swz_group_ctx(g, s);
    if (swz_device_alloc_on(c, per * 24, (void**)&d_xyz[s]) != SWZ_OK) return 2;
  }
  // staged: the shards' points and attribute columns in pinned host memory (filled once from the device-generated points)
  std::vector<double*> h_xyz(shards, nullptr);
  std::vector<uint8_t*> h_rgb(shards, nullptr);
  std::vector<uint16_t*> h_int(shards, nullptr);
  if (staged)
    for (int s = 0; s < shards; ++s) {
      if (swz_host_alloc_pinned(per * 24, (void**)&h_xyz[s]) != SWZ_OK || swz_host_alloc_pinned(per * 3, (void**)&h_rgb[s]) != SWZ_OK ||
          swz_host_alloc_pinned(per * 2, (void**)&h_int[s]) != SWZ_OK)
        return 2;
      for (uint64_t i = 0; i < per; ++i) {
        h_rgb[s][3 * i] = (uint8_t)i, h_rgb[s][3 * i + 1] = (uint8_t)(i >> 8), h_rgb[s][3 * i + 2] = (uint8_t)(i >> 16);
        h_int[s][i] = (uint16_t)(i * 7u);
      }
    }
  swz_tile_params p{};
  p.sampler = sampler;
  p.max_points_per_node = 20000;
  p.spacing_at_root = (float)(std::sqrt(3.0) / 250.0);
  p.max_depth = 100;
  p.strategy = strategy;
  p.fast_concurrency = 8;
  std::vector<swz_group_result> res(shards);
  for (int rep = 0; rep < reps; ++rep) {
    if (rep == warmup) {
      swz_profile_enable(swz_group_ctx(g, 0), 1);
      swz_profile_reset(swz_group_ctx(g, 0));
    }
    for (int s = 0; s < shards; ++s)
      if (swz_generate_uniform_device(swz_group_ctx(g, s), 0x5C4A72A1Dull + 3, (uint64_t)s * per, per, d_xyz[s]) != SWZ_OK) return 3;
    if (staged && rep == 0)
      for (int s = 0; s < shards; ++s)
        if (swz_copy_to_host(swz_group_ctx(g, s), h_xyz[s], d_xyz[s], per * 24) != SWZ_OK) return 3;
    const auto t0 = std::chrono::steady_clock::now();
    // batches: what every shard spent per stage, summed over the batches of the data set (exchange, root step, levels)
    std::vector<std::array<double, 3>> stage_sum(shards, std::array<double, 3>{{0, 0, 0}});
    auto add_stamps = [&]() {
      for (int s = 0; s < shards; ++s) {
        double t[4];
        swz_group_shard_timing(g, s, t);
        stage_sum[s][0] += t[0];
        stage_sum[s][1] += std::max(0.0, t[2] - t[1]);
        stage_sum[s][2] += std::max(0.0, t[3] - t[2]);
      }
    };
    if (batches > 1) {
      if (swz_group_tiler_open(g, mn, mx, &p, per) != SWZ_OK) {
        std::fprintf(stderr, "swz_group_tiler_open: %s\n", swz_group_last_error(g));
        return 4;
      }
      auto stage = [&](int b) {
        const uint64_t lo = per * b / batches, hi = per * (b + 1) / batches;
        std::vector<const double*> hx(shards);
        std::vector<swz_attribute_columns> ha(shards);
        std::vector<uint64_t> bn(shards, hi - lo);
        for (int s = 0; s < shards; ++s) {
          hx[s] = h_xyz[s] + lo * 3;
          ha[s] = swz_attribute_columns{};
          ha[s].column[SWZ_ATTR_RGB] = h_rgb[s] + lo * 3;
          ha[s].column[SWZ_ATTR_INTENSITY] = h_int[s] + lo;
        }
        return swz_group_stage_batch(g, hx.data(), ha.data(), bn.data());
      };
      if (staged) {
        if (stage(0) != SWZ_OK) {
          std::fprintf(stderr, "swz_group_stage_batch: %s\n", swz_group_last_error(g));
          return 4;
        }
        for (int b = 0; b < batches; ++b) {
          if ((b + 1 < batches && stage(b + 1) != SWZ_OK) || swz_group_tile_staged(g, nullptr) != SWZ_OK) {
            std::fprintf(stderr, "swz_group_tile_staged: %s\n", swz_group_last_error(g));
            return 4;
          }
          add_stamps();
        }
      }
      for (int b = 0; b < batches && !staged; ++b) {
        const uint64_t lo = per * b / batches, hi = per * (b + 1) / batches;
        std::vector<double*> bx(shards);
        std::vector<uint64_t> bn(shards, hi - lo);
        for (int s = 0; s < shards; ++s) bx[s] = d_xyz[s] + lo * 3;
        if (swz_group_add_batch(g, bx.data(), nullptr, bn.data(), nullptr) != SWZ_OK) {
          std::fprintf(stderr, "swz_group_add_batch: %s\n", swz_group_last_error(g));
          return 4;
        }
        add_stamps();
      }
      if (swz_group_finalize(g, nullptr) != SWZ_OK || swz_group_tiler_close(g) != SWZ_OK) {
        std::fprintf(stderr, "swz_group_finalize: %s\n", swz_group_last_error(g));
        return 4;
      }
    } else if (swz_group_tile(g, d_xyz.data(), nullptr, n.data(), mn, mx, &p, res.data()) != SWZ_OK) {
      std::fprintf(stderr, "swz_group_tile: %s\n", swz_group_last_error(g));
      return 4;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("rep %d: %d shards x %llu points: %.1f ms = %.0f Mpoints/s;", rep, shards, (unsigned long long)per, ms, shards * per / ms / 1e3);
    double first_root = 1e30, last_root = 0, last_all = 0;
    for (int s = 0; s < shards; ++s) {
      double t[4];
      swz_group_shard_timing(g, s, t);
      std::printf(" s%d root %.0f-%.0f levels %.0f;", s, t[1], t[2], t[3]);
      first_root = std::min(first_root, t[1]);
      last_root = std::max(last_root, t[2]);
      last_all = std::max(last_all, t[3]);
    }
    std::printf(" root phase %.1f ms, all done at %.1f ms\n", last_root - first_root, last_all);
    // one line per shard for bench.py: ms since the call began -- exchange done, root begun, root done, levels done
    for (int s = 0; s < shards; ++s) {
      double t[4];
      swz_group_shard_timing(g, s, t);
      if (batches > 1)  // (the last batch's stamps say little about a data set: the sums over its batches)
        std::printf("shardsum %d %d %.3f %.3f %.3f\n", rep, s, stage_sum[s][0], stage_sum[s][1], stage_sum[s][2]);
      else
        std::printf("shard %d %d %.3f %.3f %.3f %.3f\n", rep, s, t[0], t[1], t[2], t[3]);
    }
  }
  {
    // the kernel classes of shard 0 over the timed reps (its own stream's HIP events: what bench.py's roofline is made of)
    swz_kernel_stat ks[64];
    uint32_t nk = 0;
    if (swz_profile_get(swz_group_ctx(g, 0), ks, 64, &nk) == SWZ_OK)
      for (uint32_t i = 0; i < nk && i < 64; ++i)
        std::printf("class %s %llu %.4f %llu\n", ks[i].name, (unsigned long long)ks[i].launches, ks[i].total_ms, (unsigned long long)ks[i].algorithmic_bytes);
  }
  swz_group_destroy(g);
  return 0;
}
