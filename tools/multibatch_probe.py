#!/usr/bin/env python3
"""Per-batch wall times of the multi-batch tiler, device resident and staged.  usage: multibatch_probe.py N K SAMPLER"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import schwarzwald_amd as swz
N, K, sampler = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
ctx = swz.Context(0)
if os.environ.get("PROBE_OWN_STREAM") != "1":
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xyz = torch.empty((N, 3), dtype=torch.float64, device=dev)
ctx.generate_uniform_device(0x5C4A72A1D + 3, 0, N, xyz.data_ptr())
bmin, bmax = [0, 0, 0], [1, 1, 1]
p = swz.TileParams(sampler=getattr(swz, sampler), max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal(bmin, bmax, 250))
bounds = [(i * N) // K for i in range(K + 1)]
host = swz.pinned_empty((N, 3), np.float64)
torch.from_numpy(host).copy_(xyz)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    with swz.Tiler(ctx, bmin, bmax, p, capacity_hint=N) as t:
        times = [time.perf_counter() - t0]
        for i in range(K):
            t1 = time.perf_counter()
            st = t.add_batch_device(xyz[bounds[i]:bounds[i + 1]].data_ptr(), bounds[i + 1] - bounds[i])
            times.append(time.perf_counter() - t1)
        t1 = time.perf_counter()
        t.finalize()
        info = t.info()
        times.append(time.perf_counter() - t1)
    print("device rep %d: total %.1f ms; create, batches, finalize+info [ms]:" % (rep, (time.perf_counter() - t0) * 1e3),
          ["%.1f" % (x * 1e3) for x in times], "nodes", info["num_nodes"], flush=True)
for rep in range(int(os.environ.get('PROBE_REPS', '3'))):
    t0 = time.perf_counter()
    with swz.Tiler(ctx, bmin, bmax, p, capacity_hint=N) as t:
        times = [time.perf_counter() - t0]
        t.stage_batch(host[bounds[0]:bounds[1]])
        for i in range(K):
            t1 = time.perf_counter()
            if i + 1 < K:
                t.stage_batch(host[bounds[i + 1]:bounds[i + 2]])
            t2 = time.perf_counter()
            t.tile_staged()
            times.append((time.perf_counter() - t1, t2 - t1))
        t.finalize()
        info = t.info()
    print("staged rep %d: total %.1f ms; per batch (total, of which stage call) [ms]:" % (rep, (time.perf_counter() - t0) * 1e3),
          ["%.1f/%.1f" % (a * 1e3, b * 1e3) if isinstance(x, tuple) else "%.1f" % (x * 1e3) for x in times for a, b in [x if isinstance(x, tuple) else (x, 0)]],
          "wait %.1f ms" % info["staged_wait_ms"], flush=True)
