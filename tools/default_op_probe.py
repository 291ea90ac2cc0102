#!/usr/bin/env python3
"""Per-batch wall times of the reference's default operating point (MIN_DISTANCE, FAST, batches of 10 M points as x-y tiles)
through the multi-batch tiler: which batches are outliers?  usage: default_op_probe.py [N] [K] [REPS]
(ORDER=uniform in the environment: every batch cut out of the whole cloud instead)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import schwarzwald_amd as swz
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
ctx = swz.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xyz = torch.empty((N, 3), dtype=torch.float64, device=dev)
ctx.generate_uniform_device(0x5C4A72A1D + 3, 0, N, xyz.data_ptr())
gx = int(np.ceil(np.sqrt(K))); gy = (K + gx - 1) // gx
for i in range(K if os.environ.get("ORDER", "tiles") == "tiles" else 0):
    lo, hi = (i * N) // K, ((i + 1) * N) // K
    xyz[lo:hi, 0].mul_(1.0 / gx).add_((i % gx) / gx)
    xyz[lo:hi, 1].mul_(1.0 / gy).add_((i // gx) / gy)
torch.cuda.synchronize()
bmin, bmax = [0, 0, 0], [1, 1, 1]
p = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal(bmin, bmax, 250),
                   strategy=swz.FAST, fast_concurrency=8)
for rep in range(REPS):
    t0 = time.perf_counter()
    held0 = ctx.workspace_bytes()
    with swz.Tiler(ctx, bmin, bmax, p, capacity_hint=N) as t:
        times, held = [], []
        for i in range(K):
            t1 = time.perf_counter()
            t.add_batch_device(xyz[(i * N) // K:((i + 1) * N) // K].data_ptr(), ((i + 1) * N) // K - (i * N) // K)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t1) * 1e3)
            held.append(ctx.workspace_bytes())
        t1 = time.perf_counter()
        t.finalize()
        torch.cuda.synchronize()
        fin = (time.perf_counter() - t1) * 1e3
    tot = (time.perf_counter() - t0) * 1e3
    med = float(np.median(times))
    out = [(i, round(x, 1), round((held[i] - (held[i - 1] if i else held0)) / 1e9, 2)) for i, x in enumerate(times) if x > 2.5 * med]
    grew = [i for i in range(K) if held[i] > (held[i - 1] if i else held0)]
    print("rep %d: %d batches grew the workspace, they took %.0f ms (the median batch x that many: %.0f ms)"
          % (rep, len(grew), sum(times[i] for i in grew), med * len(grew)))
    print("rep %d: total %.0f ms, finalize %.0f ms, batches: median %.2f ms, sum %.0f ms; outliers (batch, ms, workspace growth GB): %s; workspace %.1f -> %.1f GB"
          % (rep, tot, fin, med, sum(times), out, held0 / 1e9, held[-1] / 1e9), flush=True)
