#!/usr/bin/env python3
"""How many points one GPU's multi-batch tiler holds: N points in K batches (x-strips, every fourth batch over the whole
cube), RANDOM_GRID, then the node table, the export, and a check that every point id sits in exactly one node file.
usage: tiler_capacity_probe.py [points] [batches]      (MI355X, round 4: 2.4 B in 24 batches 6.1 s, store 76 GB on the device;
3.0 B in 30 batches 17.1 s with 38 GB of the store's sides spilled to pinned host memory)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import schwarzwald_amd as swz
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda", 0)
ctx = swz.Context(0)
bmin, bmax = [0.0]*3, [1.0]*3
params = swz.TileParams(sampler=swz.RANDOM_GRID, max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal(bmin, bmax, 250))
per = n_total // k
t0 = time.perf_counter()
with swz.Tiler(ctx, bmin, bmax, params, n_total) as t:
    g = torch.Generator(device=dev); g.manual_seed(1)
    for b in range(k):
        xyz = torch.rand((per, 3), dtype=torch.float64, device=dev, generator=g)
        # tiles along x: batch b lies in [b/k, (b+1)/k) except every fourth batch, which covers everything
        if b % 4: xyz[:, 0] = (xyz[:, 0] + b) / k
        torch.cuda.synchronize()
        t.add_batch_device(xyz.data_ptr(), per)
        del xyz
    t.finalize()
    info = t.info()
    ns = int(info["num_stored"])
    print("stored", ns, "points", int(info["num_points"]), "nodes", int(info["num_nodes"]), "batches", int(info["num_batches"]), "%.1f s" % (time.perf_counter() - t0))
    assert ns == per * k
    ids = torch.empty(ns, dtype=torch.int32, device=dev)
    t.export_device(None, ids.data_ptr(), None)
    tb = t.node_table()
    print("node table: files", len(tb["level"]), "entries", int(tb["count"].sum()), "largest file", int(tb["count"].max()), "levels", int(tb["level"].min()), "..", int(tb["level"].max()))
    dev_b, host_b = t.store_residency()
    print("store residency GB: device %.1f host %.1f" % (dev_b / 1e9, host_b / 1e9))
ctx.release_workspace()
seen = torch.zeros(ns, dtype=torch.uint8, device=dev)
one = torch.ones(1, dtype=torch.uint8, device=dev)
step = 100_000_000
for a in range(0, ns, step):
    idx = ids[a:a + step].long() & 0xFFFFFFFF
    seen.index_put_((idx,), one.expand(idx.shape[0]), accumulate=True)
print("every point id exactly once:", bool((seen == 1).all().item()))
