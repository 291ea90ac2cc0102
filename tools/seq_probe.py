#!/usr/bin/env python3
"""Uniform batch, then clustered batch, in one process (fresh context each) -- the order of tests/test_gpu_fullsize.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import schwarzwald_amd as swz
N = int(sys.argv[1])
dev = torch.device("cuda:0")
for kind in ("uniform", "clustered"):
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)  # same stream as the torch kernels that make the input
    xyz = torch.empty((N, 3), dtype=torch.float64, device=dev)
    ctx.generate_uniform_device(0x5C4A72A1D + 3, 0, N, xyz.data_ptr())
    if kind == "clustered":
        g = torch.Generator(device=dev); g.manual_seed(1234)
        k = N // 3
        xyz[:k, 2] = 0.3 + 0.05 * torch.sin(6.0 * xyz[:k, 0]) * torch.cos(4.0 * xyz[:k, 1]) + 0.0005 * torch.randn(k, dtype=torch.float64, device=dev, generator=g)
        xyz[k:2 * k] = 0.6 + 0.03 * torch.randn((k, 3), dtype=torch.float64, device=dev, generator=g)
        xyz.clamp_(0.0, 1.0)
    print("input checksum", kind, float(xyz.sum()), float(xyz[:, 2].std()), flush=True)
    for sampler in ("MIN_DISTANCE", "RANDOM_GRID"):
        keys = torch.empty(N, dtype=torch.int64, device=dev); perm = torch.empty(N, dtype=torch.int32, device=dev); level = torch.empty(N, dtype=torch.int8, device=dev)
        p = swz.TileParams(sampler=getattr(swz, sampler), max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250))
        t0 = time.perf_counter()
        sys.stderr.write("== %s %s\n" % (kind, sampler)); sys.stderr.flush()
        st = ctx.tile_device(xyz.data_ptr(), N, [0, 0, 0], [1, 1, 1], p, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
        torch.cuda.synchronize()
        print("%s N=%d %s: %.1f ms" % (kind, N, sampler, (time.perf_counter() - t0) * 1e3), flush=True)
    ctx.release_workspace()
    ctx.close()
    del xyz, keys, perm, level
