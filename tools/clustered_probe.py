#!/usr/bin/env python3
"""Times one tile of surface-like clustered data (same generator as tests/test_gpu_fullsize.py) with the library's
per-level debug output.  usage: clustered_probe.py N SAMPLER [property]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import schwarzwald_amd as swz
N = int(sys.argv[1]); sampler = sys.argv[2]
dev = torch.device("cuda:0")
ctx = swz.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)  # same stream as the torch kernels that make the input
xyz = torch.empty((N, 3), dtype=torch.float64, device=dev)
ctx.generate_uniform_device(0x5C4A72A1D + 3, 0, N, xyz.data_ptr())
g = torch.Generator(device=dev); g.manual_seed(1234)
k = N // 3
xyz[:k, 2] = 0.3 + 0.05 * torch.sin(6.0 * xyz[:k, 0]) * torch.cos(4.0 * xyz[:k, 1]) + 0.0005 * torch.randn(k, dtype=torch.float64, device=dev, generator=g)
xyz[k:2 * k] = 0.6 + 0.03 * torch.randn((k, 3), dtype=torch.float64, device=dev, generator=g)
xyz.clamp_(0.0, 1.0)
part = os.environ.get('PROBE_PART')  # experiments: one component of the cloud only
if part == 'sheet': xyz = xyz[:k].contiguous()
elif part == 'blob': xyz = xyz[k:2 * k].contiguous()
elif part == 'uniform': xyz = xyz[2 * k:].contiguous()
N = xyz.shape[0]
keys = torch.empty(N, dtype=torch.int64, device=dev); perm = torch.empty(N, dtype=torch.int32, device=dev); level = torch.empty(N, dtype=torch.int8, device=dev)
p = swz.TileParams(sampler=getattr(swz, sampler), max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250),
                   flags=swz.FLAG_MIN_DISTANCE_PROPERTY if len(sys.argv) > 3 and sys.argv[3] == 'property' else 0)
for rep in range(2):  # the first pass grows the workspace
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = ctx.tile_device(xyz.data_ptr(), N, [0, 0, 0], [1, 1, 1], p, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
    torch.cuda.synchronize()
    print("N=%d %s %s: %.1f ms = %.0f Mpts/s" % (N, sampler, " ".join(sys.argv[3:]), (time.perf_counter() - t0) * 1e3,
                                                N / (time.perf_counter() - t0) / 1e6), st, flush=True)
