#!/usr/bin/env python3
"""Times swz_las_decode_device: N synthetic LAS records of a point format on the device ->
positions + attribute columns, and the same followed by the tile of those points (decode + Morton + sort + sample), the
chain a reader thread feeds.  Prints the rates and the fraction of the HBM roofline (bytes read + written per record).
usage: las_probe.py [points] [point_format]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import schwarzwald_amd as swz

SIZES = {0: 20, 1: 28, 2: 26, 3: 34, 6: 30, 7: 36, 8: 38}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
fmt = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rb = SIZES[fmt]
dev = torch.device("cuda", 0)
ctx = swz.Context(0)
# records: X, Y, Z uniform in [0, 2^20) (scale 1e-3 -> a cube of ~1 km), the other fields random bytes
rec = torch.randint(0, 256, (n, rb), dtype=torch.uint8, device=dev)
xyz_i = torch.randint(0, 1 << 20, (n, 3), dtype=torch.int32, device=dev)
rec[:, 0:12] = xyz_i.view(torch.uint8).reshape(n, 12)
del xyz_i
scale, offset = [1e-3] * 3, [0.0] * 3
bmin, bmax = [0.0] * 3, [1048.576] * 3
xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
cols = {"intensity": torch.empty(n, dtype=torch.int16, device=dev), "classification": torch.empty(n, dtype=torch.uint8, device=dev)}
out_bytes = 24 + 2 + 1
if fmt in (2, 3, 7, 8):
    cols["rgb"] = torch.empty((n, 3), dtype=torch.uint8, device=dev)
    out_bytes += 3
if fmt in (1, 3, 6, 7, 8):
    cols["gps_time"] = torch.empty(n, dtype=torch.float64, device=dev)
    out_bytes += 8
ptrs = {k: v.data_ptr() for k, v in cols.items()}


def decode():
    ctx.las_decode_device(rec.data_ptr(), n, scale, offset, bmin, bmax, fmt, rb, xyz.data_ptr(), ptrs)


for _ in range(2):
    decode()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    decode()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
gbs = n * (rb + out_bytes) / ms / 1e6
print("LAS format %d, %d records of %d B -> xyz + %s: %.2f ms = %.0f Mpts/s, %d B per record moved = %.0f GB/s = %.1f %% of 8 TB/s"
      % (fmt, n, rb, "+".join(sorted(cols)), ms, n / ms / 1e3, rb + out_bytes, gbs, gbs / 80.0))
params = swz.TileParams(sampler=swz.GRID_CENTER, max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal(bmin, bmax, 250))
keys = torch.empty(n, dtype=torch.int64, device=dev)
perm = torch.empty(n, dtype=torch.int32, device=dev)
level = torch.empty(n, dtype=torch.int8, device=dev)
for i in range(3):
    if i == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    decode()
    ctx.tile_device(xyz.data_ptr(), n, bmin, bmax, params, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
torch.cuda.synchronize()
ms2 = (time.perf_counter() - t0) / 2 * 1e3
print("decode + tile (GRID_CENTER, d = 250): %.2f ms = %.0f Mpts/s" % (ms2, n / ms2 / 1e3))
