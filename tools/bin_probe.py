#!/usr/bin/env python3
"""Times the back end of a batch: tile -> node lists -> payload gather on the device -> rows to the
host -> one BIN node file per node (swz_bin_persist_nodes, the reference's BinaryPersistence layout) into a directory.
usage: bin_probe.py [points] [directory]"""
import sys, os, time, shutil, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import schwarzwald_amd as swz

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
root = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="swz_bin_")
os.makedirs(root, exist_ok=True)
dev = torch.device("cuda", 0)
ctx = swz.Context(0)
bmin, bmax = [0.0] * 3, [1.0] * 3
xyz = torch.rand((n, 3), dtype=torch.float64, device=dev)
rgb = torch.randint(0, 256, (n, 3), dtype=torch.uint8, device=dev)
inten = torch.randint(0, 32767, (n,), dtype=torch.int16, device=dev)
params = swz.TileParams(sampler=swz.GRID_CENTER, max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal(bmin, bmax, 250))
keys = torch.empty(n, dtype=torch.int64, device=dev)
perm = torch.empty(n, dtype=torch.int32, device=dev)
level = torch.empty(n, dtype=torch.int8, device=dev)
order = torch.empty(n, dtype=torch.int32, device=dev)
out_xyz = torch.empty_like(xyz)
out_rgb, out_int = torch.empty_like(rgb), torch.empty_like(inten)


def stamp(label, t0, bytes_=None):
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print("%-46s %9.1f ms%s" % (label, ms, "  (%.2f GB/s)" % (bytes_ / ms / 1e6) if bytes_ else ""))
    return time.perf_counter()


ctx.tile_device(xyz.data_ptr(), n, bmin, bmax, params, keys.data_ptr(), perm.data_ptr(), level.data_ptr())  # warm-up
torch.cuda.synchronize()
t = time.perf_counter()
ctx.tile_device(xyz.data_ptr(), n, bmin, bmax, params, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
t = stamp("tile (GRID_CENTER, d = 250)", t)
nodes = ctx.build_node_lists_device(keys.data_ptr(), level.data_ptr(), n, order.data_ptr())
t = stamp("node lists (%d nodes)" % len(nodes["level"]), t)
ctx.gather_payload_device(perm.data_ptr(), order.data_ptr(), n, xyz.data_ptr(), {"rgb": rgb.data_ptr(), "intensity": inten.data_ptr()},
                          out_xyz.data_ptr(), {"rgb": out_rgb.data_ptr(), "intensity": out_int.data_ptr()})
t = stamp("payload gather into node order (29 B/pt)", t, n * 29 * 2)
h_xyz, h_rgb, h_int = swz.pinned_empty((n, 3), np.float64), swz.pinned_empty((n, 3), np.uint8), swz.pinned_empty((n,), np.uint16)
t = time.perf_counter()
for h, d in ((h_xyz, out_xyz), (h_rgb, out_rgb), (h_int, out_int)):  # (torch sees the pinned pages through a zero-copy view)
    torch.from_numpy(h.view(np.uint8).reshape(-1)).copy_(d.view(torch.uint8).reshape(-1), non_blocking=True)
t = stamp("rows to the host (pinned buffers)", t, n * 29)
ctx.bin_persist_nodes(root, nodes, h_xyz, {"rgb": h_rgb, "intensity": h_int})
t = stamp("BIN node files written to %s" % root, t, n * 29)
files = len(os.listdir(root))
size = sum(os.path.getsize(os.path.join(root, f)) for f in os.listdir(root))
print("%d files, %.2f GB" % (files, size / 1e9))
shutil.rmtree(root, ignore_errors=True)
