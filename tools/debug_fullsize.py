#!/usr/bin/env python3
"""Diagnoses greedy-rule violations found by tests/test_gpu_fullsize.py: which nodes, false accepts or false rejects."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scipy.spatial import cKDTree
import oracle_lib as O
import schwarzwald_amd as swz
N = int(sys.argv[1]); L = int(sys.argv[2])
dev = torch.device("cuda:0")
ctx = swz.Context(0); ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xyz = torch.empty((N, 3), dtype=torch.float64, device=dev)
ctx.generate_uniform_device(0x5C4A72A1D + 3, 0, N, xyz.data_ptr())
g = torch.Generator(device=dev); g.manual_seed(1234); k = N // 3
xyz[:k, 2] = 0.3 + 0.05 * torch.sin(6.0 * xyz[:k, 0]) * torch.cos(4.0 * xyz[:k, 1]) + 0.0005 * torch.randn(k, dtype=torch.float64, device=dev, generator=g)
xyz[k:2 * k] = 0.6 + 0.03 * torch.randn((k, 3), dtype=torch.float64, device=dev, generator=g)
xyz.clamp_(0.0, 1.0)
keys = torch.empty(N, dtype=torch.int64, device=dev); perm = torch.empty(N, dtype=torch.int32, device=dev); level = torch.empty(N, dtype=torch.int8, device=dev)
spacing = swz.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250)
st = ctx.tile_device(xyz.data_ptr(), N, [0, 0, 0], [1, 1, 1], swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=20000, spacing_at_root=spacing),
                     keys.data_ptr(), perm.data_ptr(), level.data_ptr())
torch.cuda.synchronize(); ctx.release_workspace(); print(st, flush=True)
s = np.float32(spacing) / np.float32(2.0 ** (L + 1)); sq = float(np.float32(s) * np.float32(s))
shift = 63 - 3 * (L + 1)
node = keys >> shift
active = level >= L
uniq, counts = torch.unique_consecutive(node[active], return_counts=True)
big = uniq[counts > 20000]
print("level", L, "nodes", uniq.numel(), "sampling", big.numel(), "largest", int(counts.max()), flush=True)
# check whole nodes, largest first, up to a few million points each
order = torch.argsort(counts, descending=True)
for j in order[:int(sys.argv[3]) if len(sys.argv) > 3 else 3].tolist():
    nd = int(uniq[j]); cnt = int(counts[j])
    if cnt <= 20000: continue
    sel = torch.nonzero(active & (node == nd)).squeeze(1)
    if cnt > 6_000_000:
        print("node %o has %d points: checking the first 6 M in Morton order" % (nd, cnt)); sel = sel[:6_000_000]
    P = xyz[perm[sel].long()].cpu().numpy(); tk = (level[sel] == L).cpu().numpy()
    T = P[tk]; Ti = np.nonzero(tk)[0]
    pairs = cKDTree(P).sparse_distance_matrix(cKDTree(T), float(s) * (1 + 1e-9), output_type="ndarray")
    qi, ti = pairs["i"], Ti[pairs["j"]]
    d = P[qi] - P[ti]; d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    hit = (d2 < sq) & (ti < qi)
    has = np.zeros(len(P), bool); has[qi[hit]] = True
    bad = np.nonzero(tk == has)[0]
    fa = int((tk & has).sum()); fr = int((~tk & ~has).sum())
    print("node %o: %d points, %d taken, bad %d (false accepts %d, false rejects %d)" % (nd, len(P), int(tk.sum()), len(bad), fa, fr), flush=True)
    if len(bad):
        b = bad[:5]; print("  first bad local indices", b.tolist())
        K = keys[sel].cpu().numpy().view(np.uint64)
        for bi in b[:3]:
            js = ti[(qi == bi) & hit]
            for j in js[:2]:
                dd = P[bi] - P[j]
                print("   p=%d key %021o | q=%d key %021o | d/s %.4f | cell digits (level %d + 7): p %s q %s" % (
                    bi, int(K[bi]), j, int(K[j]), float(np.sqrt((dd * dd).sum()) / float(s)), L,
                    oct(int(K[bi]) >> (3 * (20 - L - 7)))[-7:], oct(int(K[j]) >> (3 * (20 - L - 7)))[-7:]))
        # oracle greedy on the node's points (Morton order) for comparison
        acc = O.sparse_grid_greedy(P, np.arange(len(P), dtype=np.uint32), [0, 0, 0], [1, 1, 1], float(s)) if len(P) <= 6_000_000 else None
        if acc is not None:
            acc = np.asarray(acc).astype(bool)
            print("  oracle taken %d, gpu taken %d, differing %d, first differing %s" % (int(acc.sum()), int(tk.sum()), int((acc != tk).sum()), np.nonzero(acc != tk)[0][:5].tolist()))
