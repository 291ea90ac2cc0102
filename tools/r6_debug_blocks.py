"""Debugging aid for swz_mdblock.hip: one small tile against the oracle with a set of options, first mismatches printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import schwarzwald_amd as swz
UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
rng = np.random.default_rng(11)
xyz = rng.random((int(sys.argv[1]) if len(sys.argv) > 1 else 400000, 3))
d, mppn = 250, 2000
spacing = O.spacing_from_diagonal(*UNIT, d)
o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, mppn, spacing)
ctx = swz.Context(0)
import itertools
sets = [{"SWZ_SP_BLOCK_THREADS": t, "SWZ_SP_BLOCK_BITS": b} for t in ("64", "256") for b in ("6", "7", "8", "9")] + [{}, {}, {"SWZ_SP_BLOCK_WIDE": "1"}]
for opts in sets:
    for k, v in opts.items():
        ctx.set_option(k, v)
    g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
    bad = np.nonzero(g.level != o["level"])[0]
    print(opts, "mismatches", bad.size, "of", g.level.size, "first", bad[:8], "gpu", g.level[bad[:8]], "oracle", o["level"][bad[:8]], flush=True)
    for k in opts:
        ctx.set_option(k, None)

# ---- where is the accepted earlier neighbour that a wrongly accepted point missed?
ctx.set_option("SWZ_SP_BLOCK_BITS", "7")
g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
bad = np.nonzero(g.level != o["level"])[0]
keys = g.keys.astype(np.uint64)
P = xyz[g.perm]
def contract(v):
    r = 0
    for i in range(21):
        r |= ((int(v) >> (3 * i)) & 1) << i
    return r
for i in bad[:6]:
    L = int(o["level"][i]) - 1 if g.level[i] < o["level"][i] else int(g.level[i])
    L = int(g.level[i])
    s = np.float32(spacing) / np.float32(2.0 ** (L + 1))
    node = keys >> np.uint64(63 - 3 * (L + 1))
    same = np.nonzero((node == node[i]) & (np.arange(keys.size) < i) & (o["level"] == L))[0]
    dd = P[same] - P[i]
    d2 = (dd * dd).sum(axis=1)
    near = same[d2 < float(s) * float(s)]
    cl = 5 if L == 0 else 6
    cb = 20 - L - cl
    def cell(idx):
        k = int(keys[idx])
        x, y, z = contract(k >> 2), contract(k >> 1), contract(k)
        m = (1 << cl) - 1
        return ((x >> cb) & m, (y >> cb) & m, (z >> cb) & m)
    print("point", i, "level gpu", g.level[i], "oracle", o["level"][i], "cell", cell(i), "accepted earlier neighbours", [(int(j), cell(j)) for j in near[:4]], flush=True)
