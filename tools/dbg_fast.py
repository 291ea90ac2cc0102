import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import schwarzwald_amd as swz
UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
n, concurrency, max_pts, sampler = 400000, 2, 2000, 2
rng = np.random.default_rng(n + concurrency)
xyz = rng.random((n, 3))
spacing = O.spacing_from_diagonal(*UNIT, 250)
ctx = swz.Context(0)
for strategy in (swz.ACCURATE, swz.FAST):
    o = O.tile(xyz, *UNIT, sampler, max_pts, spacing, strategy=strategy, fast_concurrency=concurrency)
    p = swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing, strategy=strategy, fast_concurrency=concurrency)
    g = ctx.tile(xyz, *UNIT, p)
    bad = np.nonzero(g.level != o["level"])[0]
    bad_dup = np.nonzero(g.dup != o["dup"])[0]
    print("strategy", strategy, "start", g.stats["fast_start_levels"], "level mismatches", len(bad), "dup mismatches", len(bad_dup))
    if len(bad):
        print(" first:", bad[:10], "gpu", g.level[bad[:10]], "ref", o["level"][bad[:10]])
        lv, cnt = np.unique(o["level"][bad], return_counts=True); print(" ref levels of mismatches", dict(zip(lv.tolist(), cnt.tolist())))
    if len(bad_dup):
        print(" first dup:", bad_dup[:10], "gpu", g.dup[bad_dup[:10]], "ref", o["dup"][bad_dup[:10]], "lvl", o["level"][bad_dup[:10]])

# ---- analyse the ACCURATE mismatch
strategy = swz.ACCURATE
o = O.tile(xyz, *UNIT, sampler, max_pts, spacing, strategy=strategy)
g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing))
pos = o["xyz_clamped"][o["perm"]]
sq = float(np.float32(spacing) * np.float32(spacing))
bad = np.nonzero(g.level != o["level"])[0]
cl = 5
for p in bad[:4]:
    d = pos[:p] - pos[p]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    near = np.nonzero(d2 < sq)[0]
    print("point", p, "key %016x" % int(o["keys"][p]), "earlier points within s:", len(near))
    for a in near:
        ca = int(o["keys"][a]) >> (63 - 3 * cl); cp = int(o["keys"][p]) >> (63 - 3 * cl)
        def xyz_of(c):
            x = y = z = 0
            for b in range(cl):
                tri = (c >> (3 * b)) & 7
                x |= ((tri >> 2) & 1) << b; y |= ((tri >> 1) & 1) << b; z |= (tri & 1) << b
            return x, y, z
        print("   earlier", a, "ref lvl", o["level"][a], "gpu lvl", g.level[a], "d2/s2 %.4f" % (d2[a] / sq), "cell", xyz_of(ca), "vs", xyz_of(cp))
