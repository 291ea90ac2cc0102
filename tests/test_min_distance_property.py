"""MIN_DISTANCE in property mode (SWZ_FLAG_MIN_DISTANCE_PROPERTY, swz_mdprop.hip).  There is no reference result to
compare with point for point; what must hold is what the sampler is for and what the reference's author tests
(test/TestTiler.cpp:361-421), with the reference's compare (GridCell.cpp:43-58: (dx*dx + dy*dy) + dz*dz < the
float-squared spacing, widened to double):

  (a) spacing:    inside a sampled node no two taken points are closer than the node's spacing;
  (b) maximality: every point a sampled node hands down is closer than the spacing to one of its taken points;
  (c) everything else (keys, order, take-all of nodes with <= max_points points, one node per point) as in exact mode.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])


def _check_property(keys, level, pos, spacing_at_root, max_points, max_level, rng=None, box_points=None):
    """keys/level/pos in Morton order (numpy).  Verifies (a) and (b) on every sampled node (or, with box_points, on
    the points of a random box).  Returns (pairs checked, points checked for maximality)."""
    from scipy.spatial import cKDTree
    checked_a = checked_b = 0
    for L in range(-1, max_level + 1):
        s = np.float32(spacing_at_root) / np.float32(2.0 ** (L + 1))
        sq = float(np.float32(s) * np.float32(s))
        active = level >= L
        if not active.any():
            continue
        shift = 63 - 3 * (L + 1)
        node = (keys >> np.uint64(shift)) if shift < 63 else np.zeros_like(keys)
        uniq, inv, counts = np.unique(node[active], return_inverse=True, return_counts=True)
        sampled = np.zeros(keys.shape[0], dtype=bool)
        sampled[np.nonzero(active)[0]] = (counts > max_points)[inv]
        if not sampled.any():
            continue
        idx = np.nonzero(sampled)[0]
        if box_points is not None and idx.size > box_points:
            c = pos[idx[int(rng.integers(0, idx.size))]]
            h = 0.5 * (box_points / idx.size) ** (1.0 / 3.0)
            c = np.clip(c, h, 1.0 - h)
            near = np.all((pos[idx] >= c - h - 1.01 * float(s)) & (pos[idx] <= c + h + 1.01 * float(s)), axis=1)
            inner_box = (c - h, c + h)
            idx = idx[near]
        else:
            inner_box = None
        P, nd = pos[idx], node[idx]
        taken = level[idx] == L
        T, Tn = P[taken], nd[taken]
        assert T.shape[0] > 0
        tree = cKDTree(T)
        # (a) no two taken points of one node closer than the spacing
        pairs = tree.query_pairs(float(s) * (1.0 + 1e-9), output_type="ndarray")
        if pairs.size:
            d = T[pairs[:, 0]] - T[pairs[:, 1]]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            bad = (d2 < sq) & (Tn[pairs[:, 0]] == Tn[pairs[:, 1]])
            assert not bad.any(), "level %d: %d taken pairs closer than the spacing" % (L, int(bad.sum()))
        checked_a += int(T.shape[0])
        # (b) every point handed down has a taken point of its node within the spacing
        rest = ~taken
        if inner_box is not None:
            rest &= np.all((P >= inner_box[0]) & (P <= inner_box[1]), axis=1)
        Q, Qn = P[rest], nd[rest]
        if Q.shape[0]:
            m = cKDTree(Q).sparse_distance_matrix(tree, float(s) * (1.0 + 1e-9), output_type="ndarray")
            d = Q[m["i"]] - T[m["j"]]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            ok = (d2 < sq) & (Qn[m["i"]] == Tn[m["j"]])
            covered = np.zeros(Q.shape[0], dtype=bool)
            covered[m["i"][ok]] = True
            assert covered.all(), "level %d: %d points left out although no taken point is within the spacing" % (
                L, int((~covered).sum()))
            checked_b += int(Q.shape[0])
    return checked_a, checked_b


def _clustered(rng, n):
    k = n // 3
    a = np.column_stack([rng.random(k), rng.random(k), 0.3 + 0.002 * rng.standard_normal(k)])
    b = 0.6 + 0.03 * rng.standard_normal((k, 3))
    c = rng.random((n - 2 * k, 3))
    return np.clip(np.vstack([a, b, c]), 0.0, 1.0)[rng.permutation(n)]


@pytest.fixture(scope="module")
def ctx():
    import schwarzwald_amd as swz
    c = swz.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kind,d,max_points", [("uniform", 250, 2000), ("uniform", 40, 300), ("clustered", 250, 1000),
                                               ("clustered", 60, 200)])
def test_property_mode_small(ctx, kind, d, max_points):
    import schwarzwald_amd as swz
    rng = np.random.default_rng(d)
    n = 400000
    xyz = rng.random((n, 3)) if kind == "uniform" else _clustered(rng, n)
    sp = O.spacing_from_diagonal(*UNIT, d)
    exact = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=max_points, spacing_at_root=sp))
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=max_points, spacing_at_root=sp,
                            flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
    r = ctx.tile(xyz, *UNIT, params)
    r2 = ctx.tile(xyz, *UNIT, params)
    assert np.array_equal(r.level, r2.level)                                    # deterministic
    assert np.array_equal(r.keys, exact.keys) and np.array_equal(r.perm, exact.perm)
    assert r.level.min() >= -1 and r.stats["max_level"] == int(r.level.max())
    pos = r.xyz_clamped[r.perm]
    a, b = _check_property(r.keys, r.level, pos, sp, max_points, r.stats["max_level"])
    assert a > 0 and b > 0
    # same kind of sample as the exact greedy: the root takes a similar number of points
    te, tp = int((exact.level == -1).sum()), int((r.level == -1).sum())
    assert 0.85 < tp / te < 1.15, (te, tp)


def test_property_mode_kill_paths_agree(ctx):
    """Round 5: the first kill passes of levels with dozens of points per cell read the winners around their cell from
    per-cell records, the per-cell steps run over the list of alive points once it is short.  Both are accelerators: with
    either switched off (the loop over the mask everywhere / the cell grid in every round) the set must be the same, bit
    for bit -- and it must have the properties."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(55)
    n = 3000000
    xyz = rng.random((n, 3))
    sp = O.spacing_from_diagonal(*UNIT, 40)  # root cells of ~700 points, level 0 ~90: records; level 1 ~11: the mask loop
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=sp, flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
    base = ctx.tile(xyz, *UNIT, params)
    # round 6: the kill pass of the rounds over all points runs by blocks of cells with the winners around them in LDS (default);
    # without it the records / the mask loop of round 5 take over -- alone and combined with the older switches
    variants = ({"SWZ_MD_ROUNDS_BLOCK": "0"}, {"SWZ_MD_ROUNDS_BLOCK": "0", "SWZ_MD_ROUNDS_WLIST_MIN_POP": "1e9"},
                {"SWZ_MD_ROUNDS_CELL_LISTS": "0"}, {"SWZ_MD_ROUNDS_LIST": "0"}, {"SWZ_MD_ROUNDS_BLOCK": "0", "SWZ_MD_ROUNDS_LIST": "0"})
    for opts in variants:
        for opt, val in opts.items():
            ctx.set_option(opt, val)
        try:
            other = ctx.tile(xyz, *UNIT, params)
        finally:
            for opt in opts:
                ctx.set_option(opt, None)
        assert np.array_equal(other.level, base.level), opts
    a, b = _check_property(base.keys, base.level, base.xyz_clamped[base.perm], sp, 2000, base.stats["max_level"])
    assert a > 0 and b > 0


@pytest.mark.parametrize("kind", ["uniform", "clustered", "odd bounds"])
def test_property_mode_block_kill_pass_changes_nothing(ctx, kind):
    """The same comparison on clouds whose levels differ in cell population (blocks of 8, 4 and 2 cells per edge), on bounds
    that are not a power of two wide, and with nodes that take everything beside sampled ones."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(91)
    n = 2500000
    if kind == "clustered":
        xyz, bounds, d, mppn = _clustered(rng, n), UNIT, 250, 1500
    elif kind == "odd bounds":
        lo, side = np.array([-512.25, 1000.5, -3.125]), 777.7
        xyz, bounds, d, mppn = lo + rng.random((n, 3)) * side, (list(lo), list(lo + side)), 120, 5000
        xyz[: n // 3] = lo + (0.2 + 0.1 * rng.random((n // 3, 3))) * side   # a dense corner: sampled nodes beside take-all ones
    else:
        xyz, bounds, d, mppn = rng.random((n, 3)), UNIT, 100, 3000
    sp = O.spacing_from_diagonal(*bounds, d)
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=sp, flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
    ctx.set_option("SWZ_MD_ROUNDS_BLOCK_MIN_POP", "3")   # blocks only on levels of three points per cell or more (default: all)
    try:
        base = ctx.tile(xyz, *bounds, params)
    finally:
        ctx.set_option("SWZ_MD_ROUNDS_BLOCK_MIN_POP", None)
    ctx.set_option("SWZ_MD_ROUNDS_BLOCK", "0")
    try:
        other = ctx.tile(xyz, *bounds, params)
    finally:
        ctx.set_option("SWZ_MD_ROUNDS_BLOCK", None)
    assert np.array_equal(other.level, base.level)
    assert np.array_equal(ctx.tile(xyz, *bounds, params).level, base.level)   # ... and with the default choice per level
    if bounds is UNIT:
        a, b = _check_property(base.keys, base.level, base.xyz_clamped[base.perm], sp, mppn, base.stats["max_level"])
        assert a > 0 and b > 0


def test_property_mode_fast_strategy_and_multibatch(ctx):
    """The flag travels through FAST's reconstruction and the multi-batch tiler (AlwaysAdhereToMinSpacing nodes)."""
    import schwarzwald_amd as swz
    import torch
    rng = np.random.default_rng(8)
    n = 300000
    xyz = rng.random((n, 3))
    sp = O.spacing_from_diagonal(*UNIT, 100)
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=500, spacing_at_root=sp, strategy=swz.FAST,
                            fast_concurrency=2, flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
    r = ctx.tile(xyz, *UNIT, params)
    assert r.stats["fast_start_levels"] >= 3 and int(r.level.min()) >= r.stats["fast_start_levels"] - 1
    p2 = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=500, spacing_at_root=sp,
                        flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
    with swz.Tiler(ctx, *UNIT, p2) as t:
        for part in np.array_split(xyz, 3):
            d = torch.from_numpy(np.ascontiguousarray(part)).cuda()
            torch.cuda.synchronize()
            t.add_batch_device(d.data_ptr(), part.shape[0])
        t.finalize()
        info = t.info()
        table = t.node_table()
        ids = torch.empty(int(info["num_stored"]), dtype=torch.int32, device="cuda")
        t.export_device(0, ids.data_ptr(), 0)
        ids = ids.cpu().numpy().view(np.uint32)
    assert info["num_stored"] == n and np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))
    # spacing inside every node that has children (it was sampled in some batch)
    from scipy.spatial import cKDTree
    keyset = {(int(l), int(k)) for l, k in zip(table["level"], table["key"])}
    checked = 0
    for j in range(len(table["level"])):
        L, key = int(table["level"][j]), int(table["key"][j])
        has_child = any((L + 1, key | (o << (3 * (19 - L)))) in keyset for o in range(8)) if L < 20 else False
        if not has_child:
            continue
        pts = xyz[ids[int(table["offset"][j]):int(table["offset"][j] + table["count"][j])]]
        s = np.float32(sp) / np.float32(2.0 ** (L + 1))
        sq = float(np.float32(s) * np.float32(s))
        pairs = cKDTree(pts).query_pairs(float(s) * (1.0 + 1e-9), output_type="ndarray")
        if pairs.size:
            dd = pts[pairs[:, 0]] - pts[pairs[:, 1]]
            d2 = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
            assert not (d2 < sq).any(), (L, key)
        checked += 1
    assert checked > 0


def _check_property_level(torch, keys, perm, level, xyz, spacing, L, rng, target_points=1_500_000):
    """(a) and (b) on the points of a random box of level L (any level: the taken points within one spacing of the box
    are part of the selection).  Returns (taken points checked, left-out points checked)."""
    import test_gpu_fullsize as F
    from scipy.spatial import cKDTree
    b = F.box_of_level(torch, keys, perm, level, xyz, spacing, L, rng, target_points)
    if b is None:
        return 0, 0
    P, node, taken, c, h, s, sq = b["P"], b["node"], b["taken"], b["c"], b["h"], b["s"], b["sq"]
    T, Tn = P[taken], node[taken]
    assert T.shape[0] > 0, "level %d: a box of a sampling node without a single taken point" % L
    tree = cKDTree(T)
    # (a) no two taken points of one node closer than the spacing
    pairs = tree.query_pairs(float(s) * (1.0 + 1e-9), output_type="ndarray")
    if pairs.size:
        d = T[pairs[:, 0]] - T[pairs[:, 1]]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        bad = (d2 < sq) & (Tn[pairs[:, 0]] == Tn[pairs[:, 1]])
        assert not bad.any(), "level %d: %d taken pairs closer than the spacing" % (L, int(bad.sum()))
    # (b) every point of the box that was handed down has a taken point of its node within the spacing
    rest = ~taken & np.all((P >= c - h) & (P <= c + h), axis=1)
    if int(rest.sum()) > target_points:
        drop = rng.choice(np.nonzero(rest)[0], size=int(rest.sum()) - target_points, replace=False)
        rest[drop] = False
    Q, Qn = P[rest], node[rest]
    if Q.shape[0]:
        m = cKDTree(Q).sparse_distance_matrix(tree, float(s) * (1.0 + 1e-9), output_type="ndarray")
        d = Q[m["i"]] - T[m["j"]]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        ok = (d2 < sq) & (Qn[m["i"]] == Tn[m["j"]])
        covered = np.zeros(Q.shape[0], dtype=bool)
        covered[m["i"][ok]] = True
        assert covered.all(), "level %d: %d points left out although no taken point is within the spacing" % (L, int((~covered).sum()))
    return int(T.shape[0]), int(Q.shape[0])


@pytest.mark.parametrize("cloud", ["uniform", "clustered"])
def test_property_mode_full_size(cloud):
    """BASELINE's MIN_DISTANCE workload in property mode at the size bench.py reports it at -- 1 B uniform points on a
    288 GB part, sized from the free memory exactly like tests/test_gpu_fullsize.py (and failing instead of shrinking
    there) -- and on the 500 M surface-like cloud: spacing (a) and maximality (b) on random boxes of EVERY level, the
    dense ones (property-mode rounds) and the sparse ones (the exact block path) alike."""
    import torch
    import schwarzwald_amd as swz
    import test_gpu_fullsize as F
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    n = F.cloud_points(cloud)
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    xyz = F.make_cloud(torch, ctx, dev, n, cloud)
    sp = O.spacing_from_diagonal(*UNIT, 250)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    level = torch.empty(n, dtype=torch.int8, device=dev)
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=F.MAX_POINTS, spacing_at_root=sp,
                            flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
    stats = ctx.tile_device(xyz.data_ptr(), n, *UNIT, params, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
    torch.cuda.synchronize()
    ctx.release_workspace()
    assert int(level.min()) >= -1 and int(level.max()) == stats["max_level"]
    assert bool((keys[1:] >= keys[:-1]).all())
    rng = np.random.default_rng(3)
    total_a = total_b = 0
    for L in range(-1, stats["max_level"] + 1):
        for rep in range(2):
            a, b = _check_property_level(torch, keys, perm, level, xyz, sp, L, rng)
            total_a += a
            total_b += b
            F._log("%s cloud, %d points, property mode, level %d box %d: %d taken points at least the spacing apart, %d left-out points "
                   "each within the spacing of a taken one" % (cloud, n, L, rep, a, b))
    assert total_a > 1000 and total_b > 100000
    F._record("property MIN_DISTANCE: spacing + maximality, random boxes of every level (%d taken, %d left out)" % (total_a, total_b),
              cloud, n, stats["num_levels"], total_a + total_b)
    ctx.close()
