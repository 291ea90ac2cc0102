"""Safety nets for the parts of the oracle that no reference test vector pins (GRID_CENTER, MIN_DISTANCE,
JITTERED, tile_node, FAST): independent re-derivations and the invariants the reference's own disabled
integration tests assert (test/TestTiler.cpp:113-161 every point stored exactly once and inside its node,
:361-421 minimum distance on sampled nodes)."""
import numpy as np
import pytest

import oracle_lib as O

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])


def _brute_force_greedy(xyz_sorted, spacing):
    """accept(i) <=> all accepted j < i are at squared distance >= float(spacing)^2 (strict < rejects)."""
    sq = float(np.float32(spacing) * np.float32(spacing))
    acc = []
    flags = np.zeros(len(xyz_sorted), dtype=np.uint8)
    for i, p in enumerate(xyz_sorted):
        ok = True
        if acc:
            a = xyz_sorted[acc]
            d = a - p
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            ok = not np.any(d2 < sq)
        if ok:
            acc.append(i)
            flags[i] = 1
    return flags


@pytest.mark.parametrize("n,spacing", [(3000, 0.05), (6000, 0.02), (2000, 0.3)])
def test_sparse_grid_equals_brute_force_greedy(n, spacing):
    rng = np.random.default_rng(n)
    xyz = rng.random((n, 3))
    keys, _ = O.index_points(xyz, *UNIT)
    order = O.sort_by_key(keys)
    got = O.sparse_grid_greedy(xyz, order, *UNIT, spacing)
    assert np.array_equal(got, _brute_force_greedy(xyz[order], spacing))


def _node_of(key, level):
    return int(key) >> (3 * (20 - level)) if level >= 0 else 0


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
def test_tiler_invariants(sampler):
    rng = np.random.default_rng(5 + sampler)
    n, max_pts = 40000, 300
    xyz = rng.random((n, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 60 if sampler == O.MIN_DISTANCE else 250)
    r = O.tile(xyz, *UNIT, sampler, max_pts, spacing)
    assert r["status"] == 0
    assert sorted(r["perm"].tolist()) == list(range(n))          # every point exactly once
    assert r["level"].min() >= -1 and r["level"].max() == r["stats"]["max_level"]
    assert np.all(np.diff(r["keys"].astype(np.float64)) >= 0)
    pos = r["xyz_clamped"][r["perm"]]
    # inside the node's bounds
    for i in range(0, n, 97):
        lv = int(r["level"][i])
        mn, mx = O.bounds_from_morton_index(r["keys"][i], *UNIT, lv + 1)
        assert np.all(pos[i] >= np.array(mn)) and np.all(pos[i] <= np.array(mx))
    # a node that was sampled (it has descendants) never holds more points than grid cells / obeys spacing
    nodes = {}
    for i in range(n):
        lv = int(r["level"][i])
        nodes.setdefault((lv, _node_of(r["keys"][i], lv)), []).append(i)
    assert len(nodes) == r["stats"]["num_nodes"]
    if sampler == O.MIN_DISTANCE:
        has_descendants = set()
        for (lv, k) in nodes:
            while lv >= 0:
                lv, k = lv - 1, k >> 3
                has_descendants.add((lv, k if lv >= 0 else 0))
        checked = 0
        for (lv, k), members in nodes.items():
            if (lv, k) not in has_descendants or len(members) > 4000:
                continue
            s = np.float32(spacing / 2 ** (lv + 1))
            sq = float(s * s)
            p = pos[members]
            d = p[:, None, :] - p[None, :, :]
            d2 = (d ** 2).sum(-1) + np.eye(len(p)) * 10
            assert d2.min() >= sq
            checked += 1
        assert checked > 0


def test_tile_take_all_and_empty():
    r = O.tile(np.zeros((0, 3)), *UNIT, O.RANDOM_GRID, 10, 0.1)
    assert r["status"] == 0 and r["stats"]["num_nodes"] == 0
    rng = np.random.default_rng(1)
    xyz = rng.random((10, 3))
    for sampler in (O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED):
        r = O.tile(xyz, *UNIT, sampler, 10, 0.01)  # count <= max_points: the root takes everything
        assert r["status"] == 0 and np.all(r["level"] == -1) and r["stats"]["num_nodes"] == 1


def test_fast_reconstruction_invariants():
    rng = np.random.default_rng(9)
    n = 300000
    xyz = rng.random((n, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    r = O.tile(xyz, *UNIT, O.RANDOM_GRID, 2000, spacing, strategy=O.FAST, fast_concurrency=2)
    S = r["stats"]["fast_start_levels"]
    assert 3 <= S <= 6
    assert r["level"].min() == S - 1                     # nothing persisted above the start nodes ...
    assert np.all(r["dup"] < (1 << S))                   # ... except the reconstructed copies
    assert np.any(r["dup"] & 1)                          # the root was reconstructed
    # a point copied into level l must also be stored one level below (it was sampled from there)
    for lv in range(S - 1):
        has = (r["dup"] >> lv) & 1
        below = ((r["dup"] >> (lv + 1)) & 1) | (r["level"] == S - 1) if lv + 1 < S else r["level"] == S - 1
        assert np.all(below[has == 1] == 1)


def test_jittered_errors_match_reference_throws():
    rng = np.random.default_rng(2)
    xyz = rng.random((5000, 3))
    keys, _ = O.index_points(xyz, *UNIT)
    order = O.sort_by_key(keys)
    # fewer than 16 cells per axis: Sampling.h:632-635 throws
    t, _, _ = O.sample_points(O.JITTERED, 10, keys[order], order, xyz, 0, -1, *UNIT, 0.1, O.ALWAYS_ADHERE)
    assert t == O.ERR_JITTER_GRID_TOO_SMALL
    # a deep node: grid level >= 21 (Sampling.h:642-653)
    t, _, _ = O.sample_points(O.JITTERED, 10, keys[order], order, xyz, 0, 15, *UNIT, 0.001, O.ALWAYS_ADHERE)
    assert t == O.ERR_JITTER_NODE_TOO_DEEP


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.MIN_DISTANCE])
@pytest.mark.parametrize("strategy", [O.ACCURATE, O.FAST])
def test_threaded_oracle_gives_the_same_tiles(sampler, strategy):
    """orc_tile_mt spreads node tasks over worker threads like the reference's executor; nodes are independent,
    so the result must not depend on the thread count (this is the CPU baseline bench.py times)."""
    xyz = O.generate_uniform(77, 1_200_000)
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250)
    a = O.tile(xyz, [0, 0, 0], [1, 1, 1], sampler, 5000, spacing, strategy=strategy, threads=1)
    b = O.tile(xyz, [0, 0, 0], [1, 1, 1], sampler, 5000, spacing, strategy=strategy, threads=6)
    assert a["status"] == 0 and b["status"] == 0
    for k in ("keys", "perm", "level", "dup"):
        assert np.array_equal(a[k], b[k]), k
    assert a["stats"] == b["stats"]


# ------------------------------------------------------------------------------------------------------------------
# Independent characterisation of GRID_CENTER and JITTERED (no reference vector pins them): in the unit cube every
# box edge is a dyadic rational, so the cell centre / jitter target is exact in ANY evaluation order and a few lines of
# numpy say what sample_points must return -- in every run of equal key prefix at the grid level exactly one point,
# the FIRST (Morton order) with the smallest (dx*dx + dy*dy) + dz*dz to the target (std::min_element,
# Sampling.h:392-403 / :741-750).  The jitter tables are the reference's constants (Sampling.h:14-138).
def _compact3(v):
    out = np.zeros_like(v)
    for b in range(21):
        out |= ((v >> np.uint64(3 * b)) & np.uint64(1)) << np.uint64(b)
    return out


def _jitter_tables():
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "jitter_tables.inc")).read()
    out = {}
    for w in (16, 32, 64):
        body = text.split("SWZ_JITTER_TABLE(%d)" % w)[1].split("};")[0]
        nums = [int(x) for x in re.findall(r"\b\d+\b", body)]
        out[w] = np.array(nums[-16 * w:], dtype=np.int64).reshape(16, w)
    return out


def _expected_grid_sample(keys, pos, sampler, node_level, spacing_at_root):
    """taken flags of a sampled node at node_level whose points are keys/pos (Morton order), unit-cube root"""
    s_node = float(np.float32(spacing_at_root)) / 2.0 ** (node_level + 1)
    node_ext = 0.5 ** (node_level + 1)
    if sampler == O.GRID_CENTER:
        grid_level = max(-1, int(np.floor(np.log2(np.float32(1.0 / s_node)))) - 1)  # candidate_level_in_octree
    else:
        cells = 1 << int(np.floor(np.log2(np.uint32(node_ext / s_node))))            # prev_pow2((uint32) perfect count)
        levels = int(np.log2(cells))
        grid_level = node_level + levels
    depth = grid_level + 1
    cell = keys >> np.uint64(3 * (20 - grid_level))
    gx, gy, gz = _compact3(cell >> np.uint64(2)), _compact3(cell >> np.uint64(1)), _compact3(cell)
    size = 0.5 ** depth
    if sampler == O.GRID_CENTER:
        t = np.stack([(gx + 0.5) * size, (gy + 0.5) * size, (gz + 0.5) * size], axis=1)
    else:
        m = np.uint64(cells - 1)
        lx, ly, lz = (gx & m).astype(np.int64), (gy & m).astype(np.int64), (gz & m).astype(np.int64)
        tab = _jitter_tables()[16 if cells <= 16 else (32 if cells <= 32 else 64)]
        plen = min(cells, 64)
        start = (3 * (node_level + 1)) % 16
        px = tab[start][(ly + lz) % plen] - 1
        py = tab[(start + 1) % 16][(lx + lz) % plen] - 1
        pz = tab[(start + 2) % 16][(lx + ly) % plen] - 1
        perm_size = size / cells
        t = np.stack([gx * size + px * perm_size, gy * size + py * perm_size, gz * size + pz * perm_size], axis=1)
    d = pos - t
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    taken = np.zeros(len(keys), dtype=bool)
    starts = np.nonzero(np.r_[True, cell[1:] != cell[:-1]])[0]
    ends = np.r_[starts[1:], len(keys)]
    for a, b in zip(starts, ends):
        taken[a + int(np.argmin(d2[a:b]))] = True      # np.argmin returns the FIRST minimum
    return taken


@pytest.mark.parametrize("sampler", [O.GRID_CENTER, O.JITTERED])
@pytest.mark.parametrize("d", [250, 40])
def test_grid_center_and_jittered_against_an_independent_characterisation(sampler, d):
    rng = np.random.default_rng(31 + d)
    n = 60000
    xyz = np.vstack([rng.random((n - 5000, 3)), 0.4 + 0.01 * rng.standard_normal((5000, 3))]).clip(0.0, 1.0)
    xyz[::17] = xyz[5]                                   # exact duplicates: ties in distance
    sp = O.spacing_from_diagonal(*UNIT, d)
    keys, clamped = O.index_points(xyz, *UNIT)
    order = O.sort_by_key(keys)
    ks, pos = keys[order], clamped[order]
    # the root node (level -1) and one level-0 node
    cases = [(-1, np.arange(n))]
    in0 = np.nonzero((ks >> np.uint64(60)) == np.uint64(5))[0]
    cases.append((0, in0))
    for level, sel in cases:
        node_key = int(ks[sel[0]]) >> (3 * (20 - level)) << (3 * (20 - level)) if level >= 0 else 0
        cnt, k2, i2 = O.sample_points(sampler, 10, ks[sel], order[sel], clamped, node_key, level, *UNIT, sp, O.ALWAYS_ADHERE)
        assert cnt > 0
        want = _expected_grid_sample(ks[sel], pos[sel], sampler, level, sp)
        assert cnt == int(want.sum())
        assert np.array_equal(i2[:cnt], order[sel][want])          # taken points, in Morton order
        assert np.array_equal(i2[cnt:], order[sel][~want])         # the rest keeps its order (stable partition)


# ---------------------------------------------------------------------------------------------------------------------
# The same characterisation in bounds whose halving chain ROUNDS (the unit cube's box edges are dyadic: any evaluation
# order gives the same targets there).  The cell / node boxes are evaluated here in NumPy in the reference's operation
# order -- get_octant_bounds, OctreeAlgorithms.cpp:3-18, iterated by get_bounds_from_morton_index, OctreeAlgorithms.h:104-116:
# per level and axis  half = (max - min) / 2;  min' = bit ? min + half : min;  max' = min' + half  -- one chain per
# point, vectorised over the points, independent of oracle/oracle.cpp.
def _chain_bounds(keys, bmin, bmax, depth):
    n = len(keys)
    lo = np.tile(np.asarray(bmin, dtype=np.float64), (n, 1))
    hi = np.tile(np.asarray(bmax, dtype=np.float64), (n, 1))
    for level in range(depth):
        octant = (keys >> np.uint64(3 * (20 - level))) & np.uint64(7)
        half = (hi - lo) / 2.0
        for axis, bit in ((0, 4), (1, 2), (2, 1)):                       # octant = x << 2 | y << 1 | z
            up = (octant & np.uint64(bit)) != 0
            lo[:, axis] = np.where(up, lo[:, axis] + half[:, axis], lo[:, axis])
        hi = lo + half
    return lo, hi


def _prev_pow2(x):
    x = int(x)
    for s in (1, 2, 4, 8, 16):
        x |= x >> s
    return x - (x >> 1)


def _expected_grid_sample_in_bounds(keys, pos, sampler, node_level, spacing_at_root, bmin, bmax):
    """taken flags of a sampled node (all keys share its prefix) -- Sampling.h:314-416 (GRID_CENTER), :598-759 (JITTERED)"""
    s_node = float(np.float32(spacing_at_root)) / 2.0 ** (node_level + 1)              # float / pow(2, L + 1) in double
    ext_x_root = float(bmax[0]) - float(bmin[0])
    if sampler == O.GRID_CENTER:
        grid_level = max(-1, int(np.floor(np.log2(np.float32(ext_x_root / s_node)))) - 1)   # log2f of the narrowed ratio
        lo, hi = _chain_bounds(keys, bmin, bmax, grid_level + 1)
        t = lo + (hi - lo) / 2.0                                                            # AABB::getCenter
    else:
        nlo, nhi = _chain_bounds(keys[:1], bmin, bmax, node_level + 1)                     # the node's box
        ext_x = float(nhi[0, 0] - nlo[0, 0])
        cells = _prev_pow2(np.uint32(ext_x / s_node))
        assert cells >= 16
        levels = int(np.log2(cells))
        grid_level = node_level + levels
        rel = (keys >> np.uint64(3 * (20 - grid_level))) & np.uint64((1 << (3 * levels)) - 1)
        gx, gy, gz = (_compact3(rel >> np.uint64(2)).astype(np.int64), _compact3(rel >> np.uint64(1)).astype(np.int64),
                      _compact3(rel).astype(np.int64))
        tab = _jitter_tables()[16 if cells <= 16 else (32 if cells <= 32 else 64)]
        plen = min(cells, 64)
        start = (3 * (node_level + 1)) % 16
        px = tab[start][(gy + gz) % plen] - 1
        py = tab[(start + 1) % 16][(gx + gz) % plen] - 1
        pz = tab[(start + 2) % 16][(gx + gy) % plen] - 1
        cell_size = ext_x / cells                        # the x extent serves every axis (Sampling.h:655-668)
        perm_size = cell_size / cells
        g = np.stack([gx, gy, gz], axis=1).astype(np.float64)
        p = np.stack([px, py, pz], axis=1).astype(np.float64)
        t = nlo[0] + (g * cell_size + p * perm_size)
    cell = keys >> np.uint64(3 * (20 - grid_level)) if grid_level >= 0 else np.zeros_like(keys)
    d = pos - t
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    taken = np.zeros(len(keys), dtype=bool)
    starts = np.nonzero(np.r_[True, cell[1:] != cell[:-1]])[0]
    ends = np.r_[starts[1:], len(keys)]
    for a, b in zip(starts, ends):
        taken[a + int(np.argmin(d2[a:b]))] = True
    return taken


ROUNDING_BOUNDS = [
    ([-512.25, 1000.5, -3.125], [-512.25 + 777.7, 1000.5 + 777.7, -3.125 + 777.7]),      # cubic, chain rounds
    ([0.1, -7.3, 1e-3], [0.1 + 3.3, -7.3 + 2.9, 1e-3 + 3.7]),                            # three different extents
]


@pytest.mark.parametrize("sampler", [O.GRID_CENTER, O.JITTERED])
@pytest.mark.parametrize("bounds", ROUNDING_BOUNDS, ids=["cubic", "box"])
def test_grid_samplers_against_the_characterisation_in_bounds_whose_chain_rounds(sampler, bounds):
    bmin, bmax = bounds
    lo, hi = np.array(bmin), np.array(bmax)
    rng = np.random.default_rng(77)
    n = 50000
    u = np.vstack([rng.random((n - 4000, 3)), 0.37 + 0.01 * rng.standard_normal((4000, 3))]).clip(0.0, 1.0)
    xyz = lo + u * (hi - lo)
    xyz[::19] = xyz[7]
    sp = O.spacing_from_diagonal(bmin, bmax, 250)
    keys, clamped = O.index_points(xyz, bmin, bmax)
    order = O.sort_by_key(keys)
    ks, pos = keys[order], clamped[order]
    cases = [(-1, np.arange(n))]
    for octant in (2, 5):
        cases.append((0, np.nonzero((ks >> np.uint64(60)) == np.uint64(octant))[0]))
    deep = np.nonzero((ks >> np.uint64(57)) == (ks[n // 3] >> np.uint64(57)))[0]         # one level-1 node
    cases.append((1, deep))
    for level, sel in cases:
        if len(sel) == 0:
            continue
        node_key = int(ks[sel[0]]) >> (3 * (20 - level)) << (3 * (20 - level)) if level >= 0 else 0
        cnt, k2, i2 = O.sample_points(sampler, 10, ks[sel], order[sel], clamped, node_key, level, bmin, bmax, sp, O.ALWAYS_ADHERE)
        want = _expected_grid_sample_in_bounds(ks[sel], pos[sel], sampler, level, sp, bmin, bmax)
        assert cnt == int(want.sum())
        assert np.array_equal(i2[:cnt], order[sel][want])
        assert np.array_equal(i2[cnt:], order[sel][~want])
