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
