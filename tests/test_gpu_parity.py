"""GPU parity tests: the HIP path, called through the C ABI (include/swz_gpu.h), against the CPU oracle
on the same seeded inputs.  Integer/byte/index work must be bit-exact; the only floating-point output is
the in-place clamp, also compared bit-exactly."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])


@pytest.fixture(scope="module")
def ctx():
    import schwarzwald_amd as swz
    c = swz.Context(0)
    yield c
    c.close()


def _clustered(rng, n):
    """Surface-like clustered points (real LAS is far from uniform): a few planes + blobs + duplicates."""
    parts = []
    k = n // 4
    parts.append(np.column_stack([rng.random(k), rng.random(k), 0.3 + 0.001 * rng.standard_normal(k)]))
    parts.append(np.column_stack([rng.random(k), 0.7 + 0.0005 * rng.standard_normal(k), rng.random(k)]))
    parts.append(0.5 + 0.02 * rng.standard_normal((k, 3)))
    rest = n - 3 * k
    base = rng.random((max(rest // 8, 1), 3))
    parts.append(base[rng.integers(0, base.shape[0], rest)])  # exact duplicates
    return np.clip(np.vstack(parts), 0.0, 1.0)


# ----------------------------------------------------------------------------------------- encode
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 100003])
def test_morton_encode_matches_oracle(ctx, n):
    rng = np.random.default_rng(n + 1)
    xyz = rng.random((n, 3)) * 3.0 - 1.0  # a third of the points are outliers that must be clamped
    bmin, bmax = [-0.25, 0.0, 0.1], [1.25, 1.0, 0.9]
    keys, clamped = ctx.morton_encode(xyz, bmin, bmax)
    okeys, oclamped = O.index_points(xyz, bmin, bmax)
    assert np.array_equal(keys, okeys)
    assert np.array_equal(clamped.view(np.uint64), oclamped.view(np.uint64))


def test_morton_encode_known_answers(ctx):
    pts = np.array([[0, 0, 0], [1, 1, 1], [.5, .5, .5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.3, 0.6, 0.9],
                    [0.999999999, 1e-9, 0.25]], dtype=np.float64)
    exp = [0x0, 0x7FFFFFFFFFFFFFFF, 0x7000000000000000, 0x4924924924924924, 0x2492492492492492,
           0x1249249249249249, 0x3A56A56A56A56A56, 0x4B24924924924924]
    keys, _ = ctx.morton_encode(pts, *UNIT)
    assert [int(k) for k in keys] == exp
    mn = [-512.25, 1000.5, -3.125]
    mx = [v + 777.7 for v in mn]
    keys, _ = ctx.morton_encode(np.array([[-100, 1500.123456789, 400]]), mn, mx)
    assert int(keys[0]) == 0x7080F255F85E7F48


def test_morton_encode_reference_lattice(ctx):
    """test/TestOctreeIndexing.cpp:102-130 with 21 levels: every octant digit of the key is exact."""
    b = ([0, 0, 0], [1 << 21] * 3)
    rng = np.random.default_rng(3)
    cells = rng.integers(0, 1 << 21, size=(5000, 3))
    xyz = cells.astype(np.float64) + 0.5
    keys, _ = ctx.morton_encode(xyz, *b)
    for c, k in zip(cells[:200], keys[:200]):
        x, y, z = (int(v) for v in c)
        exp = 0
        for bit in range(21):
            exp |= ((z >> bit) & 1) << (3 * bit) | ((y >> bit) & 1) << (3 * bit + 1) | ((x >> bit) & 1) << (3 * bit + 2)
        assert int(k) == exp
    assert np.array_equal(keys, O.index_points(xyz, *b)[0])


# ----------------------------------------------------------------------------------------- sort
@pytest.mark.parametrize("n", [1, 2, 4095, 4096, 4097, 300001])
def test_sort_matches_oracle(ctx, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    perm, ks = ctx.sort_by_key(keys)
    assert np.array_equal(perm, O.sort_by_key(keys))
    assert np.array_equal(ks, keys[perm])


def test_sort_is_stable_on_ties(ctx):
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 50, size=200000, dtype=np.uint64) << np.uint64(40)  # few distinct keys, high digits
    keys[::7] = 0
    keys[1::13] = (1 << 63) - 1
    perm, ks = ctx.sort_by_key(keys)
    assert np.array_equal(perm, O.sort_by_key(keys))
    assert np.all(np.diff(ks.astype(np.float64)) >= 0)


@pytest.mark.parametrize("opts", [
    {"SWZ_SORT_HYBRID_TOP": "4"},                                                     # top 32 bits, then the run pass
    {"SWZ_SORT_HYBRID_TOP": "6"},
    {"SWZ_SORT_HYBRID_TOP": "2", "SWZ_SORT_FIX_SHORT": "3", "SWZ_SORT_FIX_LONG": "4096"},   # most runs go to workgroups
    {"SWZ_SORT_HYBRID_TOP": "4", "SWZ_SORT_FIX_SHORT": "2", "SWZ_SORT_FIX_LONG": "5"},      # runs too long: eight passes after all
    {},                                                                               # K from the sample of the keys
], ids=lambda o: "-".join(v for v in o.values()) or "sampled")
def test_sort_hybrid_top_digits_then_runs(ctx, opts):
    """Large inputs are sorted by LSD passes over the top digits only and one pass that orders the runs of equal top
    bits (swz_sort.hip); forced here onto small inputs, with run-length limits that send runs through the per-element
    ranking, the workgroup ranking and the fallback to eight passes.  Order and tie order must be the oracle's."""
    rng = np.random.default_rng(2024)
    n = 300_000
    cases = {
        "random": rng.integers(0, 1 << 63, size=n, dtype=np.uint64),
        # clusters: few distinct top bits, random low bits, plus exact duplicates (ties by index)
        "clustered": (rng.integers(0, 40, size=n, dtype=np.uint64) << np.uint64(45)) | rng.integers(0, 1 << 20, size=n, dtype=np.uint64),
        "one_run": rng.integers(0, 1 << 12, size=n, dtype=np.uint64),   # all top bits equal: one run of n
    }
    cases["clustered"][::5] = cases["clustered"][0]
    try:
        ctx.set_option("SWZ_SORT_HYBRID_MIN_N", "1")
        for k, v in opts.items():
            ctx.set_option(k, v)
        for name, keys in cases.items():
            perm, ks = ctx.sort_by_key(keys)
            assert np.array_equal(perm, O.sort_by_key(keys)), name
            assert np.array_equal(ks, keys[perm]), name
    finally:
        for k in list(opts) + ["SWZ_SORT_HYBRID_MIN_N"]:
            ctx.set_option(k, None)


@pytest.mark.parametrize("kind", ["uniform", "clusters", "duplicates", "one_run"])
def test_sort_large_inputs_choose_their_passes(ctx, kind):
    """Above 2^24 keys the sort decides by itself (from a sorted sample of the keys) how many top digits to pass over
    before the run pass, and falls back to eight passes when a run of equal top bits is too long: uniform Morton-like
    keys (four passes), keys in a few thousand clusters (six), 30 % exact duplicates of one key and all keys equal in
    their top 40 bits (both: long runs, eight passes after all).  Checked against NumPy's stable argsort."""
    rng = np.random.default_rng(77)
    n = (1 << 24) + 12345
    if kind == "uniform":
        keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    elif kind == "clusters":
        centres = rng.integers(0, 1 << 63, size=4096, dtype=np.uint64) & ~np.uint64((1 << 40) - 1)
        keys = centres[rng.integers(0, 4096, size=n)] | rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    elif kind == "duplicates":
        keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
        keys[rng.random(n) < 0.3] = keys[0]
    else:
        keys = (np.uint64(0x1234567) << np.uint64(36)) | rng.integers(0, 1 << 23, size=n, dtype=np.uint64)
    perm, ks = ctx.sort_by_key(keys)
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(perm, want)
    assert np.array_equal(ks, keys[want])


# ----------------------------------------------------------------------------------------- sample_points
def _sorted_cloud(xyz, bmin, bmax):
    keys, clamped = O.index_points(xyz, bmin, bmax)
    order = O.sort_by_key(keys)
    return keys[order], order, clamped


def _flags_from_oracle(sampler, max_pts, ks, order, xyz, node_level, bmin, bmax, spacing, behaviour):
    taken, k2, i2 = O.sample_points(sampler, max_pts, ks, order, xyz, 0, node_level, bmin, bmax, spacing, behaviour)
    if taken < 0:
        return taken, None
    flags = np.zeros(len(order), dtype=np.uint8)
    pos = {int(v): j for j, v in enumerate(order)}
    for v in i2[:taken]:
        flags[pos[int(v)]] = 1
    return taken, flags


def test_random_grid_reference_known_answer(ctx):
    """test/TestOctreeIndexing.cpp:169-252 (32^3 lattice, spacing 32, node level 0) on 21-level keys."""
    side = 32
    g = np.arange(side, dtype=np.float64) + 0.5
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    bmin, bmax = [0, 0, 0], [side] * 3
    ks, order, _ = _sorted_cloud(xyz, bmin, bmax)
    import schwarzwald_amd as swz
    flags = ctx.sample_points(swz.RANDOM_GRID, 16, ks, order, xyz, 0, 0, bmin, bmax, float(side))
    assert int(flags.sum()) == 8
    expected = [[0.5, 0.5, 0.5], [0.5, 0.5, 16.5], [0.5, 16.5, 0.5], [0.5, 16.5, 16.5],
                [16.5, 0.5, 0.5], [16.5, 0.5, 16.5], [16.5, 16.5, 0.5], [16.5, 16.5, 16.5]]
    assert xyz[order[flags == 1]].tolist() == expected


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
@pytest.mark.parametrize("kind", ["uniform", "clustered"])
@pytest.mark.parametrize("spacing,behaviour", [(0.05, O.ALWAYS_ADHERE), (0.011, O.ALWAYS_ADHERE),
                                               (0.03, O.TAKE_ALL_WHEN_BELOW_MAX)])
def test_sample_points_root_matches_oracle(ctx, sampler, kind, spacing, behaviour):
    rng = np.random.default_rng(17)
    n = 60000
    xyz = rng.random((n, 3)) if kind == "uniform" else _clustered(rng, n)
    ks, order, clamped = _sorted_cloud(xyz, *UNIT)
    taken, oflags = _flags_from_oracle(sampler, 1000, ks, order, clamped, -1, *UNIT, spacing, behaviour)
    flags = ctx.sample_points(sampler, 1000, ks, order, clamped, 0, -1, *UNIT, spacing, behaviour)
    assert int(flags.sum()) == taken
    assert np.array_equal(flags, oflags)


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
def test_sample_points_inner_node_matches_oracle(ctx, sampler):
    """A node at level 1 (two octants below the root): node-relative spacing, bounds and jitter tables."""
    rng = np.random.default_rng(23)
    lo, hi = np.array([0.25, 0.5, 0.75]), np.array([0.5, 0.75, 1.0])
    xyz = lo + rng.random((40000, 3)) * (hi - lo) * 0.999999
    ks, order, clamped = _sorted_cloud(xyz, *UNIT)
    node_key = int(ks[0]) >> (3 * 19) << (3 * 19)
    assert np.all((ks >> np.uint64(3 * 19)) == (ks[0] >> np.uint64(3 * 19)))
    taken, k2, i2 = O.sample_points(sampler, 100, ks, order, clamped, node_key, 1, *UNIT, 0.04, O.ALWAYS_ADHERE)
    assert taken > 0
    oflags = np.zeros(len(order), dtype=np.uint8)
    pos = {int(v): j for j, v in enumerate(order)}
    for v in i2[:taken]:
        oflags[pos[int(v)]] = 1
    flags = ctx.sample_points(sampler, 100, ks, order, clamped, node_key, 1, *UNIT, 0.04, O.ALWAYS_ADHERE)
    assert np.array_equal(flags, oflags)


def test_jittered_small_grid_is_an_error(ctx):
    import schwarzwald_amd as swz
    rng = np.random.default_rng(5)
    xyz = rng.random((5000, 3))
    ks, order, clamped = _sorted_cloud(xyz, *UNIT)
    taken, _, _ = O.sample_points(O.JITTERED, 10, ks, order, clamped, 0, -1, *UNIT, 0.2, O.ALWAYS_ADHERE)
    assert taken == O.ERR_JITTER_GRID_TOO_SMALL
    with pytest.raises(swz.SwzError) as e:
        ctx.sample_points(swz.JITTERED, 10, ks, order, clamped, 0, -1, *UNIT, 0.2, swz.ALWAYS_ADHERE_TO_MIN_SPACING)
    assert e.value.code == 3


# ----------------------------------------------------------------------------------------- whole tiler
def _tile_both(ctx, xyz, bmin, bmax, sampler, max_pts, spacing, max_depth=100):
    import schwarzwald_amd as swz
    o = O.tile(xyz, bmin, bmax, sampler, max_pts, spacing, max_depth=max_depth)
    p = swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing, max_depth=max_depth)
    g = ctx.tile(xyz, bmin, bmax, p)
    return o, g


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
@pytest.mark.parametrize("kind,n,max_pts,d", [("uniform", 200000, 2000, 250), ("uniform", 65536, 64, 32),
                                               ("clustered", 150000, 500, 250)])
def test_tile_accurate_matches_oracle(ctx, sampler, kind, n, max_pts, d):
    rng = np.random.default_rng(n + d)
    xyz = rng.random((n, 3)) if kind == "uniform" else _clustered(rng, n)
    spacing = O.spacing_from_diagonal(*UNIT, d)
    o, g = _tile_both(ctx, xyz, *UNIT, sampler, max_pts, spacing)
    assert o["status"] == 0
    assert np.array_equal(g.keys, o["keys"])
    assert np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])
    assert g.stats["num_nodes"] == o["stats"]["num_nodes"]
    assert g.stats["points_visited"] == o["stats"]["points_visited"]
    assert g.stats["max_level"] == o["stats"]["max_level"]


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
@pytest.mark.parametrize("n", [70001, 131073])
def test_tile_odd_counts_and_arbitrary_bounds(ctx, sampler, n):
    """Odd point counts (the y and z columns of the sorted positions then start 8 bytes off a 16-byte boundary, the
    last thread of the grid samplers holds one point) in root bounds whose halving chain rounds at every level."""
    rng = np.random.default_rng(n)
    bmin = [512345.678, -1234.5678901, 98.7654321]
    side = 123.456789
    bmax = [bmin[0] + side, bmin[1] + side, bmin[2] + side]
    xyz = np.array(bmin) + rng.random((n, 3)) * side
    spacing = O.spacing_from_diagonal(bmin, bmax, 100)
    o, g = _tile_both(ctx, xyz, bmin, bmax, sampler, 700, spacing)
    assert o["status"] == 0
    assert np.array_equal(g.keys, o["keys"])
    assert np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])
    assert g.stats["num_nodes"] == o["stats"]["num_nodes"]


@pytest.mark.parametrize("name", ["md_gridmap", "md_cell128", "sp_table", "md_pos"])
def test_min_distance_out_of_memory_gives_back_the_scratch_of_earlier_levels(ctx, name):
    """When the device runs out of memory, the workspace frees the MIN_DISTANCE buffers no level has asked for since an
    earlier level (or call) and allocates again (SWZ_FAIL_ALLOC pretends the first attempt failed): same results."""
    rng = np.random.default_rng(77)
    clouds = [rng.random((150000, 3)), _clustered(rng, 120000)]
    spacing = O.spacing_from_diagonal(*UNIT, 120)
    try:
        ctx.set_option("SWZ_FAIL_ALLOC", name)
        for xyz in clouds:  # the second call meets the buffers of the first
            o, g = _tile_both(ctx, xyz, *UNIT, O.MIN_DISTANCE, 800, spacing)
            assert o["status"] == 0
            assert np.array_equal(g.perm, o["perm"])
            assert np.array_equal(g.level, o["level"])
    finally:
        ctx.set_option("SWZ_FAIL_ALLOC", None)


@pytest.mark.parametrize("depth", [0, 2, 6])
@pytest.mark.parametrize("sampler", [O.GRID_CENTER, O.JITTERED])
def test_grid_samplers_box_table_depths(ctx, sampler, depth):
    """The first steps of the cell bounds chain come from a table of boxes (its depth normally follows the level's
    point count): none, a shallow and the deepest table, in bounds whose chain rounds."""
    rng = np.random.default_rng(31 + depth)
    bmin = [-7.123456789, 100.000001, 3.3333333]
    side = 17.71717171
    bmax = [b + side for b in bmin]
    xyz = np.array(bmin) + rng.random((180001, 3)) * side
    spacing = O.spacing_from_diagonal(bmin, bmax, 180)
    try:
        ctx.set_option("SWZ_GRID_TABLE_DEPTH", depth)
        ctx.set_option("SWZ_JITTER_TABLE", 0)  # JITTERED's own per-node table (what every other test runs) off: box table + chain
        o, g = _tile_both(ctx, xyz, bmin, bmax, sampler, 900, spacing)
    finally:
        ctx.set_option("SWZ_GRID_TABLE_DEPTH", None)
        ctx.set_option("SWZ_JITTER_TABLE", None)
    assert o["status"] == 0
    assert np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])


def test_jittered_fine_grid_uses_the_wide_index_path(ctx):
    """Spacing = diagonal / 4000: the root's jitter grid has 2048 cells a side (11 levels, 33 bits of cell index: the
    64-bit bit tricks); a tight blob puts thousands of points into single cells so that the targets decide."""
    rng = np.random.default_rng(2048)
    xyz = np.vstack([rng.random((50000, 3)), 0.4321 + 0.0004 * rng.standard_normal((100001, 3))])
    spacing = O.spacing_from_diagonal(*UNIT, 4000)
    o, g = _tile_both(ctx, xyz, *UNIT, O.JITTERED, 1000, spacing)
    assert o["status"] == 0
    assert o["stats"]["max_level"] >= 1
    assert np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])


def test_tile_more_nodes_than_the_node_kernels_have_threads(ctx):
    """A spacing of half the diagonal lets every node keep one point only: 1.5 M points make a full octree whose level 6
    has more than 2048 x 256 nodes (the per-node kernels stride over them)."""
    rng = np.random.default_rng(99)
    xyz = rng.random((1500000, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 2)
    o, g = _tile_both(ctx, xyz, *UNIT, O.RANDOM_GRID, 1, spacing)
    assert o["status"] == 0
    assert o["stats"]["num_nodes"] > 2 * 2048 * 256
    assert np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])
    assert g.stats["num_nodes"] == o["stats"]["num_nodes"]


def test_tile_max_depth_makes_terminal_nodes(ctx):
    rng = np.random.default_rng(8)
    xyz = rng.random((100000, 3))
    spacing = 0.2  # coarse: 64 grid cells at the root, so most points reach the terminal level
    for sampler in (O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE):
        o, g = _tile_both(ctx, xyz, *UNIT, sampler, 100, spacing, max_depth=2)
        assert o["status"] == 0 and o["stats"]["max_level"] == 2
        assert np.array_equal(g.level, o["level"])


def test_tile_outliers_and_small_inputs(ctx):
    rng = np.random.default_rng(4)
    for n in (1, 2, 63, 1000):
        xyz = rng.random((n, 3)) * 1.4 - 0.2
        o, g = _tile_both(ctx, xyz, *UNIT, O.GRID_CENTER, 10, 0.05)
        assert np.array_equal(g.keys, o["keys"]) and np.array_equal(g.perm, o["perm"])
        assert np.array_equal(g.level, o["level"])
        assert np.array_equal(g.xyz_clamped.view(np.uint64), o["xyz_clamped"].view(np.uint64))


def test_node_lists_group_points_by_node(ctx):
    rng = np.random.default_rng(6)
    xyz = rng.random((50000, 3))
    import schwarzwald_amd as swz
    p = swz.TileParams(sampler=swz.RANDOM_GRID, max_points_per_node=500,
                       spacing_at_root=O.spacing_from_diagonal(*UNIT, 250))
    g = ctx.tile(xyz, *UNIT, p)
    order, nodes = ctx.build_node_lists(g.keys, g.level)
    assert sorted(order.tolist()) == list(range(50000))
    assert len(nodes["level"]) == g.stats["num_nodes"]
    for lvl, key, off, cnt in zip(nodes["level"], nodes["key"], nodes["offset"], nodes["count"]):
        members = order[int(off):int(off + cnt)]
        assert np.all(g.level[members] == lvl)
        sh = 63 if lvl < 0 else 3 * (20 - int(lvl))
        assert np.all((g.keys[members] >> np.uint64(sh)) == (np.uint64(key) >> np.uint64(sh)))
        assert np.all(np.diff(members.astype(np.int64)) > 0)  # Morton order inside the node


# ----------------------------------------------------------------------------------------- FAST (TilingAlgorithmV3)
@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
@pytest.mark.parametrize("n,concurrency,max_pts", [(400000, 2, 2000), (150000, 8, 300)])
def test_tile_fast_matches_oracle(ctx, sampler, n, concurrency, max_pts):
    """FAST = start below the root (level chosen from the point distribution and the thread count), then
    reconstruct the skipped levels from the children's samples; dup marks the duplicated points."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(n + concurrency)
    xyz = rng.random((n, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    o = O.tile(xyz, *UNIT, sampler, max_pts, spacing, strategy=O.FAST, fast_concurrency=concurrency)
    assert o["status"] == 0
    p = swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing, strategy=swz.FAST,
                       fast_concurrency=concurrency)
    g = ctx.tile(xyz, *UNIT, p)
    assert g.stats["fast_start_levels"] == o["stats"]["fast_start_levels"]
    assert np.array_equal(g.keys, o["keys"]) and np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])
    assert np.array_equal(g.dup, o["dup"])
    assert g.stats["num_nodes"] == o["stats"]["num_nodes"]
    assert g.level.min() >= o["stats"]["fast_start_levels"] - 1  # nothing is persisted above the start level


def test_tile_fast_start_level_depends_on_concurrency(ctx):
    import schwarzwald_amd as swz
    rng = np.random.default_rng(77)
    xyz = rng.random((1_000_000, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    seen = set()
    for conc in (1, 4, 64):
        o = O.tile(xyz, *UNIT, O.RANDOM_GRID, 20000, spacing, strategy=O.FAST, fast_concurrency=conc)
        p = swz.TileParams(sampler=swz.RANDOM_GRID, max_points_per_node=20000, spacing_at_root=spacing,
                           strategy=swz.FAST, fast_concurrency=conc)
        g = ctx.tile(xyz, *UNIT, p)
        assert g.stats["fast_start_levels"] == o["stats"]["fast_start_levels"]
        assert np.array_equal(g.level, o["level"]) and np.array_equal(g.dup, o["dup"])
        seen.add(g.stats["fast_start_levels"])
    assert len(seen) > 1


def test_min_distance_sparse_path_and_its_fallback(ctx, monkeypatch):
    """Sparse levels use a thread-per-point fixpoint; when the data is locally dense it gives up and the frontier
    sweep takes over.  Both must give the oracle's set (SWZ_MD_SPARSE_LIMIT forces the sparse path onto denser data)."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(321)
    xyz = np.vstack([rng.random((120000, 3)), 0.5 + 0.004 * rng.standard_normal((30000, 3))])  # a dense blob inside
    xyz = np.clip(xyz, 0.0, 1.0)
    try:
        for limit, d in (("1000", 250), ("1000", 60), ("0", 250)):
            ctx.set_option("SWZ_MD_SPARSE_LIMIT", limit)
            spacing = O.spacing_from_diagonal(*UNIT, d)
            o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, 300, spacing)
            g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=300, spacing_at_root=spacing))
            assert np.array_equal(g.level, o["level"])
    finally:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", None)


@pytest.mark.gpu
@pytest.mark.parametrize("eps", ["1e30", "0.3", "1e-9"])
def test_min_distance_sparse_path_float_filter(ctx, eps):
    """The sparse path compares float copies of the positions first and repeats every compare inside the error band
    in double on the original positions.  Forced here: a band that holds every compare (all exact), a wide band, and
    -- to show that the test would notice a filter that decides too much -- a band far below the float error, which
    must NOT be relied on to give the oracle's set (it only has to run)."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(4711)
    xyz = rng.random((400000, 3)) * np.array([1.0, 0.7, 0.3]) + 0.123456789  # not dyadic: float copies do round
    bmin, bmax = np.zeros(3), np.full(3, 1.5)
    try:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", "1000")  # the sparse path on every level
        ctx.set_option("SWZ_SP_FILTER_EPS", eps)
        for d in (250, 90):
            spacing = O.spacing_from_diagonal(bmin, bmax, d)
            o = O.tile(xyz, bmin, bmax, O.MIN_DISTANCE, 2000, spacing)
            g = ctx.tile(xyz, bmin, bmax, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=spacing))
            if eps != "1e-9":
                assert np.array_equal(g.level, o["level"])
    finally:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", None)
        ctx.set_option("SWZ_SP_FILTER_EPS", None)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [
    {"SWZ_MD_PATIENT": "0", "SWZ_MD_LAZY": "0", "SWZ_MD_LATEST_FIRST": "0"},
    {"SWZ_MD_PATIENT": "0", "SWZ_MD_LAZY": "0", "SWZ_MD_LATEST_FIRST": "1", "SWZ_MD_BIG": "1"},
    {"SWZ_MD_PATIENT": "1", "SWZ_MD_LAZY": "1", "SWZ_MD_LATEST_FIRST": "1"},
    {"SWZ_MD_PATIENT": "1", "SWZ_MD_LAZY": "1", "SWZ_MD_LAZY_FRAC": "0", "SWZ_MD_LATEST_FIRST": "0", "SWZ_MD_BIG": "0"},
    {"SWZ_MD_PATIENT": "1", "SWZ_MD_LAZY": "1", "SWZ_MD_LAZY_FRAC": "1", "SWZ_MD_BIG": "1"},
    {"SWZ_MD_PATIENT": "0", "SWZ_MD_LAZY": "1", "SWZ_MD_ABLATE": "8"},  # 8: no dead-point test
    {"SWZ_MD_NBR_GRID": "3", "SWZ_MD_GRID": "40"},  # tiny launch grids: the grid-stride loops must cover every cell
    {"SWZ_MD_PERSISTENT": "1", "SWZ_MD_ROUNDS_PER_LAUNCH": "7"},  # the rounds inside persistent launches of 7 rounds
    # the build for cells of hundreds of points (four chunks per round trip, fast-forward) on cells that large
    {"SWZ_MD_BIG": "1", "SWZ_MD_COARSEN": "2", "SWZ_MD_COARSEN_MIN": "0", "SWZ_MD_LAZY": "0", "SWZ_MD_FF_MIN": "64"},
    {"SWZ_MD_BIG": "0", "SWZ_MD_COARSEN": "2", "SWZ_MD_COARSEN_MIN": "0"},  # ... and the small-cell build on them
    {"SWZ_MD_GROUPS": "1"},                          # all nodes of a level in one set of rounds
    {"SWZ_MD_GROUPS": "3", "SWZ_MD_LAZY": "0"},      # ... dealt to three sets on three streams
], ids=lambda m: "-".join("%s%s" % (k[7:10], v) for k, v in m.items()))
def test_min_distance_sweep_scheduling_modes(ctx, monkeypatch, mode):
    """The frontier sweep picks its scheduling per level from the cell statistics (patient stalls, lazy start, scan
    order, the kernel build for large or small cells); the big-level choices never trigger on test-sized inputs, so
    they are forced here.
    Scheduling must never change the result."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(99)
    xyz = np.vstack([rng.random((400000, 3)), 0.25 + 0.01 * rng.standard_normal((50000, 3))])
    xyz = np.clip(xyz, 0.0, 1.0)
    try:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", "0")  # keep every level on the sweep
        for k, v in mode.items():
            ctx.set_option(k, v)
        for d, mppn in ((250, 2000), (40, 500)):
            spacing = O.spacing_from_diagonal(*UNIT, d)
            o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, mppn, spacing)
            g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
            assert np.array_equal(g.keys, o["keys"]) and np.array_equal(g.perm, o["perm"])
            assert np.array_equal(g.level, o["level"])
    finally:
        for k in list(mode) + ["SWZ_MD_SPARSE_LIMIT"]:
            ctx.set_option(k, None)


@pytest.mark.parametrize("presort", [False, True])
def test_shard_api_with_ghosts_matches_oracle(presort):
    """Two shards (octants 0-3 / 4-7) driven by hand through swz_shard_*: the upper shard receives the lower
    shard's root samples as ghosts, either through swz_shard_begin_device alone or after swz_shard_presort_device."""
    import torch
    import schwarzwald_amd as swz
    rng = np.random.default_rng(808)
    n = 300_000
    xyz = rng.random((n, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, 2000, spacing)
    keys = O.index_points(xyz, *UNIT)[0]
    upper = (keys >> np.uint64(62)) & np.uint64(1) == 1           # top octant bit: x >= 0.5
    dev = torch.device("cuda:0")
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=spacing)
    ghosts = None
    got_level = np.empty(n, dtype=np.int8)
    for part in (~upper, upper):
        ids = np.nonzero(part)[0]
        ctx = swz.Context(0)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        loc = torch.from_numpy(xyz[ids]).to(dev)
        m = len(ids)
        g = 0 if ghosts is None else ghosts.shape[0]
        if presort:
            ctx.shard_presort_device(loc.data_ptr(), m, *UNIT, params, 100_000)
        taken = ctx.shard_begin_device(loc.data_ptr(), m, *UNIT, params, n, ghosts.data_ptr() if g else None, g)
        mine = torch.empty((taken, 3), dtype=torch.float64, device=dev)
        ctx.shard_root_taken_device(mine.data_ptr())
        k = torch.empty(m, dtype=torch.int64, device=dev)
        p = torch.empty(m, dtype=torch.int32, device=dev)
        lv = torch.empty(m, dtype=torch.int8, device=dev)
        ctx.shard_finish_device(k.data_ptr(), p.data_ptr(), lv.data_ptr())
        torch.cuda.synchronize()
        got_level[ids[p.cpu().numpy()]] = lv.cpu().numpy()
        assert np.array_equal(k.cpu().numpy().view(np.uint64), np.sort(keys[ids]))
        ghosts = mine
        ctx.close()
    want = np.empty(n, dtype=np.int8)
    want[o["perm"]] = o["level"]
    assert np.array_equal(got_level, want)


def test_sample_points_checks_node_key(ctx):
    """MIN_DISTANCE / JITTERED take the node's box from node_key: a range outside that node is refused instead of
    silently sampled as several pseudo-nodes; RANDOM_GRID / GRID_CENTER ignore node_key like the
    reference (its own test hands them a range spanning all octants, test/TestOctreeIndexing.cpp:169-252)."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(2)
    xyz = rng.random((5000, 3))
    ks, order, _ = _sorted_cloud(xyz, *UNIT)
    with pytest.raises(swz.SwzError) as e:
        ctx.sample_points(swz.MIN_DISTANCE, 10, ks, order, xyz, 0, 0, *UNIT, 0.05)
    assert e.value.code == 2 and "node_key" in str(e.value)
    inside = (ks >> np.uint64(60)) == 3
    node_key = 3 << 60
    got = ctx.sample_points(swz.MIN_DISTANCE, 10, ks[inside], order[inside], xyz, node_key, 0, *UNIT, 0.05)
    cnt, k2, i2 = O.sample_points(O.MIN_DISTANCE, 10, ks[inside], order[inside], xyz, node_key, 0, *UNIT, 0.05)
    assert int(got.sum()) == cnt and np.array_equal(order[inside][got == 1], i2[:cnt])
    # the whole cloud as ONE node at level 0 for the grid samplers: count and candidate level only
    for sampler in (swz.RANDOM_GRID, swz.GRID_CENTER):
        got = ctx.sample_points(sampler, 10, ks, order, xyz, 0, 0, *UNIT, 0.05)
        cnt, k2, i2 = O.sample_points(sampler, 10, ks, order, xyz, 0, 0, *UNIT, 0.05)
        assert int(got.sum()) == cnt and np.array_equal(order[got == 1], i2[:cnt])
