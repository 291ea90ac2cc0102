"""MIN_DISTANCE decided on key coordinates (schwarzwald_amd/csrc/swz_mdkeys.hip) against the CPU oracle.

The sweep compares integer key coordinates and evaluates only the pairs inside the quantisation band around the spacing
on the exact positions; the accepted set must be the oracle's, point for point, whatever the scheduling, the record
size, the width of the band or the share of pairs that takes the exact path."""
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
    parts = []
    k = n // 4
    parts.append(np.column_stack([rng.random(k), rng.random(k), 0.3 + 0.001 * rng.standard_normal(k)]))
    parts.append(np.column_stack([rng.random(k), 0.7 + 0.0005 * rng.standard_normal(k), rng.random(k)]))
    parts.append(0.5 + 0.02 * rng.standard_normal((k, 3)))
    rest = n - 3 * k
    base = rng.random((max(rest // 8, 1), 3))
    parts.append(base[rng.integers(0, base.shape[0], rest)])  # exact duplicates
    return np.clip(np.vstack(parts), 0.0, 1.0)


def _check(ctx, xyz, bmin, bmax, d, mppn, options, **kw):
    import schwarzwald_amd as swz
    spacing = O.spacing_from_diagonal(bmin, bmax, d)
    o = O.tile(xyz, bmin, bmax, O.MIN_DISTANCE, mppn, spacing, **kw)
    assert o["status"] == 0
    try:
        for k, v in options.items():
            ctx.set_option(k, v)
        g = ctx.tile(xyz, bmin, bmax, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
    finally:
        for k in options:
            ctx.set_option(k, None)
    assert np.array_equal(g.keys, o["keys"]) and np.array_equal(g.perm, o["perm"])
    bad = np.flatnonzero(g.level != o["level"])
    assert bad.size == 0, "%d points differ, first at sorted position %d: level %d, oracle %d" % (
        bad.size, bad[0], g.level[bad[0]], o["level"][bad[0]])
    return g


MODES = [
    {},
    {"SWZ_MD_PATIENT": "0", "SWZ_MD_LAZY": "0"},
    {"SWZ_MD_PATIENT": "1", "SWZ_MD_LAZY": "1", "SWZ_MD_LAZY_FRAC": "0"},
    {"SWZ_MD_PATIENT": "1", "SWZ_MD_LAZY": "1", "SWZ_MD_LAZY_FRAC": "1", "SWZ_MD_BIG": "1"},
    {"SWZ_MD_PATIENT": "0", "SWZ_MD_LAZY": "1", "SWZ_MD_ABLATE": "8"},       # blocker scans without the dead-point test
    {"SWZ_MD_GRID": "3"},                                                    # 12 workgroups stride over every queue
    {"SWZ_MD_BIG": "1", "SWZ_MD_COARSEN": "2", "SWZ_MD_COARSEN_MIN": "0", "SWZ_MD_LAZY": "0"},  # cells of hundreds of points
    {"SWZ_MD_BIG": "0", "SWZ_MD_COARSEN": "2", "SWZ_MD_COARSEN_MIN": "0"},   # ... through the one-chunk build
    {"SWZ_MD_GROUPS": "1"},
    {"SWZ_MD_GROUPS": "3", "SWZ_MD_LAZY": "0"},
    {"SWZ_MD_KEYS_RG": "4", "SWZ_MD_COARSEN": "1", "SWZ_MD_COARSEN_MIN": "0"},  # 3 inline accepted points, most in the overflow
    {"SWZ_MD_KEYS_RG": "8", "SWZ_MD_BATCH": "1"},
    {"SWZ_MD_CHAIN": "1"},                                                   # no cell run by the wavefront that woke it
    {"SWZ_MD_CHAIN": "8", "SWZ_MD_LAZY": "0", "SWZ_MD_PATIENT": "0"},        # chains of up to 8 such cells
    {"SWZ_MD_CHAIN": "8", "SWZ_MD_GRID": "3"},
    {"SWZ_MD_CHAIN": "8", "SWZ_MD_BIG": "1", "SWZ_MD_COARSEN": "2", "SWZ_MD_COARSEN_MIN": "0", "SWZ_MD_LAZY": "0"},
    # dense cells: every cell of 64 / 200 points and more yields after an accept and is thinned by the reject passes
    {"SWZ_MD_DENSE_MIN": "64", "SWZ_MD_BIG": "1", "SWZ_MD_GROUPS": "1"},
    {"SWZ_MD_DENSE_MIN": "200", "SWZ_MD_BIG": "1", "SWZ_MD_GROUPS": "1", "SWZ_MD_COARSEN": "2", "SWZ_MD_COARSEN_MIN": "0", "SWZ_MD_LAZY": "0"},
    {"SWZ_MD_DENSE_MIN": "0", "SWZ_MD_BIG": "1", "SWZ_MD_GROUPS": "1"},
]


@pytest.mark.parametrize("mode", MODES, ids=lambda m: "-".join("%s%s" % (k[7:11], v) for k, v in m.items()) or "default")
def test_key_sweep_matches_oracle_under_every_scheduling(ctx, mode):
    rng = np.random.default_rng(99)
    xyz = np.vstack([rng.random((400000, 3)), 0.25 + 0.01 * rng.standard_normal((50000, 3))])
    xyz = np.clip(xyz, 0.0, 1.0)
    opts = dict(mode)
    opts["SWZ_MD_SPARSE_LIMIT"] = "0"  # every level on the sweep
    for d, mppn in ((250, 2000), (40, 500)):
        g = _check(ctx, xyz, *UNIT, d, mppn, opts)
    assert g.stats["min_distance_rounds"] > 0


@pytest.mark.parametrize("band", ["0", "40", "1e9"])
def test_key_sweep_band_sends_pairs_to_the_exact_compare(ctx, band):
    """Extra band width: 40 key cells send a few per cent of the near pairs to the exact compare on the original
    positions, 1e9 all of them (nothing is decided on keys but 'far beyond reach').  Bounds that are not dyadic, so
    that positions and key cells do not line up."""
    rng = np.random.default_rng(4711)
    side = 1.7320508
    bmin = np.array([-3.25, 10.125, 0.7])
    bmax = bmin + side
    xyz = bmin + rng.random((300000, 3)) * side * np.array([1.0, 0.8, 0.25])
    opts = {"SWZ_MD_SPARSE_LIMIT": "0", "SWZ_MD_KEYS_BAND": band}
    for d in (250, 90):
        _check(ctx, xyz, bmin.tolist(), bmax.tolist(), d, 1500, opts)


def test_key_sweep_needs_the_band(ctx):
    """The test above would not notice a sweep that trusts the keys too far unless such a sweep fails: with a NEGATIVE
    extra band (pairs within a cell of the spacing decided on keys) the result must differ from the oracle's
    somewhere in a cloud this size -- or the compare is not where the decisions are taken."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(5)
    xyz = rng.random((600000, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, 2000, spacing)
    try:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", "0")
        ctx.set_option("SWZ_MD_KEYS_BAND", "-1.7499")
        g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=spacing))
    finally:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", None)
        ctx.set_option("SWZ_MD_KEYS_BAND", None)
    assert not np.array_equal(g.level, o["level"])


def test_key_sweep_clustered_and_duplicates(ctx):
    rng = np.random.default_rng(12)
    xyz = _clustered(rng, 300000)
    for d, mppn in ((250, 1000), (120, 400)):
        _check(ctx, xyz, *UNIT, d, mppn, {"SWZ_MD_SPARSE_LIMIT": "0"})
        _check(ctx, xyz, *UNIT, d, mppn, {})  # sparse levels on their own path, dense ones on the key sweep


def test_key_sweep_dense_cells(ctx):
    """Blobs of tens of thousands of points inside single cells (and a sheet): dense cells yield after every accepted point
    and a workgroup per cell kills what the accepted points around it reject between the launches."""
    rng = np.random.default_rng(2048)
    xyz = np.vstack([rng.random((150000, 3)),
                     0.31 + 0.002 * rng.standard_normal((120000, 3)),
                     0.72 + 0.004 * rng.standard_normal((150000, 3)),
                     np.column_stack([0.1 + 0.05 * rng.random(80000), 0.1 + 0.05 * rng.random(80000), 0.555 + 1e-4 * rng.standard_normal(80000)])])
    xyz = np.clip(xyz, 0.0, 1.0)
    for opts in ({"SWZ_MD_SPARSE_LIMIT": "0"}, {"SWZ_MD_SPARSE_LIMIT": "0", "SWZ_MD_DENSE_MIN": "512", "SWZ_MD_CHAIN": "1"}, {}):
        g = _check(ctx, xyz, *UNIT, 250, 3000, opts)
    assert g.stats["max_level"] >= 5


def test_key_sweep_deep_levels_fall_back_to_positions(ctx):
    """A tight blob drives nodes so deep that the spacing spans fewer key cells than the band allows for: those levels
    run on gathered positions, the upper ones on keys, in one call."""
    rng = np.random.default_rng(77)
    xyz = np.vstack([rng.random((100000, 3)), 0.4321 + 2e-5 * rng.standard_normal((200000, 3))])
    xyz = np.clip(xyz, 0.0, 1.0)
    g = _check(ctx, xyz, *UNIT, 250, 300, {"SWZ_MD_SPARSE_LIMIT": "0"})
    assert g.stats["max_level"] >= 8
    _check(ctx, xyz, *UNIT, 250, 300, {"SWZ_MD_SPARSE_LIMIT": "0", "SWZ_MD_KEYS_MIN_CELLS": "2000"})  # keys on two levels only


def test_key_sweep_off_gives_the_same_result(ctx):
    rng = np.random.default_rng(3)
    xyz = rng.random((250000, 3))
    _check(ctx, xyz, *UNIT, 250, 1000, {"SWZ_MD_SPARSE_LIMIT": "0", "SWZ_MD_KEYS": "0"})


def test_key_sweep_fast_strategy(ctx):
    """FAST: start below the root, reconstruct the skipped levels with AlwaysAdhereToMinSpacing from a gathered subset
    (the active set there is not the sorted array itself)."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(400008)
    xyz = rng.random((400000, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, 2000, spacing, strategy=O.FAST, fast_concurrency=2)
    try:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", "0")
        g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=spacing,
                                                strategy=swz.FAST, fast_concurrency=2))
    finally:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", None)
    assert np.array_equal(g.level, o["level"]) and np.array_equal(g.dup, o["dup"])


@pytest.mark.parametrize("band", ["0", "25", "1e9"])
def test_sparse_levels_on_keys(ctx, band):
    """The thread-per-point path of the sparse levels takes its records from the keys as well; forced onto every level
    here, with the band widened so that a large share of its compares goes to the exact positions."""
    rng = np.random.default_rng(815)
    side = 2.5
    bmin = np.array([100.0, -7.75, 0.3])
    bmax = bmin + side
    xyz = bmin + rng.random((400000, 3)) * side * np.array([1.0, 0.7, 0.3])
    opts = {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_MD_KEYS_BAND": band}
    for d in (250, 90):
        _check(ctx, xyz, bmin.tolist(), bmax.tolist(), d, 2000, opts)


def test_no_position_gather_on_keys(ctx):
    """With cubic bounds exact MIN_DISTANCE never brings the positions into Morton order: the profile of a call has no
    gather_positions entry (and has one when the key path is switched off)."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(1)
    xyz = rng.random((200000, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    p = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=spacing)
    ctx.profile_enable(True)
    try:
        ctx.profile_reset()
        ctx.tile(xyz, *UNIT, p)
        on_keys = ctx.profile_get()
        ctx.set_option("SWZ_MD_KEYS", "0")
        ctx.profile_reset()
        ctx.tile(xyz, *UNIT, p)
        on_positions = ctx.profile_get()
    finally:
        ctx.set_option("SWZ_MD_KEYS", None)
        ctx.profile_enable(False)
    assert "gather_positions" not in on_keys
    assert "gather_positions" in on_positions
