"""The sparse MIN_DISTANCE levels by blocks of cells staged in LDS (swz_mdblock.hip, round 6): the exact greedy set
(PoissonDiskSampling / SparseGrid::add, core/tiling/Sampling.h:421-471, core/datastructures/SparseGrid.cpp:116-146) whatever
the kernel's format, capacities, cell size or fall-back -- every variant against the oracle, point for point."""
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


def _cloud(seed, n=400000, blob=40000):
    rng = np.random.default_rng(seed)
    xyz = np.vstack([rng.random((n, 3)), 0.3 + 0.01 * rng.standard_normal((blob, 3))]) if blob else rng.random((n, 3))
    return np.clip(xyz, 0.0, 1.0)


def _levels(ctx, xyz, d, mppn, options, bounds=UNIT, profile=False):
    import schwarzwald_amd as swz
    spacing = O.spacing_from_diagonal(*bounds, d)
    try:
        for k, v in options.items():
            ctx.set_option(k, v)
        if profile:
            ctx.profile_enable(True)
            ctx.profile_reset()
        g = ctx.tile(xyz, *bounds, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
        return g
    finally:
        for k in options:
            ctx.set_option(k, None)
        if profile:
            ctx.profile_enable(False)


VARIANTS = {
    "default": {},
    "thread-per-point path": {"SWZ_SP_BLOCK": "0"},
    "wide points": {"SWZ_SP_BLOCK_WIDE": "1"},
    "every level": {"SWZ_MD_SPARSE_LIMIT": "1000"},
    "every level, wide": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_BLOCK_WIDE": "1"},
    "coarse cells": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_BLOCK_CL": "4"},
    "blocks that do not fit": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_BLOCK_OWN": "64", "SWZ_SP_BLOCK_HALO": "64"},
    "capacities estimated too small (launches repeated)": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_BLOCK_CAP_SCALE": "0.3"},
    "... and coarse cells on top (finer cells in the end)": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_BLOCK_CAP_SCALE": "0.1", "SWZ_SP_BLOCK_MIN": "1500"},
    "one workgroup per CU": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_BLOCK_PER_CU": "1"},
    "every pair on the original positions": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_FILTER_EPS": "1e30"},
}


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("d,mppn", [(250, 2000), (90, 500)])
def test_block_path_variants_give_the_oracles_set(ctx, name, d, mppn):
    """SWZ_MD_SPARSE_LIMIT=1000 sends every level the keys can decide to the block path -- dense ones too: points with more
    undecided neighbours than a list holds (they search again), blocks beyond the LDS capacity (the level falls back)."""
    xyz = _cloud(11)
    spacing = O.spacing_from_diagonal(*UNIT, d)
    o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, mppn, spacing)
    g = _levels(ctx, xyz, d, mppn, VARIANTS[name])
    assert np.array_equal(g.keys, o["keys"]) and np.array_equal(g.perm, o["perm"])
    assert np.array_equal(g.level, o["level"])


def test_block_path_with_bounds_that_are_not_dyadic_and_take_all_nodes(ctx):
    """Bounds whose key cells are no powers of two, and a max_points_per_node that leaves some nodes of a level unsampled
    (the granule table then skips their points)."""
    rng = np.random.default_rng(5)
    xyz = rng.random((500000, 3)) * 1.37 + 0.211
    xyz[:150000] = 0.3 + xyz[:150000] * 0.25      # a denser corner: its nodes sample while others of the level take all
    bounds = ([0.2, 0.2, 0.2], [1.7, 1.7, 1.7])
    for d, mppn in ((250, 3000), (120, 800)):
        spacing = O.spacing_from_diagonal(*bounds, d)
        o = O.tile(xyz, *bounds, O.MIN_DISTANCE, mppn, spacing)
        for opts in ({}, {"SWZ_MD_SPARSE_LIMIT": "1000"}):
            g = _levels(ctx, xyz, d, mppn, opts, bounds=bounds)
            assert np.array_equal(g.level, o["level"])


def test_block_path_is_what_runs_and_is_deterministic(ctx):
    """The kernel profile of a call holds the class the block kernel reports under, and two runs agree byte for byte."""
    xyz = _cloud(23, n=600000, blob=0)
    a = _levels(ctx, xyz, 250, 2000, {"SWZ_MD_SPARSE_LIMIT": "1000"}, profile=True)
    b = _levels(ctx, xyz, 250, 2000, {"SWZ_MD_SPARSE_LIMIT": "1000"})
    assert np.array_equal(a.level, b.level) and np.array_equal(a.perm, b.perm)
    o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, 2000, O.spacing_from_diagonal(*UNIT, 250))
    assert np.array_equal(a.level, o["level"])


def test_a_wait_that_times_out_is_an_error_not_a_hang(ctx):
    """A time-out of zero: the first wavefront that has to wait for an earlier block gives up, every workgroup leaves, and the
    call returns an error (the library never restarts anything); with the default time-out the same call succeeds."""
    import schwarzwald_amd as swz
    xyz = _cloud(31, n=1500000, blob=0)
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    p = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=spacing)
    try:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", "1000")
        ctx.set_option("SWZ_SP_BLOCK_TIMEOUT_MS", "0")
        try:
            ctx.tile(xyz, *UNIT, p)
            timed_out = False          # (no wavefront ever had to wait twice: nothing to time out -- allowed)
        except swz.SwzError as e:
            timed_out = True
            assert e.code == swz.api.ERR_INTERNAL and "time-out" in str(e)
    finally:
        ctx.set_option("SWZ_SP_BLOCK_TIMEOUT_MS", None)
    try:
        g = ctx.tile(xyz, *UNIT, p)
    finally:
        ctx.set_option("SWZ_MD_SPARSE_LIMIT", None)
    o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, 2000, spacing)
    assert np.array_equal(g.level, o["level"])
    print("timed out:", timed_out)
