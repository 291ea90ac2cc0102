"""GRID_CENTER / JITTERED decided on key coordinates (grid_argmin_keys_kernel, swz_level.hip) against the CPU oracle.

Per grid cell the point with the smallest upper bound of its distance to the target wins outright when every other
point's lower bound is larger; the runs that leaves undecided are repeated with the reference's arithmetic on the original
positions.  The chosen points must be the oracle's whatever the slack of the bounds, the shape of the bounds or their
distance from the origin."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
SAMPLERS = [O.GRID_CENTER, O.JITTERED]


@pytest.fixture(scope="module")
def ctx():
    import schwarzwald_amd as swz
    c = swz.Context(0)
    yield c
    c.close()


def _cloud(rng, n, bmin, bmax):
    bmin, bmax = np.asarray(bmin), np.asarray(bmax)
    u = np.vstack([rng.random((n // 2, 3)), 0.37 + 0.01 * rng.standard_normal((n // 4, 3)),
                   rng.random((max(n // 64, 1), 3))[rng.integers(0, max(n // 64, 1), n - n // 2 - n // 4)]])  # + exact duplicates
    return bmin + np.clip(u, 0.0, 1.0) * (bmax - bmin)


def _check(ctx, xyz, bmin, bmax, sampler, d, mppn, options, expect_equal=True, **kw):
    import schwarzwald_amd as swz
    spacing = O.spacing_from_diagonal(bmin, bmax, d)
    o = O.tile(xyz, bmin, bmax, sampler, mppn, spacing, **kw)
    assert o["status"] == 0
    try:
        for k, v in options.items():
            ctx.set_option(k, v)
        g = ctx.tile(xyz, bmin, bmax, swz.TileParams(sampler=sampler, max_points_per_node=mppn, spacing_at_root=spacing,
                                                     strategy=kw.get("strategy", swz.ACCURATE),
                                                     fast_concurrency=kw.get("fast_concurrency", 8)))
    finally:
        for k in options:
            ctx.set_option(k, None)
    assert np.array_equal(g.keys, o["keys"]) and np.array_equal(g.perm, o["perm"])
    same = np.array_equal(g.level, o["level"])
    if expect_equal:
        bad = np.flatnonzero(g.level != o["level"])
        assert same, "%d points differ, first at sorted position %d: level %d, oracle %d" % (
            bad.size, bad[0], g.level[bad[0]], o["level"][bad[0]])
    return same


@pytest.mark.parametrize("sampler", SAMPLERS)
@pytest.mark.parametrize("slack", [None, "0.2", "1e9"])
def test_grid_samplers_on_keys_match_the_oracle(ctx, sampler, slack):
    """Default slack, a wide one, and one so wide that every run of more than one point goes through the exact pass."""
    rng = np.random.default_rng(5150 + sampler)
    opts = {} if slack is None else {"SWZ_GRID_KEYS_SLACK": slack}
    for bounds in (UNIT, ([-3.25, 10.125, 0.7], [-3.25 + 1.7320508, 10.125 + 1.7320508, 0.7 + 1.7320508])):
        xyz = _cloud(rng, 300000, *bounds)
        for d, mppn in ((250, 1500), (60, 400)):
            _check(ctx, xyz, *bounds, sampler, d, mppn, opts)


@pytest.mark.parametrize("sampler", SAMPLERS)
def test_grid_samplers_on_keys_need_their_bounds(ctx, sampler):
    """Without any width (a slack of MINUS half a cell: every point taken to sit at the centre of its key cell) some cell
    of a cloud with a dozen points per grid cell must come out wrong -- or the bounds are not what decides."""
    rng = np.random.default_rng(99)
    xyz = rng.random((600000, 3))
    assert not _check(ctx, xyz, *UNIT, sampler, 60, 400, {"SWZ_GRID_KEYS_SLACK": "-0.5"}, expect_equal=False)


def test_grid_center_on_keys_with_bounds_that_are_no_cube(ctx):
    """Key cells are boxes then: the distances weigh the axes (JITTERED keeps the positions there: its grid cells are
    cubes of the x-extent whatever the bounds, Sampling.h:621-668 -- checked too, through that path)."""
    rng = np.random.default_rng(8)
    bounds = ([10.0, -2.0, 100.0], [10.0 + 3.3, -2.0 + 1.1, 100.0 + 0.77])
    xyz = _cloud(rng, 250000, *bounds)
    for sampler in SAMPLERS:
        _check(ctx, xyz, *bounds, sampler, 120, 800, {})
        _check(ctx, xyz, *bounds, sampler, 120, 800, {"SWZ_GRID_KEYS_SLACK": "1e9"})


@pytest.mark.parametrize("sampler", SAMPLERS)
def test_grid_samplers_far_from_the_origin(ctx, sampler):
    """UTM-like coordinates: the reference's targets carry the rounding of 5e6-sized coordinates, the slack grows with it;
    bounds so far out that it would pass a quarter of a key cell go back to the positions."""
    rng = np.random.default_rng(31 + sampler)
    for origin, side in (([5.1e6, 4.4e5, 250.0], 1234.5), ([1e9, -2e9, 5e8], 100.0)):
        bounds = (origin, [o + side for o in origin])
        xyz = _cloud(rng, 200000, *bounds)
        _check(ctx, xyz, *bounds, sampler, 250, 1000, {})


@pytest.mark.parametrize("sampler", SAMPLERS)
def test_grid_samplers_keys_off_and_fast_strategy(ctx, sampler):
    import schwarzwald_amd as swz
    rng = np.random.default_rng(77)
    xyz = _cloud(rng, 300000, *UNIT)
    _check(ctx, xyz, *UNIT, sampler, 250, 1500, {"SWZ_GRID_KEYS": "0"})
    _check(ctx, xyz, *UNIT, sampler, 250, 1500, {}, strategy=O.FAST, fast_concurrency=2)


def test_no_position_gather_for_the_grid_samplers(ctx):
    """Neither sampler brings the positions into Morton order any more (the profile of a call has no gather_positions
    entry; it has one when the key path is switched off)."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(1)
    xyz = rng.random((200000, 3))
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    for sampler in (swz.GRID_CENTER, swz.JITTERED):
        p = swz.TileParams(sampler=sampler, max_points_per_node=2000, spacing_at_root=spacing)
        ctx.profile_enable(True)
        try:
            ctx.profile_reset()
            ctx.tile(xyz, *UNIT, p)
            on_keys = ctx.profile_get()
            ctx.set_option("SWZ_GRID_KEYS", "0")
            ctx.profile_reset()
            ctx.tile(xyz, *UNIT, p)
            on_positions = ctx.profile_get()
        finally:
            ctx.set_option("SWZ_GRID_KEYS", None)
            ctx.profile_enable(False)
        assert "gather_positions" not in on_keys
        assert "gather_positions" in on_positions
