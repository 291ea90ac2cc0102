"""Hierarchy metadata of the node table: Entwine names, node boxes, geometric error.
Host functions of the library (no GPU involved), checked against the reference's own vector
(test/TestOctreeNodeIndex.cpp:434-447: "13-410-7041-4059" <-> octants 2,3,1,3,7,7,1,0,5,5,0,5,3 and :449-457:
"22-0-0-0" does not fit 21 levels) and against the oracle's get_bounds_from_morton_index."""
import numpy as np
import pytest

import oracle_lib as O

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])


def _key(octants):
    k = 0
    for l, o in enumerate(octants):
        k |= o << (3 * (20 - l))
    return k


def test_entwine_name_reference_vector():
    import schwarzwald_amd as swz
    octants = [2, 3, 1, 3, 7, 7, 1, 0, 5, 5, 0, 5, 3]
    key = _key(octants)
    assert swz.node_name_entwine(len(octants) - 1, key) == "13-410-7041-4059"
    assert swz.node_from_entwine_name("13-410-7041-4059") == (12, key)
    assert swz.node_name(12, key) == "r" + "".join(map(str, octants))
    assert swz.node_name_entwine(-1, 0) == "0-0-0-0" and swz.node_from_entwine_name("0-0-0-0") == (-1, 0)
    for bad in ("22-0-0-0", "3-8-0-0", "1-2", "x-1-1-1", "2-1-1-1-1"):
        with pytest.raises(ValueError):
            swz.node_from_entwine_name(bad)


def test_entwine_round_trip_and_bounds_match_the_oracle():
    import schwarzwald_amd as swz
    rng = np.random.default_rng(3)
    bmin, bmax = [-512.25, 1000.5, -3.125], [-512.25 + 777.7, 1000.5 + 777.7, -3.125 + 777.7]
    for _ in range(300):
        depth = int(rng.integers(0, 22))
        key = _key([int(o) for o in rng.integers(0, 8, depth)])
        name = swz.node_name_entwine(depth - 1, key)
        assert swz.node_from_entwine_name(name) == (depth - 1, key)
        mn, mx = swz.node_bounds(depth - 1, key, bmin, bmax)
        omn, omx = O.bounds_from_morton_index(key, bmin, bmax, depth)
        assert mn == omn and mx == omx          # bit-identical: same order of operations as get_octant_bounds
    # level-7 box of (0.3, 0.6, 0.9) in the unit cube
    key = O.morton_index([0.3, 0.6, 0.9], *UNIT)
    mn, mx = swz.node_bounds(6, key, *UNIT)
    assert mn == [0.296875, 0.59375, 0.8984375] and mx == [0.3046875, 0.6015625, 0.90625]


def test_geometric_error_halves_per_level():
    import schwarzwald_amd as swz
    s = O.spacing_from_diagonal(*UNIT, 250)
    assert swz.node_geometric_error(-1, s) == float(np.float32(s))           # the root: depth 0
    assert swz.node_geometric_error(3, s) == float(np.float32(s)) / 16.0
