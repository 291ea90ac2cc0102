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


def _python_tileset(levels, keys, bmin, bmax, spacing, offset):
    """Cesium3DTilesPersistence::on_write_node restated on node names (Cesium3DTilesPersistence.cpp:80-156): every
    written node creates itself and its missing ancestors; bounds by descending get_octant_bounds from the root."""
    import schwarzwald_amd as swz
    names = {}
    for lv, key in zip(levels, keys):
        name = swz.node_name(int(lv), int(key))
        for cut in range(1, len(name) + 1):
            names.setdefault(name[:cut], False)
        names[name] = True
    out = {}
    for name, content in names.items():
        mn, mx = list(bmin), list(bmax)
        for ch in name[1:]:
            mn, mx = O.octant_bounds(int(ch), mn, mx)
        depth = len(name) - 1
        out[name] = dict(has_content=content, geometric_error=float(np.float32(spacing)) / 2.0 ** depth,
                         bounds_min=[a + b for a, b in zip(mn, offset)], bounds_max=[a + b for a, b in zip(mx, offset)],
                         children=sorted(n for n in names if len(n) == len(name) + 1 and n.startswith(name)),
                         is_tileset_root=depth % 3 == 0)
    return out


def _check_tileset(levels, keys, bmin, bmax, spacing, offset):
    import schwarzwald_amd as swz
    tree = swz.tileset_build(levels, keys, bmin, bmax, spacing, offset)
    want = _python_tileset(levels, keys, bmin, bmax, spacing, offset)
    assert len(tree) == len(want)
    names = [swz.node_name(t["level"], t["key"]) for t in tree]
    assert names == sorted(names, key=lambda s: (len(s), s)) and set(names) == set(want)
    for i, (t, name) in enumerate(zip(tree, names)):
        w = want[name]
        assert t["has_content"] == w["has_content"] and t["is_tileset_root"] == w["is_tileset_root"]
        assert t["geometric_error"] == w["geometric_error"]
        assert t["bounds_min"] == w["bounds_min"] and t["bounds_max"] == w["bounds_max"]
        kids = [names[j] for j in range(t["first_child"], t["first_child"] + t["num_children"])] if t["num_children"] else []
        assert kids == w["children"]
        for j in range(t["first_child"], t["first_child"] + t["num_children"]):
            assert tree[j]["parent"] == i
        assert (t["parent"] == -1) == (name == "r")
    return tree


def test_tileset_tree_fills_in_missing_ancestors():
    """A FAST-style table without the skipped levels: the ancestors appear without content."""
    keys = [_key([1, 2, 3]), _key([1, 2, 4]), _key([7, 0, 0]), _key([7, 0, 0, 5])]
    levels = [2, 2, 2, 3]
    bmin, bmax = [-512.25, 1000.5, -3.125], [-512.25 + 777.7, 1000.5 + 777.7, -3.125 + 777.7]
    tree = _check_tileset(levels, keys, bmin, bmax, 5.5, [10.0, -20.0, 0.25])
    assert sum(t["has_content"] for t in tree) == 4 and len(tree) == 1 + 2 + 2 + 3 + 1


@pytest.mark.gpu
def test_tileset_tree_of_a_gpu_node_table():
    """Names, bounds, geometric errors and the child lists derived from a node table the GPU produced."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(12)
    xyz = rng.random((200000, 3))
    sp = O.spacing_from_diagonal(*UNIT, 250)
    with swz.Context(0) as ctx:
        r = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.GRID_CENTER, max_points_per_node=500, spacing_at_root=sp))
        order, nodes = ctx.build_node_lists(r.keys, r.level)
    ref = O.tile(xyz, *UNIT, O.GRID_CENTER, 500, sp)
    assert len(nodes["level"]) == ref["stats"]["num_nodes"]
    tree = _check_tileset(nodes["level"], nodes["key"], *UNIT, sp, [0.0, 0.0, 0.0])
    assert all(t["has_content"] for t in tree)          # ACCURATE: every ancestor holds points itself
    assert len(tree) == len(nodes["level"])


def test_required_morton_index_depth_matches_the_oracle_near_powers_of_two():
    """log2f of a ratio narrowed to float decides the depth; spacings whose ratio to the extent
    is within a few float ulps of a power of two are where a different evaluation order would show."""
    from schwarzwald_amd.api import required_morton_index_depth
    rng = np.random.default_rng(9)
    cases = 0
    for ext in (1.0, 777.7, 1048576.0, 0.001953125, 3.3e-3):
        bmin = [-1.5, 2.0, 0.25]
        bmax = [bmin[0] + ext, bmin[1] + ext, bmin[2] + ext]
        e = bmax[0] - bmin[0]
        for k in range(1, 30):
            base = np.float32(e / 2.0 ** k)
            for ulps in (-3, -2, -1, 0, 1, 2, 3):
                sp = float(np.nextafter(base, np.float32(np.inf if ulps > 0 else -np.inf)) if abs(ulps) == 1 else
                           base * np.float32(1.0 + ulps * 2.0 ** -23))
                for sampler in (O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED):
                    for lv in (-1, 0, 1, 5, 13, 14, 19, 20):
                        want = O.lib().orc_required_morton_index_depth(sampler, lv, O._vec3(bmin), O._vec3(bmax),
                                                                       O.C.c_float(sp))
                        assert required_morton_index_depth(sampler, lv, bmin, bmax, sp) == want, (ext, k, ulps, sampler, lv)
                        cases += 1
        for _ in range(200):
            sp = float(np.float32(e * 10.0 ** rng.uniform(-7, 0.5)))
            for sampler in range(4):
                lv = int(rng.integers(-1, 21))
                want = O.lib().orc_required_morton_index_depth(sampler, lv, O._vec3(bmin), O._vec3(bmax), O.C.c_float(sp))
                assert required_morton_index_depth(sampler, lv, bmin, bmax, sp) == want
    assert cases > 10000
