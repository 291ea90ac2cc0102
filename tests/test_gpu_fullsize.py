"""Full-size checks of the HIP path through properties that do not need the oracle to run at that size.

The greedy MIN_DISTANCE sample of a node (PoissonDiskSampling / SparseGrid::add, Sampling.h:421-471,
SparseGrid.cpp:116-146) is characterised point by point:

    taken(p)  <=>  no point taken EARLIER (Morton order) in the same node is closer than the spacing to p

(induction over the order), so any subset of points can be verified independently against the set of taken points
around it.  The test tiles BASELINE's MIN_DISTANCE workload (uniform points, d = 250, 20000 points per node) at
SWZ_FULLSIZE_POINTS points (default 100 M; 1000000000 is the bench configuration), then verifies every point of a
random box per level that way, with the exact arithmetic of the reference ((dx*dx + dy*dy) + dz*dz < float spacing
squared in float).  Sortedness, the permutation and the keys are checked on the whole output.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
MAX_POINTS = 20000
SEED = 0x5C4A72A1D + 3
_N = None


def _points():
    """The number of points: SWZ_FULLSIZE_POINTS, else what the GPU has room for -- BASELINE's 1 B points (configs[2],
    the bench size) on a 288 GB part: 24 GB of positions, 13 GB of outputs and the library's workspace beside them."""
    global _N
    if _N is None:
        env = int(os.environ.get("SWZ_FULLSIZE_POINTS", "0"))
        import torch
        free, _ = torch.cuda.mem_get_info(0)
        if env:
            _N = env
            # a part that has room for the bench size is checked AT the bench size: a smaller run there must be asked for
            if env < 1_000_000_000 and free >= 230e9 and not os.environ.get("SWZ_FULLSIZE_ALLOW_SMALL"):
                pytest.fail("SWZ_FULLSIZE_POINTS=%d on a GPU with %.0f GB free: the full-size checks run at 1 B points there "
                            "(set SWZ_FULLSIZE_ALLOW_SMALL=1 to shrink them on purpose)" % (env, free / 1e9))
        else:
            _N = 1_000_000_000 if free >= 230e9 else (500_000_000 if free >= 120e9 else 100_000_000)
    return _N


def _record(check, cloud, n, levels, points_checked):
    """One line per full-size check for the terminal summary (tests/conftest.py prints them at the end of the run, so that
    what was verified at which size shows in the captured tail of `pytest -m gpu`)."""
    import conftest
    conftest.FULLSIZE_RECORDS.append({"check": check, "cloud": cloud, "points": int(n), "levels": int(levels),
                                      "points_checked": int(points_checked)})


def _log(msg):
    """stdout (pytest shows it with -s or on failure) and a log file that travels back with gpurun_out/: what was verified
    at which size is kept whether or not the run was watched (copied to profiles/rNN/ by tools/profile_round.sh)"""
    print(msg)
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "fullsize_verification.log"), "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass


def cloud_points(kind):
    """the surface-like cloud makes levels of hundreds of millions of cells (per-cell tables): half the points there"""
    return _points() if kind == "uniform" else max(_points() // 2, min(_points(), 100_000_000))


def make_cloud(torch, ctx, dev, N, kind):
    xyz = torch.empty((N, 3), dtype=torch.float64, device=dev)
    ctx.generate_uniform_device(SEED, 0, N, xyz.data_ptr())
    if kind == "clustered":
        # surface-like data (what LAS files look like): a thin wavy sheet, a dense blob and a sparse background,
        # so that sparse and dense MIN_DISTANCE levels, the sparse path's give-up and deep nodes all occur
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        k = N // 3
        xyz[:k, 2] = 0.3 + 0.05 * torch.sin(6.0 * xyz[:k, 0]) * torch.cos(4.0 * xyz[:k, 1]) \
            + 0.0005 * torch.randn(k, dtype=torch.float64, device=dev, generator=g)
        xyz[k:2 * k] = 0.6 + 0.03 * torch.randn((k, 3), dtype=torch.float64, device=dev, generator=g)
        xyz.clamp_(0.0, 1.0)
    return xyz


@pytest.fixture(scope="module", params=["uniform", "clustered"])
def tiled(request):
    import torch
    import schwarzwald_amd as swz
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()  # the library allocates with hipMalloc: hand back what earlier tests left in torch's cache
    N = cloud_points(request.param)
    ctx = swz.Context(0)
    # the context has its own non-blocking stream: run it on torch's, or the tile could start while the torch
    # kernels below are still writing the points (that race once made this fixture look like a hang)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    xyz = make_cloud(torch, ctx, dev, N, request.param)
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    out = {}
    for sampler in (swz.MIN_DISTANCE, swz.RANDOM_GRID):
        keys = torch.empty(N, dtype=torch.int64, device=dev)
        perm = torch.empty(N, dtype=torch.int32, device=dev)
        level = torch.empty(N, dtype=torch.int8, device=dev)
        params = swz.TileParams(sampler=sampler, max_points_per_node=MAX_POINTS, spacing_at_root=spacing)
        stats = ctx.tile_device(xyz.data_ptr(), N, UNIT[0], UNIT[1], params, keys.data_ptr(), perm.data_ptr(),
                                level.data_ptr())
        torch.cuda.synchronize()
        out[sampler] = (keys, perm, level, stats)
        _log("%s cloud, %d points, %s: %d nodes, deepest level %d, %d levels, visit factor %.3f, %d MIN_DISTANCE rounds" % (
            request.param, N, "MIN_DISTANCE" if sampler == swz.MIN_DISTANCE else "RANDOM_GRID", stats["num_nodes"], stats["max_level"],
            stats["num_levels"], stats["points_visited"] / N, stats["min_distance_rounds"]))
    ctx.release_workspace()
    yield {"xyz": xyz, "spacing": spacing, "out": out, "torch": torch, "swz": swz, "n": N, "cloud": request.param}
    ctx.close()


def test_output_is_a_sorted_permutation_with_matching_keys(tiled):
    torch, swz = tiled["torch"], tiled["swz"]
    keys, perm, level, stats = tiled["out"][swz.MIN_DISTANCE]
    N = tiled["n"]
    assert bool((keys[1:] >= keys[:-1]).all())
    ties = keys[1:] == keys[:-1]
    assert bool((perm[1:][ties] > perm[:-1][ties]).all())  # canonical order: (key, original index)
    seen = torch.zeros(N, dtype=torch.bool, device=keys.device)
    seen[perm.long()] = True
    assert bool(seen.all())
    del seen
    # keys against the oracle's encoder on a random sample
    rng = np.random.default_rng(5)
    pick = torch.from_numpy(np.sort(rng.choice(N, size=min(N, 1_000_000), replace=False))).to(keys.device)
    pts = tiled["xyz"][perm[pick].long()].cpu().numpy()
    want = O.index_points(pts, *UNIT)[0]
    assert np.array_equal(keys[pick].cpu().numpy().view(np.uint64), want)
    # every point was stored exactly once, on a level the statistics know about
    assert int(level.min()) >= -1 and int(level.max()) == stats["max_level"]
    assert stats["points_visited"] >= N
    _log("%s cloud, %d points: output sorted with the canonical tie order, perm is a permutation, %d sampled keys equal the "
         "oracle's encoder" % (tiled["cloud"], N, pick.numel()))
    _record("sorted permutation, keys = oracle encoder on a sample", tiled["cloud"], N, stats["num_levels"], N)


def box_of_level(torch, keys, perm, level, xyz, spacing, L, rng, target_points):
    """The points of sampling nodes that are active at level L inside a random box (plus a margin of one spacing): their
    positions, sorted indices, node prefixes and taken flags on the host, the box, the spacing and its float square.
    None when no node of the level samples."""
    s = np.float32(spacing) / np.float32(2.0 ** (L + 1))   # exact in float (power of two)
    sq = float(np.float32(s) * np.float32(s))                # SparseGrid.cpp:18: float product
    active = level >= L
    n_active = int(active.sum())
    if n_active == 0:
        return None
    shift = 63 - 3 * (L + 1)
    node = (keys >> shift) if shift < 63 else torch.zeros_like(keys)
    # nodes that sample (more than MAX_POINTS arrive) -- the others take everything
    uniq, counts = torch.unique_consecutive(node[active], return_counts=True)
    sampling_nodes = uniq[counts > MAX_POINTS]
    if sampling_nodes.numel() == 0:
        return None
    in_sampling = active & torch.isin(node, sampling_nodes)
    # a box holding about target_points of them
    frac = min(1.0, target_points / max(1, int(in_sampling.sum())))
    h = 0.5 * frac ** (1.0 / 3.0)
    idx_all = torch.nonzero(in_sampling).squeeze(1)
    pos = xyz[perm[idx_all].long()]
    # centred on a random point of the level (so that dense parts are visited in proportion to their points) and
    # shrunk until it holds about target_points
    c = pos[int(rng.integers(0, pos.shape[0]))].cpu().numpy()
    c = np.clip(c, h, 1.0 - h)
    while True:
        lo = torch.tensor(c - h - float(s) * 1.01, device=pos.device)
        hi = torch.tensor(c + h + float(s) * 1.01, device=pos.device)
        near = ((pos >= lo) & (pos <= hi)).all(dim=1)
        if int(near.sum()) <= 2 * target_points or h < 4.0 * float(s):
            break
        h *= 0.75
    idx = idx_all[near]
    return {"P": pos[near].cpu().numpy(), "sorted_index": idx.cpu().numpy(), "node": node[idx].cpu().numpy(),
            "taken": (level[idx] == L).cpu().numpy(), "c": c, "h": h, "s": s, "sq": sq}


def _check_level(tiled, L, rng, target_points=1_500_000):
    """Verifies taken(p) <=> no earlier taken point of the same node within the spacing, for all points of a random
    box that are active at level L.  Returns (#points checked, #taken among them)."""
    from scipy.spatial import cKDTree
    torch, swz = tiled["torch"], tiled["swz"]
    keys, perm, level, _ = tiled["out"][swz.MIN_DISTANCE]
    b = box_of_level(torch, keys, perm, level, tiled["xyz"], tiled["spacing"], L, rng, target_points)
    if b is None:
        return 0, 0
    P, sorted_index, node_h, taken, c, h, s, sq = b["P"], b["sorted_index"], b["node"], b["taken"], b["c"], b["h"], b["s"], b["sq"]
    inner = np.all((P >= c - h) & (P <= c + h), axis=1)              # the points to verify ...
    if int(inner.sum()) > target_points:                              # ... a random subset of them in dense parts
        drop = rng.choice(np.nonzero(inner)[0], size=int(inner.sum()) - target_points, replace=False)
        inner[drop] = False
    Q, Qi = P[inner], np.nonzero(inner)[0]
    T, Ti = P[taken], np.nonzero(taken)[0]
    pairs = cKDTree(Q).sparse_distance_matrix(cKDTree(T), float(s) * (1.0 + 1e-9), output_type="ndarray")
    qi, ti = Qi[pairs["i"]], Ti[pairs["j"]]
    d = P[qi] - P[ti]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]  # Vector3.h:55-62 order of operations
    hit = (d2 < sq) & (sorted_index[ti] < sorted_index[qi]) & (node_h[ti] == node_h[qi])
    has_earlier = np.zeros(P.shape[0], dtype=bool)
    has_earlier[qi[hit]] = True
    bad = np.nonzero(inner & (taken == has_earlier))[0]
    assert bad.size == 0, "level %d: %d of %d points decided differently from the greedy rule, first sorted index %d" % (
        L, bad.size, int(inner.sum()), int(sorted_index[bad[0]]))
    return int(inner.sum()), int((taken & inner).sum())


def test_min_distance_levels_obey_the_greedy_rule(tiled):
    swz = tiled["swz"]
    stats = tiled["out"][swz.MIN_DISTANCE][3]
    rng = np.random.default_rng(11)
    checked = 0
    for L in range(-1, stats["max_level"] + 1):
        for _ in range(2):
            n, t = _check_level(tiled, L, rng)
            _log("%s cloud, %d points, MIN_DISTANCE level %d: %d points of a random box verified against the greedy rule, "
                 "%d of them taken" % (tiled["cloud"], tiled["n"], L, n, t))
            checked += n
    assert checked > 1_000_000
    _record("exact MIN_DISTANCE: taken(p) <=> no earlier taken point within the spacing, random boxes of every level",
            tiled["cloud"], tiled["n"], stats["max_level"] + 2, checked)


def test_random_grid_takes_the_first_point_of_every_cell(tiled):
    """RandomSortedGridSampling (Sampling.h:187-308): in a sampling node at level L the taken points are exactly
    the first points of the runs of equal key prefix at level L + 7 (d = 250 on a cube: 128 cells per axis)."""
    torch, swz = tiled["torch"], tiled["swz"]
    keys, perm, level, stats = tiled["out"][swz.RANDOM_GRID]
    checked_rg, levels_rg = 0, 0
    for L in range(-1, stats["max_level"] + 1):
        active = level >= L
        k = keys[active]
        lv = level[active]
        shift = 63 - 3 * (L + 1)
        node = (k >> shift) if shift < 63 else torch.zeros_like(k)
        uniq, inv, counts = torch.unique_consecutive(node, return_inverse=True, return_counts=True)
        sampling = (counts > MAX_POINTS)[inv]
        cand = L + 7
        if cand > 20:
            continue
        cell = k >> (3 * (20 - cand))
        head = torch.ones_like(cell, dtype=torch.bool)
        head[1:] = cell[1:] != cell[:-1]
        want_taken = torch.where(sampling, head, torch.ones_like(head))
        assert bool(((lv == L) == want_taken).all()), "level %d" % L
        _log("%s cloud, %d points, RANDOM_GRID level %d: %d active points, %d taken = the heads of the cell runs (every point checked)" % (
            tiled["cloud"], tiled["n"], L, int(active.sum()), int((lv == L).sum())))
        checked_rg += int(active.sum())
        levels_rg += 1
    _record("RANDOM_GRID: taken = heads of the cell runs, every point of every level", tiled["cloud"], tiled["n"], levels_rg, checked_rg)


# ---------------------------------------------------------------------------------------------------------------------
# GRID_CENTER and JITTERED at full size (BASELINE config 2: 100 M uniform points, GRID_CENTER, one GPU), checked against
# an independent characterisation evaluated with torch on the whole output: in a sampling node at level L every run of
# equal key prefix at the grid level (L + 7 for d = 250 on a cube) takes exactly one point, the FIRST one (Morton
# order) with the smallest (dx*dx + dy*dy) + dz*dz to the cell's target -- the centre of the cell's box
# (GridCenterSampling, Sampling.h:387-403) or the jittered target node_min + (g * cell + (P - 1) * cell / cells)
# (JitteredSampling, Sampling.h:655-750).  In the unit cube every box edge is a dyadic rational, so targets are exact
# in any evaluation order and the check does not depend on the oracle.
def _contract3(v, torch):
    """every third bit of v (torch int64) -> compact integer: contract_bits_by_3 (stuff.h:223-234)"""
    v = v & 0x1249249249249249
    v = (v | (v >> 2)) & 0x30C30C30C30C30C3
    v = (v | (v >> 4)) & 0xF00F00F00F00F00F
    v = (v | (v >> 8)) & 0x00FF0000FF0000FF
    v = (v | (v >> 16)) & 0x00FF00000000FFFF
    v = (v | (v >> 32)) & 0x00000000FFFFFFFF
    return v


def _jitter_tables():
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "jitter_tables.inc")).read()
    out = {}
    for w in (16, 32, 64):
        body = text.split("SWZ_JITTER_TABLE(%d)" % w)[1].split("};")[0]
        nums = [int(x) for x in re.findall(r"\b\d+\b", body)]
        out[w] = np.array(nums[-16 * w:], dtype=np.int64).reshape(16, w)
    return out


@pytest.mark.parametrize("sampler_name", ["GRID_CENTER", "JITTERED"])
def test_grid_samplers_take_the_first_argmin_of_every_cell(sampler_name):
    import torch
    import schwarzwald_amd as swz
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    n = min(_points(), 100_000_000)
    xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
    ctx.generate_uniform_device(SEED, 0, n, xyz.data_ptr())
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    level = torch.empty(n, dtype=torch.int8, device=dev)
    params = swz.TileParams(sampler=getattr(swz, sampler_name), max_points_per_node=MAX_POINTS, spacing_at_root=spacing)
    stats = ctx.tile_device(xyz.data_ptr(), n, *UNIT, params, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
    torch.cuda.synchronize()
    ctx.release_workspace()
    ctx.close()
    assert int(level.min()) >= -1 and int(level.max()) == stats["max_level"]
    tables = _jitter_tables()
    levels_grid = 7  # prev_pow2((uint32)(extent / spacing)) = 128 cells per axis and node at d = 250
    checked = 0
    nlevels_checked = 0
    for L in range(-1, stats["max_level"] + 1):
        active = torch.nonzero(level >= L).squeeze(1)
        if active.numel() == 0:
            continue
        nlevels_checked += 1
        k = keys[active]
        shift = 63 - 3 * (L + 1)
        node = (k >> shift) if shift < 63 else torch.zeros_like(k)
        _, inv, counts = torch.unique_consecutive(node, return_inverse=True, return_counts=True)
        sampling = (counts > MAX_POINTS)[inv]
        taken = level[active] == L
        assert bool(taken[~sampling].all()), "level %d: a node with <= max points did not take everything" % L
        if not bool(sampling.any()):
            continue
        grid_level = L + levels_grid
        assert grid_level <= 20
        csh = 3 * (20 - grid_level)
        cell = k >> csh                                    # prefix with grid_level + 1 octants
        depth = grid_level + 1
        gx, gy, gz = _contract3(cell >> 2, torch), _contract3(cell >> 1, torch), _contract3(cell, torch)
        pos = xyz[perm[active].long()]
        size = 0.5 ** depth
        if sampler_name == "GRID_CENTER":
            tx = (gx.double() + 0.5) * size
            ty = (gy.double() + 0.5) * size
            tz = (gz.double() + 0.5) * size
        else:
            cells = 1 << levels_grid
            m = cells - 1
            lx, ly, lz = gx & m, gy & m, gz & m           # cell inside the node
            start = (3 * (L + 1)) % 16
            tab = torch.from_numpy(tables[64]).to(dev)
            plen = 64
            px = tab[start][((ly + lz) % plen)] - 1
            py = tab[(start + 1) % 16][((lx + lz) % plen)] - 1
            pz = tab[(start + 2) % 16][((lx + ly) % plen)] - 1
            perm_size = size / cells
            tx = gx.double() * size + px.double() * perm_size   # node_min + (g_local * cell + p * perm) with exact dyadics
            ty = gy.double() * size + py.double() * perm_size
            tz = gz.double() * size + pz.double() * perm_size
        dx, dy, dz = pos[:, 0] - tx, pos[:, 1] - ty, pos[:, 2] - tz
        d2 = (dx * dx + dy * dy) + dz * dz
        del pos, tx, ty, tz, dx, dy, dz
        _, cinv = torch.unique_consecutive(cell, return_inverse=True)
        ncell = int(cinv.max()) + 1
        dmin = torch.full((ncell,), float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, cinv, d2, "amin")
        idx = torch.arange(k.numel(), device=dev)
        cand_idx = torch.where(d2 == dmin[cinv], idx, torch.full_like(idx, k.numel()))
        first = torch.full((ncell,), k.numel(), dtype=torch.int64, device=dev).scatter_reduce(0, cinv, cand_idx, "amin")
        want = torch.zeros_like(taken)
        want[first] = True
        bad = (want != taken) & sampling
        assert not bool(bad.any()), "level %d: %d points decided differently" % (L, int(bad.sum()))
        checked += int(sampling.sum())
        del k, node, cell, d2, cinv, dmin, cand_idx, first, want, bad
    _log("%s, %d uniform points: %d point decisions verified against the torch evaluation (first arg-min per cell)" % (sampler_name, n, checked))
    _record("%s: first arg-min per cell against an independent torch evaluation, every sampled node" % sampler_name, "uniform", n, nlevels_checked, checked)
    assert checked >= n
