"""Multi-batch tiling: swz_tiler fed k batches must produce the node files the
multi-batch oracle (oracle/oracle.cpp MBTiler: read_pnts_from_disk, merge_node_data_sorted, the behaviour switch on
the cached count, FAST later iterations + finalize) produces: same nodes, same point ids, same order in every file."""
import numpy as np
import pytest

import oracle_lib as O

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
ODD = ([-512.25, 1000.5, -3.125], [-512.25 + 777.7, 1000.5 + 777.7, -3.125 + 777.7])
SAMPLERS = [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED]


def _oracle_files(bounds, xyz, k, sampler, max_points, spacing, strategy, concurrency, max_depth=100):
    t = O.Tiler(bounds[0], bounds[1], sampler, max_points, spacing, max_depth=max_depth, strategy=strategy,
                fast_concurrency=concurrency)
    for part in np.array_split(xyz, k):
        st = t.add_batch(part)
        assert st == 0, st
    assert t.finalize() == 0
    ex = t.export()
    c = t.counts()
    t.close()
    return ex, c


def _points(rng, n, bounds, clustered):
    lo, hi = np.array(bounds[0]), np.array(bounds[1])
    if not clustered:
        return lo + rng.random((n, 3)) * (hi - lo)
    k = n // 3
    a = np.column_stack([rng.random(k), rng.random(k), 0.3 + 0.002 * rng.standard_normal(k)])
    b = 0.6 + 0.03 * rng.standard_normal((k, 3))
    base = rng.random((max((n - 2 * k) // 6, 1), 3))
    d = base[rng.integers(0, base.shape[0], n - 2 * k)]  # exact duplicates: equal keys across batches
    u = np.clip(np.vstack([a, b, d]), 0.0, 1.0)
    u = u[rng.permutation(n)]
    return lo + u * (hi - lo)


# --------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("sampler", SAMPLERS)
@pytest.mark.parametrize("strategy", [O.ACCURATE, O.FAST])
def test_oracle_one_batch_equals_single_batch_tiler(sampler, strategy):
    """k = 1 must reproduce orc_tile (the single-batch restatement pinned in round 1)."""
    rng = np.random.default_rng(5)
    n = 40000
    xyz = _points(rng, n, ODD, False)
    sp = O.spacing_from_diagonal(*ODD, 32)
    one = O.tile(xyz, *ODD, sampler, 64, sp, strategy=strategy, fast_concurrency=2)
    assert one["status"] == 0
    ex, c = _oracle_files(ODD, xyz, 1, sampler, 64, sp, strategy, 2)
    expect = {}
    for i in range(n):
        lv = int(one["level"][i])
        masks = [lv] + [b - 1 for b in range(22) if (int(one["dup"][i]) >> b) & 1]
        for l in masks:
            sh = (20 - l) * 3 if l >= 0 else 63
            key = ((int(one["keys"][i]) >> sh) << sh) if l >= 0 else 0
            expect.setdefault((l, key), []).append(int(one["perm"][i]))
    got = {(int(ex["level"][j]), int(ex["key"][j])): list(ex["ids"][int(ex["offset"][j]):int(ex["offset"][j] + ex["count"][j])])
           for j in range(len(ex["level"]))}
    assert got == expect


@pytest.mark.parametrize("sampler", SAMPLERS)
def test_oracle_multibatch_invariants(sampler):
    """The invariants of the reference's (disabled) integration test, test/TestTiler.cpp:113-161: every point stored
    exactly once (ACCURATE), inside its node; a node that was sampled before never takes all."""
    rng = np.random.default_rng(11)
    n = 30000
    xyz = _points(rng, n, UNIT, True)
    sp = O.spacing_from_diagonal(*UNIT, 32)
    ex, c = _oracle_files(UNIT, xyz, 4, sampler, 200, sp, O.ACCURATE, 2)
    assert c["num_stored"] == n and np.array_equal(np.sort(ex["ids"]), np.arange(n, dtype=np.uint32))
    keys, clamped = O.index_points(xyz, *UNIT)
    for j in range(len(ex["level"])):
        lv = int(ex["level"][j])
        if lv < 0:
            continue
        ids = ex["ids"][int(ex["offset"][j]):int(ex["offset"][j] + ex["count"][j])]
        sh = (20 - lv) * 3
        assert np.all((keys[ids] >> np.uint64(sh)) == (ex["key"][j] >> np.uint64(sh)))


# --------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    import schwarzwald_amd as swz
    c = swz.Context(0)
    yield c
    c.close()


def _gpu_files(ctx, bounds, xyz, k, sampler, max_points, spacing, strategy, concurrency, staged, max_depth=100,
               fail_alloc_in_finalize=None):
    import schwarzwald_amd as swz
    import torch
    params = swz.TileParams(sampler=sampler, max_points_per_node=max_points, spacing_at_root=spacing, max_depth=max_depth,
                            strategy=strategy, fast_concurrency=concurrency)
    visited = 0
    with swz.Tiler(ctx, bounds[0], bounds[1], params) as t:
        parts = np.array_split(xyz, k)
        if staged:
            pinned = []
            for p in parts:
                a = swz.pinned_empty(p.shape, np.float64)
                a[...] = p
                pinned.append(a)
            t.stage_batch(pinned[0])
            for i in range(k):
                if i + 1 < k:
                    t.stage_batch(pinned[i + 1])
                visited += t.tile_staged()["points_visited"]
        else:
            for p in parts:
                d = torch.from_numpy(np.ascontiguousarray(p)).cuda()
                torch.cuda.synchronize()
                visited += t.add_batch_device(d.data_ptr(), p.shape[0])["points_visited"]
        if fail_alloc_in_finalize:
            ctx.set_option("SWZ_FAIL_ALLOC", fail_alloc_in_finalize)
        try:
            t.finalize()
        finally:
            if fail_alloc_in_finalize:
                ctx.set_option("SWZ_FAIL_ALLOC", None)
        info = t.info()
        table = t.node_table()
        ns = int(info["num_stored"])
        d_keys = torch.empty(max(ns, 1), dtype=torch.int64, device="cuda")
        d_ids = torch.empty(max(ns, 1), dtype=torch.int32, device="cuda")
        d_lvl = torch.empty(max(ns, 1), dtype=torch.int8, device="cuda")
        t.export_device(d_keys.data_ptr(), d_ids.data_ptr(), d_lvl.data_ptr())
        ids = d_ids.cpu().numpy().view(np.uint32)[:ns]
        lvl = d_lvl.cpu().numpy()[:ns]
    return dict(table=table, ids=ids, level=lvl, info=info, visited=visited)


def _compare(g, ex, c):
    tb = g["table"]
    assert len(tb["level"]) == len(ex["level"]) == c["num_nodes"]
    assert np.array_equal(tb["level"], ex["level"])
    assert np.array_equal(tb["key"], ex["key"])
    assert np.array_equal(tb["offset"], ex["offset"])
    assert np.array_equal(tb["count"], ex["count"])
    assert np.array_equal(g["ids"], ex["ids"])
    # the exported per-entry level agrees with the table
    assert np.array_equal(g["level"], np.repeat(ex["level"], ex["count"].astype(np.int64)))


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", SAMPLERS)
@pytest.mark.parametrize("strategy", [O.ACCURATE, O.FAST])
@pytest.mark.parametrize("k", [1, 2, 5])
def test_gpu_multibatch_matches_oracle(ctx, sampler, strategy, k):
    rng = np.random.default_rng(100 * sampler + 10 * strategy + k)
    n = 120000
    bounds = ODD if (sampler + k) % 2 else UNIT
    xyz = _points(rng, n, bounds, clustered=(k == 5))
    sp = O.spacing_from_diagonal(*bounds, 32)
    ex, c = _oracle_files(bounds, xyz, k, sampler, 300, sp, strategy, 2)
    g = _gpu_files(ctx, bounds, xyz, k, sampler, 300, sp, strategy, 2, staged=(k == 2))
    assert g["info"]["rekey_inversions"] == 0 and c["unsorted_cached_nodes"] == 0
    _compare(g, ex, c)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", SAMPLERS)
def test_gpu_multibatch_d250_large_nodes(ctx, sampler):
    """BASELINE's spacing (diagonal / 250) and real node sizes: 5 batches of 200 k points, max 20 000 per node."""
    rng = np.random.default_rng(7 + sampler)
    n = 1000000
    xyz = _points(rng, n, UNIT, clustered=False)
    sp = O.spacing_from_diagonal(*UNIT, 250)
    ex, c = _oracle_files(UNIT, xyz, 5, sampler, 20000, sp, O.ACCURATE, 8)
    g = _gpu_files(ctx, UNIT, xyz, 5, sampler, 20000, sp, O.ACCURATE, 8, staged=True)
    _compare(g, ex, c)
    assert g["info"]["num_points"] == n and g["info"]["num_batches"] == 5


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", SAMPLERS)
@pytest.mark.parametrize("strategy", [O.ACCURATE, O.FAST])
def test_gpu_multibatch_spatially_coherent_batches(ctx, sampler, strategy):
    """Batches that arrive the way LAS tiles do: each reaches a few subtrees and leaves the files of all other nodes
    alone.  The node store then appends the rewritten files behind the untouched ones and gathers the live files into its
    other side when a side is full -- 24 batches so that this happens several times per level --, and a call between two
    batches (info: the node table) must not disturb it."""
    import schwarzwald_amd as swz
    import torch
    rng = np.random.default_rng(41 + 10 * sampler + strategy)
    n, k = 240000, 24
    xyz = _points(rng, n, UNIT, clustered=False)
    # strips along x, every third strip a second time around (a tile visited again later)
    xyz = xyz[np.argsort(xyz[:, 0], kind="stable")]
    parts = np.array_split(xyz, k)
    order = [i for i in range(k) if i % 3] + [i for i in range(k) if i % 3 == 0]
    xyz = np.vstack([parts[i] for i in order])
    sp = O.spacing_from_diagonal(*UNIT, 64)
    ex, c = _oracle_files(UNIT, xyz, k, sampler, 500, sp, strategy, 2)
    g = _gpu_files(ctx, UNIT, xyz, k, sampler, 500, sp, strategy, 2, staged=False)
    assert g["info"]["rekey_inversions"] == 0 and c["unsorted_cached_nodes"] == 0
    _compare(g, ex, c)
    # the same with the node table read in the middle of the data set
    params = swz.TileParams(sampler=sampler, max_points_per_node=500, spacing_at_root=sp, strategy=strategy, fast_concurrency=2)
    ctx2 = swz.Context(0)  # (a workspace of its own: what the calls below allocate is then known)
    with swz.Tiler(ctx2, UNIT[0], UNIT[1], params) as t:
        for i, p in enumerate(np.array_split(xyz, k)):
            d = torch.from_numpy(np.ascontiguousarray(p)).cuda()
            torch.cuda.synchronize()
            t.add_batch_device(d.data_ptr(), p.shape[0])
            if i in (5, 6, 17):
                # (once with the first allocation of the call pretending to run out of memory: the workspace then frees the
                # scratch of the batches -- nobody holds on to it between two batches -- and the next batch allocates again)
                held = ctx2.workspace_bytes()
                if i == 6:
                    ctx2.set_option("SWZ_FAIL_ALLOC", "tl_head_pos")
                try:
                    mid = t.node_table()
                finally:
                    ctx2.set_option("SWZ_FAIL_ALLOC", None)
                if i == 6:
                    assert ctx2.workspace_bytes() < held, "the batches' scratch should have been given back"
                assert int(mid["count"].sum()) == int(t.info()["num_stored"])
        t.finalize()
        tb = t.node_table()
        ns = int(t.info()["num_stored"])
        d_ids = torch.empty(ns, dtype=torch.int32, device="cuda")
        t.export_device(None, d_ids.data_ptr(), None)
        assert np.array_equal(tb["count"], ex["count"]) and np.array_equal(tb["key"], ex["key"])
        assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), ex["ids"])
    ctx2.close()


INCREMENTAL = {
    # every level the keys can decide goes to the block path; a batch on top of files samples only what it can change
    "always": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_INCREMENTAL": "0.001", "SWZ_SP_INCREMENTAL_MAX": "1.0"},
    "always, subset blocks that do not fit": {"SWZ_MD_SPARSE_LIMIT": "1000", "SWZ_SP_INCREMENTAL": "0.001", "SWZ_SP_INCREMENTAL_MAX": "1.0",
                                              "SWZ_SP_BLOCK_OWN": "64", "SWZ_SP_BLOCK_HALO": "64"},
    "default thresholds": {},
    "never": {"SWZ_SP_INCREMENTAL": "0"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(INCREMENTAL))
@pytest.mark.parametrize("strategy", [O.ACCURATE, O.FAST])
@pytest.mark.parametrize("clustered", [False, True])
def test_gpu_multibatch_min_distance_samples_only_what_a_batch_can_change(ctx, name, strategy, clustered):
    """MIN_DISTANCE on a batch merged with the files of earlier batches (swz_mdblock.hip, sb_incremental): the entries of a
    file with more than max_points points never reject one another, so only the new points and the old points within a
    spacing of one are sampled again -- the files must be the oracle's, which samples the whole union (tile_node,
    TilingAlgorithms.cpp:421-442; the behaviour switch on the cached count :272-275)."""
    rng = np.random.default_rng(900 + 2 * strategy + clustered)
    n, k = 400000, 8
    xyz = _points(rng, n, UNIT, clustered=clustered)
    if not clustered:
        # later batches thin out: new points few and far between on top of full files
        xyz = np.vstack([xyz[:300000], xyz[300000:][np.argsort(rng.random(100000))]])
    sp = O.spacing_from_diagonal(*UNIT, 128)
    ex, c = _oracle_files(UNIT, xyz, k, O.MIN_DISTANCE, 1000, sp, strategy, 2)
    try:
        for key, v in INCREMENTAL[name].items():
            ctx.set_option(key, v)
        g = _gpu_files(ctx, UNIT, xyz, k, O.MIN_DISTANCE, 1000, sp, strategy, 2, staged=False)
    finally:
        for key in INCREMENTAL[name]:
            ctx.set_option(key, None)
    _compare(g, ex, c)


@pytest.mark.gpu
def test_gpu_multibatch_uneven_batches_on_full_files(ctx, capfd):
    """One big batch fills the files, then many small ones arrive: almost every point of a level is an old one and only the
    neighbourhoods of the few new points are looked at (the case the subset is for)."""
    rng = np.random.default_rng(77)
    n = 600000
    xyz = _points(rng, n, ODD, clustered=False)
    sizes = [500000] + [10000] * 10
    sp = O.spacing_from_diagonal(*ODD, 160)
    t = O.Tiler(ODD[0], ODD[1], O.MIN_DISTANCE, 2000, sp, max_depth=100, strategy=O.ACCURATE, fast_concurrency=2)
    off = 0
    for sz in sizes:
        assert t.add_batch(xyz[off:off + sz]) == 0
        off += sz
    assert t.finalize() == 0
    ex, c = t.export(), t.counts()
    t.close()
    import schwarzwald_amd as swz
    import torch
    params = swz.TileParams(sampler=O.MIN_DISTANCE, max_points_per_node=2000, spacing_at_root=sp, max_depth=100, strategy=O.ACCURATE,
                            fast_concurrency=2)
    with swz.Tiler(ctx, ODD[0], ODD[1], params) as tl:
        off = 0
        for sz in sizes:
            d = torch.from_numpy(np.ascontiguousarray(xyz[off:off + sz])).cuda()
            torch.cuda.synchronize()
            ctx.set_option("SWZ_DEBUG", "1")  # (the library then says which path a level took)
            try:
                tl.add_batch_device(d.data_ptr(), sz)
            finally:
                ctx.set_option("SWZ_DEBUG", None)
            off += sz
        said = capfd.readouterr().err
        # with the default thresholds: some level of some small batch was sampled as the subset the batch can change
        assert "(what the new points can change)" in said, said[-2000:]
        tl.finalize()
        tb = tl.node_table()
        ns = int(tl.info()["num_stored"])
        d_ids = torch.empty(ns, dtype=torch.int32, device="cuda")
        tl.export_device(None, d_ids.data_ptr(), None)
        assert np.array_equal(tb["count"], ex["count"]) and np.array_equal(tb["key"], ex["key"])
        assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), ex["ids"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_gpu_multibatch_subset_sampling_changes_nothing_at_scale(kind):
    """40 M points in 20 batches cut out of the whole cloud, BASELINE's spacing and node size, FAST: the files with the subset
    path (levels of hundreds of nodes, subset blocks that are clumps, launches repeated when a clump does not fit) are the
    files without it, entry for entry -- the oracle would take minutes at this size, the small cases above are against it."""
    import torch
    import schwarzwald_amd as swz
    from test_gpu_fullsize import make_cloud
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    n, k = 40_000_000, 20
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    xyz = make_cloud(torch, ctx, dev, n, kind)
    xyz = xyz[torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))].contiguous()
    torch.cuda.synchronize()
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal(*UNIT, 250),
                            strategy=swz.FAST, fast_concurrency=8)
    out = {}
    for inc in ("0.3", "0"):
        ctx.set_option("SWZ_SP_INCREMENTAL", inc)
        with swz.Tiler(ctx, UNIT[0], UNIT[1], params, capacity_hint=n) as t:
            for i in range(k):
                lo, hi = (i * n) // k, ((i + 1) * n) // k
                t.add_batch_device(xyz[lo:hi].data_ptr(), hi - lo)
            t.finalize()
            tb = t.node_table()
            ns = int(t.info()["num_stored"])
            d_ids = torch.empty(ns, dtype=torch.int32, device=dev)
            t.export_device(None, d_ids.data_ptr(), None)
            out[inc] = (tb, d_ids.cpu().numpy().view(np.uint32))
    ctx.set_option("SWZ_SP_INCREMENTAL", None)
    ctx.close()
    a, b = out["0.3"], out["0"]
    assert len(a[1]) >= n  # (FAST: the levels above the start level are rebuilt from copies of their children's points)
    for col in ("level", "key", "offset", "count"):
        assert np.array_equal(a[0][col], b[0][col]), col
    assert np.array_equal(a[1], b[1])


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1])
def test_gpu_fast_finalize_survives_out_of_memory_inside_its_levels(flags):
    """FAST + MIN_DISTANCE: finalize rebuilds the skipped levels from ALL stored points, so its levels are larger than any
    batch's and the sampler scratch grows there.  When such an allocation runs out of memory the workspace frees scratch
    of earlier epochs -- but not the arrays finalize itself holds across the level ("tl_keys", "tl_wx", ...), although no
    batch is open (they were freed and read afterwards).  SWZ_FAIL_ALLOC="md_*" makes the first attempt of every
    growing MIN_DISTANCE buffer fail."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(77)
    n, k = 150000, 3
    xyz = _points(rng, n, UNIT, clustered=False)
    sp = O.spacing_from_diagonal(*UNIT, 64)
    ex, c = _oracle_files(UNIT, xyz, k, O.MIN_DISTANCE, 500, sp, O.FAST, 2)
    ctx2 = swz.Context(0)
    try:
        if flags:
            ctx2.set_option("SWZ_POISON", "205")  # freed-and-reallocated arrays then never hold the old bytes by luck
        g = _gpu_files(ctx2, UNIT, xyz, k, O.MIN_DISTANCE, 500, sp, O.FAST, 2, staged=False, fail_alloc_in_finalize="md_*")
    finally:
        ctx2.close()
    _compare(g, ex, c)


@pytest.mark.gpu
def test_gpu_multibatch_terminal_nodes(ctx):
    """max_depth = 2 makes level 2 terminal: its nodes append new ++ cached without sampling
    (tile_node :421-442, merge_node_data_unsorted)."""
    rng = np.random.default_rng(3)
    xyz = _points(rng, 60000, UNIT, clustered=True)
    sp = O.spacing_from_diagonal(*UNIT, 32)
    for sampler in (O.RANDOM_GRID, O.MIN_DISTANCE):
        ex, c = _oracle_files(UNIT, xyz, 3, sampler, 100, sp, O.ACCURATE, 2, max_depth=2)
        g = _gpu_files(ctx, UNIT, xyz, 3, sampler, 100, sp, O.ACCURATE, 2, staged=False, max_depth=2)
        _compare(g, ex, c)
        assert int(ex["level"].max()) == 2


# --------------------------------------------------------------------------------------------------- re-rooting
def _deep_cloud(rng, n):
    """All points inside a corner 1e-5 of the root's extent wide: with spacing = extent / 4096 the grid samplers need
    more than 21 key levels from node level 9 on (tile_node re-roots there, TilingAlgorithms.cpp:444-483) -- the
    situation of the reference's disabled test "Tiler with deep tree works" (test/TestTiler.cpp:164-190)."""
    return rng.random((n, 3)) * 0.01


DEEP = ([0.0, 0.0, 0.0], [1024.0, 1024.0, 1024.0])


@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.JITTERED])
def test_oracle_reroots_deep_nodes(sampler):
    rng = np.random.default_rng(4)
    n = 60000
    xyz = _deep_cloud(rng, n)
    sp = float(np.float32(1024.0 / 4096.0))
    assert O.lib().orc_required_morton_index_depth(sampler, 9, O._vec3(DEEP[0]), O._vec3(DEEP[1]), O.C.c_float(sp)) >= 21
    assert O.lib().orc_required_morton_index_depth(sampler, 8, O._vec3(DEEP[0]), O._vec3(DEEP[1]), O.C.c_float(sp)) < 21
    ex, c = _oracle_files(DEEP, xyz, 2, sampler, 200, sp, O.ACCURATE, 2)
    assert c["num_stored"] == n and np.array_equal(np.sort(ex["ids"]), np.arange(n, dtype=np.uint32))
    assert int(ex["level"].max()) > 9          # nodes below the first re-rooted level exist


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.JITTERED])
@pytest.mark.parametrize("k", [1, 3])
def test_gpu_reroots_deep_nodes_like_the_oracle(ctx, sampler, k):
    rng = np.random.default_rng(40 + sampler + k)
    n = 100000
    xyz = _deep_cloud(rng, n)
    sp = float(np.float32(1024.0 / 4096.0))
    ex, c = _oracle_files(DEEP, xyz, k, sampler, 200, sp, O.ACCURATE, 2)
    g = _gpu_files(ctx, DEEP, xyz, k, sampler, 200, sp, O.ACCURATE, 2, staged=False)
    _compare(g, ex, c)
    assert int(ex["level"].max()) > 9


@pytest.mark.gpu
def test_single_batch_entry_point_reports_reroot(ctx):
    """swz_tile's output (one level per point of the sorted batch) cannot express the file order of re-rooted nodes:
    it fails loudly and points to the multi-batch tiler."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(1)
    xyz = _deep_cloud(rng, 50000)
    sp = float(np.float32(1024.0 / 4096.0))
    with pytest.raises(swz.SwzError) as e:
        ctx.tile(xyz, *DEEP, swz.TileParams(sampler=swz.RANDOM_GRID, max_points_per_node=200, spacing_at_root=sp))
    assert e.value.code == 5
    assert "swz_tile_nodes_begin_device" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", SAMPLERS)
def test_single_batch_as_node_files_reroots_like_the_oracle(ctx, sampler):
    """swz_tile_nodes_begin_device / _end_device: ONE batch comes back as node files (node table + contents), re-rooted
    subtrees included (TilingAlgorithms.cpp:444-483) -- the oracle's files of the same batch, node for node and in file
    order."""
    import torch
    import schwarzwald_amd as swz
    rng = np.random.default_rng(70 + sampler)
    n = 100000
    xyz = _deep_cloud(rng, n)
    sp = float(np.float32(1024.0 / 4096.0))
    ex, c = _oracle_files(DEEP, xyz, 1, sampler, 200, sp, O.ACCURATE, 2)
    assert int(ex["level"].max()) > 9
    d = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
    bufs = {}

    def alloc(ns):
        bufs["k"] = torch.empty(max(ns, 1), dtype=torch.int64, device="cuda")
        bufs["i"] = torch.empty(max(ns, 1), dtype=torch.int32, device="cuda")
        bufs["l"] = torch.empty(max(ns, 1), dtype=torch.int8, device="cuda")
        return bufs["k"].data_ptr(), bufs["i"].data_ptr(), bufs["l"].data_ptr()
    params = swz.TileParams(sampler=sampler, max_points_per_node=200, spacing_at_root=sp, fast_concurrency=2)
    stats, table, ns = ctx.tile_nodes_device(d.data_ptr(), n, *DEEP, params, alloc)
    torch.cuda.synchronize()
    g = dict(table=table, ids=bufs["i"].cpu().numpy().view(np.uint32)[:ns], level=bufs["l"].cpu().numpy()[:ns])
    _compare(g, ex, c)
    assert stats["num_nodes"] == c["num_nodes"] and ns == n
    # the context is free again: a second call and a tiler both work
    stats2, table2, ns2 = ctx.tile_nodes_device(d.data_ptr(), n, *DEEP, params, alloc)
    assert np.array_equal(table2["key"], table["key"]) and ns2 == ns
    with swz.Tiler(ctx, DEEP[0], DEEP[1], params) as t:
        assert t.info()["num_points"] == 0


@pytest.mark.gpu
def test_gpu_tiler_is_poisoned_by_a_batch_that_fails_part_way(ctx):
    """A batch that fails after the first levels have been merged into the node store leaves the tiler unusable: every
    later call must say so (SWZ_ERR_TILER_FAILED) instead of reusing point ids on a half-updated store."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(31)
    parts = [rng.random((60000, 3)) for _ in range(3)]
    spacing = O.spacing_from_diagonal(*UNIT, 250)
    params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=500, spacing_at_root=spacing)
    with swz.Tiler(ctx, UNIT[0], UNIT[1], params) as t:
        t.add_batch(parts[0])
        try:
            ctx.set_option("SWZ_MD_SPARSE_LIMIT", "0")
            ctx.set_option("SWZ_MD_ROUND_LIMIT", "1")  # the first MIN_DISTANCE level of the batch gives up
            with pytest.raises(swz.SwzError) as e1:
                t.add_batch(parts[1])
            assert e1.value.code == swz.api.ERR_INTERNAL
        finally:
            ctx.set_option("SWZ_MD_SPARSE_LIMIT", None)
            ctx.set_option("SWZ_MD_ROUND_LIMIT", None)
        for call in (lambda: t.add_batch(parts[2]), t.finalize, lambda: t.level_count(-1), t.node_table):
            with pytest.raises(swz.SwzError) as e2:
                call()
            assert e2.value.code == swz.api.ERR_TILER_FAILED
            assert "earlier batch failed" in str(e2.value)
        assert t.info()["num_points"] == 60000  # still answers, so that a caller can report what was lost
    # a tiler poisoned from outside (what the multi-GPU driver does on the ranks that did not fail)
    with swz.Tiler(ctx, UNIT[0], UNIT[1], params) as t:
        t.add_batch(parts[0])
        t.poison("peer failed")
        with pytest.raises(swz.SwzError) as e3:
            t.add_batch(parts[1])
        assert e3.value.code == swz.api.ERR_TILER_FAILED and "peer failed" in str(e3.value)


def _constructed_inversion(bounds):
    """Two points a, b of one level-0 node whose order flips when their node re-reads them (read_pnts_from_disk,
    TilingAlgorithms.cpp:80-99: the lower key levels come from an index computed against the NODE's bounds).  a sits a
    few ulps from a key-cell boundary in x, where (x - root_min) * (2^21 / root_extent) and the same expression on the
    node's box round to different cells; b sits in the cell between the two places a can have (same x cell as a's
    lower one, next y cell), so a < b in one order and b < a in the other.  The boundary is a boundary of the level-0
    sampling grid (and not of the root's), so that the root leaves both to level 0 and level 0 takes both."""
    mn, ext = bounds[0][0], bounds[1][0] - bounds[0][0]
    two21 = 2097152.0
    mx = mn + ext
    half = (mx - mn) / 2.0
    nmin = mn + half
    nmax = nmin + half

    def root_cell(x):
        return min(int((x - mn) * (two21 / ext)), 2 ** 21 - 1)

    def node_cell(x):
        return 2 ** 20 + (min(int((x - nmin) * (two21 / (nmax - nmin))), 2 ** 21 - 1) >> 1)

    for i in range(2 ** 20 + 8192, 2 ** 21, 16384):        # boundaries of level-7 cells that are not level-6 boundaries
        x = mn + i * (ext / two21)
        for _ in range(64):
            x = float(np.nextafter(x, -np.inf))
            if root_cell(x) != node_cell(x) and {root_cell(x), node_cell(x)} == {i, i - 1}:
                return i, x
    raise AssertionError("no such point found")


@pytest.mark.gpu
def test_gpu_rekey_inversion_is_counted_and_confined(ctx):
    """swz_tiler sorts the cached points of a node when their re-keyed order is no longer ascending; the reference
    merges them as they are (a lossless store is read without a sort, TilingAlgorithms.cpp:103-106).  The library counts
    such nodes' inversions (swz_tiler_info.rekey_inversions) and documents that results may differ from the reference
    only then.  Here an inversion is constructed: the counter must see it (the oracle's own counter too), every point
    must still be stored exactly once, and whatever differs from the oracle must be confined to the node in question."""
    import schwarzwald_amd as swz
    bounds = ODD
    lo, side = np.array(bounds[0]), bounds[1][0] - bounds[0][0]
    i, xa = _constructed_inversion(bounds)
    cell = side / 2097152.0
    jy, kz = 2 ** 20 + 40000, 2 ** 20 + 50000               # even y cell, any z cell, both in the upper halves (octant 7)
    a = np.array([xa, lo[1] + (jy + 0.5) * cell, lo[2] + (kz + 0.5) * cell])
    b = np.array([lo[0] + (i - 1 + 0.5) * cell, lo[1] + (jy + 1 + 0.5) * cell, lo[2] + (kz + 0.5) * cell])
    # a decoy at the start of the root's grid cell (level 6: 16384 key cells wide) so that the root takes neither a nor b
    decoy = np.array([lo[0] + ((i - 1) // 16384 * 16384 + 0.5) * cell, lo[1] + (jy // 16384 * 16384 + 0.5) * cell,
                      lo[2] + (kz // 16384 * 16384 + 0.5) * cell])
    rng = np.random.default_rng(2)
    fill1 = lo + rng.random((4000, 3)) * side
    fill2 = lo + (0.5 + 0.5 * rng.random((3000, 3))) * side   # the second batch touches octant 7, whose file holds a and b
    batches = [np.vstack([fill1, decoy[None], b[None], a[None]]), fill2]
    ia, ib = 4002, 4001
    sp = O.spacing_from_diagonal(*bounds, 250)
    t = O.Tiler(bounds[0], bounds[1], O.RANDOM_GRID, 100, sp)
    for part in batches:
        assert t.add_batch(part.copy()) == 0
    assert t.finalize() == 0
    ex, oc = t.export(), t.counts()
    t.close()
    params = swz.TileParams(sampler=swz.RANDOM_GRID, max_points_per_node=100, spacing_at_root=sp)
    with swz.Tiler(ctx, bounds[0], bounds[1], params) as g:
        for part in batches:
            g.add_batch(part)
        g.finalize()
        info = g.info()
        table = g.node_table()
        import torch
        ns = int(info["num_stored"])
        d_ids = torch.empty(ns, dtype=torch.int32, device="cuda")
        g.export_device(None, d_ids.data_ptr(), None)
        ids = d_ids.cpu().numpy().view(np.uint32)
    assert info["rekey_inversions"] >= 1          # the library saw the inversion ...
    assert oc["unsorted_cached_nodes"] >= 1       # ... and so did the restatement of the reference
    n = sum(len(p) for p in batches)
    assert ns == n and np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))      # every point stored exactly once
    # node files: identical to the oracle's except in the subtree of the level-0 node that holds a and b, where the
    # two orders take different points of the sampling cell the re-keyed a falls into
    def files(level, key, offset, count, idlist):
        out = {}
        for l, k, o, c in zip(level, key, offset, count):
            out[(int(l), int(k))] = idlist[int(o):int(o + c)].tolist()
        return out
    gf = files(table["level"], table["key"], table["offset"], table["count"], ids)
    of = files(ex["level"], ex["key"], ex["offset"], ex["count"], ex["ids"])
    octant7 = 7 << 60
    differing = [k for k in set(gf) | set(of) if gf.get(k) != of.get(k)]
    for (l, k) in differing:
        assert l >= 0 and (k >> 60) == 7, "a file outside octant 7 differs: level %d key %x" % (l, k)
    moved = set()
    for k in differing:
        moved |= set(gf.get(k, [])) ^ set(of.get(k, []))
    assert moved <= {ia, ib}, moved     # only a and b change files (one of them is displaced a level down in either order)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["host", "fail_alloc", "budget"])
def test_gpu_tiler_spills_its_pools_to_pinned_host_memory(mode):
    """A tiler whose pools find no device memory keeps them in mapped page-locked host memory: the node files are the
    oracle's, positions and attribute rows come back from the pools unchanged.  "host": from the first batch on;
    "fail_alloc": a growth of the position pool is refused in the second batch, so a pool with content MOVES; "budget":
    the context's device workspace may not pass a budget that the pools never fit into."""
    import schwarzwald_amd as swz
    import torch
    rng = np.random.default_rng(77)
    n, k = 150000, 3
    xyz = _points(rng, n, ODD, clustered=True)
    rgb = rng.integers(0, 255, size=(n, 3), dtype=np.uint8)
    gps = rng.random(n)
    sp = O.spacing_from_diagonal(*ODD, 32)
    ctx = swz.Context(0)  # a fresh workspace: the pools of an earlier tiler would simply be reused
    try:
        if mode == "host":
            ctx.set_option("SWZ_TILER_SPILL", "host")
        elif mode == "budget":
            ctx.set_option("SWZ_TILER_DEVICE_BUDGET_MB", "1")
        for sampler in (O.MIN_DISTANCE, O.GRID_CENTER):
            ex, c = _oracle_files(ODD, xyz, k, sampler, 300, sp, O.ACCURATE, 2)
            params = swz.TileParams(sampler=sampler, max_points_per_node=300, spacing_at_root=sp)
            with swz.Tiler(ctx, ODD[0], ODD[1], params, capacity_hint=1000) as t:
                parts = np.array_split(np.arange(n), k)
                for i, idx in enumerate(parts):
                    if mode == "fail_alloc" and i == 1:
                        ctx.set_option("SWZ_FAIL_ALLOC", "tiler_pool_xyz")
                    t.add_batch(xyz[idx], {"rgb": rgb[idx], "gps_time": gps[idx]})
                    ctx.set_option("SWZ_FAIL_ALLOC", None)
                t.finalize()
                dev_b, host_b = t.pool_residency()
                assert host_b >= n * 24, (dev_b, host_b)
                if mode == "host":
                    assert dev_b == 0
                # the node store follows the same rule: with a budget it never fits into, its sides live on the host too
                # (the merges stream through them over the host link); "host" leaves nothing of the tiler on the device
                sdev_b, shost_b = t.store_residency()
                if mode in ("host", "budget"):
                    assert shost_b > 0 and (sdev_b == 0 or mode == "budget"), (sdev_b, shost_b)
                info, table = t.info(), t.node_table()
                ns = int(info["num_stored"])
                d_keys = torch.empty(ns, dtype=torch.int64, device="cuda")
                d_ids = torch.empty(ns, dtype=torch.int32, device="cuda")
                d_lvl = torch.empty(ns, dtype=torch.int8, device="cuda")
                t.export_device(d_keys.data_ptr(), d_ids.data_ptr(), d_lvl.data_ptr())
                g = dict(table=table, ids=d_ids.cpu().numpy().view(np.uint32), level=d_lvl.cpu().numpy(), info=info)
                _compare(g, ex, c)
                # the pools by point id: clamped positions and the rows of the attribute column, read through the library
                p_xyz, p_attr = t.pools_device()
                got_xyz = ctx.copy_to_host(p_xyz, n * 24).view(np.float64).reshape(n, 3)
                got_rgb = ctx.copy_to_host(p_attr["rgb"], n * 3).reshape(n, 3)
                got_gps = ctx.copy_to_host(p_attr["gps_time"], n * 8).view(np.float64)
                lo, hi = np.array(ODD[0]), np.array(ODD[1])
                assert np.array_equal(got_xyz, np.clip(xyz, lo, hi)) and np.array_equal(got_rgb, rgb) and np.array_equal(got_gps, gps)
            if mode != "host":
                ctx.release_workspace()  # the next tiler starts on the device again
    finally:
        ctx.close()
