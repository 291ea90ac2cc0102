"""world_size-2 tests of the multi-GPU path over gloo.

CPU part (no GPU needed): the exchange plumbing -- owner map, send counts, all_to_all of row blocks -- moves
every point to the rank that owns its level-0 octant, keeping the order inside each (source, octant) block.
GPU part (-m gpu): both ranks share cuda:0 (collectives over gloo on CPU tensors), run the full sharded
tiler and the union of their shards must equal the oracle's single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)


def _cloud(n, seed):
    return O.generate_uniform(seed, n)


# ------------------------------------------------------------------ CPU: exchange plumbing
def _exchange_worker(rank, world, port, n, q, chunk_bytes=None):
    _init(rank, world, port)
    from schwarzwald_amd import sharded
    xyz = _cloud(n, 100 + rank)
    keys, _ = O.index_points(xyz, [0, 0, 0], [1, 1, 1])
    octant = (keys >> np.uint64(60)).astype(np.int64)
    order = np.argsort(octant, kind="stable")  # what swz_partition_by_octant_device produces
    counts = np.bincount(octant, minlength=8).tolist()
    send_counts = sharded.rank_send_counts(counts, world)
    rows = torch.from_numpy(xyz[order])
    buf, recv_counts = sharded.exchange_rows(rows, send_counts, headroom=3, chunk_bytes=chunk_bytes)
    q.put((rank, buf[3:].numpy(), recv_counts))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunk_bytes", [(2, None), (2, 24 * 257), (4, 24 * 100)])
def test_exchange_moves_points_to_octant_owner(world, chunk_bytes):
    """chunk_bytes forces the exchange into many rounds of grouped point-to-point transfers (the code path RCCL
    runs at full size); the result must not depend on it."""
    n = 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n, q, chunk_bytes)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, recv, rc = q.get(timeout=120)
        got[r] = (recv, rc)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from schwarzwald_amd import sharded
    clouds = [_cloud(n, 100 + r) for r in range(world)]
    for r in range(world):
        recv, rc = got[r]
        keys, _ = O.index_points(recv, [0, 0, 0], [1, 1, 1])
        owners = np.array([sharded.owner_of_octant(int(k >> np.uint64(60)), world) for k in keys])
        assert np.all(owners == r)
        # block from source s = that source's points owned by r, in (octant, original index) order
        off = 0
        for s in range(world):
            k, _ = O.index_points(clouds[s], [0, 0, 0], [1, 1, 1])
            octant = (k >> np.uint64(60)).astype(np.int64)
            order = np.argsort(octant, kind="stable")
            mine = order[np.array([sharded.owner_of_octant(int(o), world) == r for o in octant[order]])]
            assert rc[s] == len(mine)
            assert np.array_equal(recv[off:off + rc[s]], clouds[s][mine])
            off += rc[s]
    assert sum(len(got[r][0]) for r in range(world)) == world * n


def _gather_bytes_worker(rank, world, port, q):
    _init(rank, world, port)
    from schwarzwald_amd import sharded

    class Stub:  # what ShardedTiler._all_gather_bytes looks at
        group, device = None, torch.device("cpu")
    Stub.world = world
    mine = bytes([rank + 1]) * (40 + 7 * 0) if rank != 1 else b""   # rank 1 contributes nothing: padded with zeros
    q.put((rank, sharded.ShardedTiler._all_gather_bytes(Stub, mine)))
    dist.destroy_process_group()


def test_all_gather_of_byte_blobs_behind_the_joint_root_exchange():
    """The collective the library's IPC exchange runs on (swz_shard_joint_root_begin's callback): every rank's bytes in
    rank order; a rank with nothing to say reads as zeros of the common length."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_bytes_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [bytes([1]) * 40, bytes(40), bytes([3]) * 40]
    for r in range(world):
        assert got[r] == want


def test_exchange_plan_covers_every_row_exactly_once():
    """The per-rank plans of all ranks, executed together in one process, are an all-to-all."""
    from schwarzwald_amd import sharded
    rng = np.random.default_rng(5)
    for world, chunk in ((1, 7), (2, 3), (4, 5), (8, 1000), (8, 1)):
        counts = rng.integers(0, 23, (world, world))           # counts[s][d]: rows source s sends to d
        counts[rng.integers(0, world)][rng.integers(0, world)] = 0
        send = [np.arange(counts[s].sum()) + 1000 * s for s in range(world)]
        head = 2
        bufs = [np.full(head + counts[:, d].sum(), -1) for d in range(world)]
        plans = [sharded.exchange_plan(counts[r].tolist(), counts[:, r].tolist(), r, chunk, head) for r in range(world)]
        for r, ((s0, r0, cnt), rounds) in enumerate(plans):
            bufs[r][r0:r0 + cnt] = send[r][s0:s0 + cnt]
        nrounds = max(len(p[1]) for p in plans)
        for k in range(nrounds):
            # every send of round k must meet a receive of the same length in the peer's round k
            for r, (_, rounds) in enumerate(plans):
                if k >= len(rounds):
                    continue
                for peer, (ss, sl), (rs, rl) in rounds[k]:
                    assert sl <= chunk and rl <= chunk
                    if sl:
                        match = [op for op in plans[peer][1][k] if op[0] == r]
                        assert len(match) == 1 and match[0][2][1] == sl
                        o = match[0][2][0]
                        assert np.all(bufs[peer][o:o + sl] == -1)
                        bufs[peer][o:o + sl] = send[r][ss:ss + sl]
        for d in range(world):
            want = np.concatenate([send[s][counts[s][:d].sum():counts[s][:d].sum() + counts[s][d]] for s in range(world)])
            assert np.array_equal(bufs[d][head:], want) and np.all(bufs[d][:head] == -1)


def test_owner_map_is_monotone_and_balanced():
    from schwarzwald_amd import sharded
    for world in (1, 2, 4, 8):
        owners = [sharded.owner_of_octant(o, world) for o in range(8)]
        assert owners == sorted(owners) and set(owners) == set(range(world))
        assert sharded.rank_send_counts([1] * 8, world) == [8 // world] * world


_QUEUE_TIMEOUT = int(os.environ.get("SWZ_TEST_QUEUE_TIMEOUT", "300"))  # seconds a test waits for a worker's result


# ------------------------------------------------------------------ GPU: full sharded tiler, 2 ranks on one GPU
@pytest.mark.parametrize("n,concurrency,clustered", [(3000, 2, False), (200000, 2, False), (200000, 8, True), (50000, 32, True)])
def test_fast_start_level_from_summed_prefix_counts(n, concurrency, clustered):
    """FAST on a sharded batch takes its start level from the ranks' summed 6-octant prefix histograms
    (swz_fast_start_level_from_counts, a host function): it must be the level the single-process oracle derives from the
    whole batch (estimate_start_node_level_in_octree, TilingAlgorithms.cpp:1473-1535), however the points are dealt out."""
    import schwarzwald_amd as swz
    rng = np.random.default_rng(n + concurrency)
    xyz = _cloud(n, 77)
    if clustered:
        xyz[: n // 2] = 0.4 + 0.01 * rng.standard_normal((n // 2, 3))
        xyz = np.clip(xyz, 0.0, 1.0)
    keys, _ = O.index_points(xyz, [0, 0, 0], [1, 1, 1])
    parts = np.array_split(rng.permutation(n), 4)            # four "ranks", any points each
    counts = np.zeros(1 << 18, dtype=np.uint64)
    for part in parts:
        counts += np.bincount((keys[part] >> np.uint64(45)).astype(np.int64), minlength=1 << 18).astype(np.uint64)
    assert int(counts.sum()) == n
    got = swz.api.fast_start_level_from_counts(counts, concurrency)
    ref = O.tile(xyz, [0, 0, 0], [1, 1, 1], O.RANDOM_GRID, 300, O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 64), strategy=O.FAST,
                 fast_concurrency=concurrency)
    assert ref["status"] == 0
    assert got == ref["stats"]["fast_start_levels"]


def _corner_cloud(n, seed, world):
    """Points in the octants of rank 0 only: every other rank's shard is empty (flat terrain in a cubic
    root box leaves whole octants, i.e. whole ranks, without points)."""
    xyz = _cloud(n, seed)
    xyz[:, 0] *= 0.5            # x < 0.5: octants 0..3
    if world > 2:
        xyz[:, 1] *= 0.5        # and y < 0.5: octants 0, 1
    return xyz


def _tile_worker(rank, world, port, n, sampler, max_pts, spacing, q, corner=False, backend="gloo", joint=False, flags=0):
    # the joint root is the default (after a collective probe of the IPC mappings); "0" keeps the chain of ghosts
    # (joint = None: the variable is left alone -- the default must sweep the root jointly after its probe)
    if joint is None:
        os.environ.pop("SWZ_SHARD_JOINT_ROOT", None)
        joint = True
    else:
        os.environ["SWZ_SHARD_JOINT_ROOT"] = "1" if joint else "0"
    import schwarzwald_amd as swz
    from schwarzwald_amd import sharded
    # gloo: all ranks share GPU 0 (what a one-GPU box can run); nccl (= RCCL): one GPU per rank
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    _init(rank, world, port, backend)
    ctx = swz.Context(dev.index)
    params = swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing, flags=flags)
    xyz = torch.from_numpy(_corner_cloud(n, 300 + rank, world) if corner else _cloud(n, 300 + rank)).to(dev)
    tiler = sharded.ShardedTiler(ctx, dev, [0, 0, 0], [1, 1, 1], params)
    assert tiler.joint_root == joint
    stats = tiler.tile(xyz)
    assert tiler.used_joint_root == (joint and sampler == swz.MIN_DISTANCE)
    assert stats["root_mode"] == ("joint" if tiler.used_joint_root else ("chain" if sampler == swz.MIN_DISTANCE else "local"))
    assert set(tiler.stage_timings()) >= {"exchange_ms", "root_ms", "levels_ms"}
    recv, keys, perm, level = tiler.result
    q.put((rank, recv.cpu().numpy(), keys.cpu().numpy().view(np.uint64), perm.cpu().numpy().view(np.uint32),
           level.cpu().numpy(), stats))
    ctx.close()
    dist.destroy_process_group()


def _visible_gpus():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


# (sampler, diagonal fraction, points per rank, backend): MIN_DISTANCE at d = 60 (dense sweeps at this size) and at the
# bench's d = 250; the last case runs the exchange over RCCL with one GPU per rank and enables itself where two GPUs are
# visible (this pool's boxes have one)
SHARDED_CASES = [(O.RANDOM_GRID, 250, 60000, "gloo"), (O.GRID_CENTER, 250, 60000, "gloo"), (O.MIN_DISTANCE, 60, 60000, "gloo"),
                 (O.MIN_DISTANCE, 250, 150000, "gloo"), (O.JITTERED, 250, 60000, "gloo"), (O.MIN_DISTANCE, 250, 150000, "nccl"),
                 # the MIN_DISTANCE root swept by both ranks at once, each reading the other's root arrays through IPC mappings
                 (O.MIN_DISTANCE, 60, 60000, "gloo+joint"), (O.MIN_DISTANCE, 250, 150000, "gloo+joint"),
                 # ... which is what a driver gets that sets nothing (probe of the IPC mappings, then the joint sweep)
                 (O.MIN_DISTANCE, 250, 150000, "gloo+default")]


@pytest.mark.gpu
@pytest.mark.parametrize("sampler,d,n,backend", SHARDED_CASES)
def test_sharded_tile_matches_oracle(sampler, d, n, backend):
    if backend == "nccl" and _visible_gpus() < 2:
        pytest.skip("the RCCL exchange needs one GPU per rank: fewer than two GPUs visible")
    joint = None if backend.endswith("+default") else backend.endswith("+joint")
    backend = backend.split("+")[0]
    world, max_pts = 2, 500
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], d)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, n, sampler, max_pts, spacing, q, False, backend, joint))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=_QUEUE_TIMEOUT)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process oracle on the union of all points
    union = np.vstack([_cloud(n, 300 + r) for r in range(world)])
    ref = O.tile(union, [0, 0, 0], [1, 1, 1], sampler, max_pts, spacing)
    assert ref["status"] == 0
    ref_xyz = ref["xyz_clamped"][ref["perm"]]  # positions in Morton order
    # concatenating the shards in rank order gives the global Morton order (ties aside: compare as multisets
    # of (key, level, position) which is order independent)
    keys = np.concatenate([got[r][1] for r in range(world)])
    level = np.concatenate([got[r][3] for r in range(world)])
    pos = np.vstack([got[r][0][got[r][2]] for r in range(world)])
    assert np.array_equal(keys, ref["keys"])

    def canon(k, lv, p):
        rec = np.rec.fromarrays([k, lv, p[:, 0], p[:, 1], p[:, 2]], names="k,l,x,y,z")
        return np.sort(rec, order=["k", "x", "y", "z", "l"])

    a, b = canon(keys, level, pos), canon(ref["keys"], ref["level"], ref_xyz)
    assert np.array_equal(a, b)
    assert sum(got[r][4]["num_nodes"] for r in range(world)) - (world - 1) == ref["stats"]["num_nodes"]


@pytest.mark.gpu
@pytest.mark.parametrize("joint", [False, True])
def test_sharded_property_mode_keeps_spacing_and_maximality(joint):
    """SWZ_FLAG_MIN_DISTANCE_PROPERTY on a sharded batch: the root, which spans the ranks, is sampled exactly (in turns or by
    all ranks at once), the levels below in property mode.  The union must have the property on every sampled node: no two
    taken points closer than the node's spacing, every point handed down closer than that to a taken one."""
    from test_min_distance_property import _check_property
    import schwarzwald_amd as swz
    world, n, d, max_pts = 2, 150000, 250, 500
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], d)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, n, O.MIN_DISTANCE, max_pts, spacing, q, False, "gloo", joint,
                                                    swz.FLAG_MIN_DISTANCE_PROPERTY)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=_QUEUE_TIMEOUT)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    keys = np.concatenate([got[r][1] for r in range(world)])
    level = np.concatenate([got[r][3] for r in range(world)])
    pos = np.vstack([got[r][0][got[r][2]] for r in range(world)])
    assert np.all(keys[1:] >= keys[:-1])
    a, b = _check_property(keys, level, pos, spacing, max_pts, int(level.max()))
    assert a > 0 and b > 0
    # the root is the exact one: what the single-process oracle takes there
    union = np.vstack([_cloud(n, 300 + r) for r in range(world)])
    ref = O.tile(union, [0, 0, 0], [1, 1, 1], O.MIN_DISTANCE, max_pts, spacing)
    assert np.array_equal(np.sort(keys[level == -1]), np.sort(ref["keys"][ref["level"] == -1]))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED, "MIN_DISTANCE+joint"])
def test_sharded_tile_with_empty_shards(sampler, world):
    """Ranks whose octants hold no point must neither fail nor hang the others (all ranks share cuda:0 here) -- also when
    the MIN_DISTANCE root is swept by all ranks at once: a rank without points meets the others in the exchange."""
    joint = sampler == "MIN_DISTANCE+joint"
    sampler = O.MIN_DISTANCE if joint else sampler
    n, max_pts = 30000, 500
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250 if sampler != O.MIN_DISTANCE else 60)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, n, sampler, max_pts, spacing, q, True, "gloo", joint))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=_QUEUE_TIMEOUT)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert got[r][0].shape[0] == 0 and got[r][4]["points_visited"] == 0
    union = np.vstack([_corner_cloud(n, 300 + r, world) for r in range(world)])
    ref = O.tile(union, [0, 0, 0], [1, 1, 1], sampler, max_pts, spacing)
    assert ref["status"] == 0
    assert np.array_equal(got[0][1], ref["keys"])
    ref_xyz = ref["xyz_clamped"][ref["perm"]]
    pos = got[0][0][got[0][2]]
    rec = lambda k, lv, p: np.sort(np.rec.fromarrays([k, lv, p[:, 0], p[:, 1], p[:, 2]], names="k,l,x,y,z"),
                                   order=["k", "x", "y", "z", "l"])
    assert np.array_equal(rec(got[0][1], got[0][3], pos), rec(ref["keys"], ref["level"], ref_xyz))


def _fast_tile_worker(rank, world, port, n, sampler, max_pts, spacing, concurrency, q, corner):
    import schwarzwald_amd as swz
    from schwarzwald_amd import sharded
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _init(rank, world, port, "gloo")
    ctx = swz.Context(0)
    params = swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing, strategy=swz.FAST,
                            fast_concurrency=concurrency)
    xyz = torch.from_numpy(_corner_cloud(n, 300 + rank, world) if corner else _cloud(n, 300 + rank)).to(dev)
    tiler = sharded.ShardedTiler(ctx, dev, [0, 0, 0], [1, 1, 1], params)
    stats = tiler.tile(xyz)
    recv, keys, perm, level, dup = tiler.result
    q.put((rank, recv.cpu().numpy(), keys.cpu().numpy().view(np.uint64), perm.cpu().numpy().view(np.uint32),
           level.cpu().numpy(), dup.cpu().numpy().view(np.uint32), stats))
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
@pytest.mark.parametrize("world,corner", [(2, False), (4, False), (4, True)])
def test_sharded_single_batch_fast_strategy_matches_oracle(sampler, world, corner):
    """FAST (TilingAlgorithmV3, the reference's default) through ShardedTiler.tile: the start level from the ranks' summed
    prefix histograms, the skipped levels rebuilt per rank, the root from all ranks' level-0 nodes on rank 0.  Level and
    the ancestors a point is stored in as well (dup) must be the single-process oracle's, also with ranks that own nothing."""
    n, max_pts, concurrency = 60000, 300, 2
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 60 if sampler == O.MIN_DISTANCE else 250)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fast_tile_worker, args=(r, world, port, n, sampler, max_pts, spacing, concurrency, q, corner))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=_QUEUE_TIMEOUT)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    union = np.vstack([(_corner_cloud(n, 300 + r, world) if corner else _cloud(n, 300 + r)) for r in range(world)])
    ref = O.tile(union, [0, 0, 0], [1, 1, 1], sampler, max_pts, spacing, strategy=O.FAST, fast_concurrency=concurrency)
    assert ref["status"] == 0 and ref["stats"]["fast_start_levels"] >= 1
    ref_xyz = ref["xyz_clamped"][ref["perm"]]
    keys = np.concatenate([got[r][1] for r in range(world)])
    level = np.concatenate([got[r][3] for r in range(world)])
    dup = np.concatenate([got[r][4] for r in range(world)])
    pos = np.vstack([got[r][0][got[r][2]] for r in range(world)])
    assert np.array_equal(keys, ref["keys"])
    assert all(got[r][5]["fast_start_levels"] == ref["stats"]["fast_start_levels"] for r in range(world))

    def canon(k, lv, du, p):
        rec = np.rec.fromarrays([k, lv, du, p[:, 0], p[:, 1], p[:, 2]], names="k,l,d,x,y,z")
        return np.sort(rec, order=["k", "x", "y", "z", "l", "d"])

    assert np.array_equal(canon(keys, level, dup, pos), canon(ref["keys"], ref["level"], ref["dup"], ref_xyz))
    assert int((dup & 1).sum()) > 0  # the reconstructed root holds points


# ------------------------------------------------------------------ GPU: sharded AND multi-batch (BASELINE config 5's shape)
def _mb_cloud(n, batch, rank, corner_world=0):
    xyz = _cloud(n, 500 + 10 * batch + rank)
    if corner_world:          # rank 0's octants only: every other shard stays empty
        xyz[:, 0] *= 0.5
        if corner_world > 2:
            xyz[:, 1] *= 0.5
    return xyz


def _mb_worker(rank, world, port, n, k, sampler, max_pts, spacing, q, corner, strategy=0, concurrency=8, root="default"):
    # the MIN_DISTANCE root of every batch: swept by all ranks at once (the default, after the IPC probe) or in turns
    # ("spilled": rank 0 -- whose root arrays the higher ranks map -- keeps its pools in page-locked host memory, which has
    # no IPC handle: the per-batch vote must then take the chain on EVERY rank)
    if root == "chain":
        os.environ["SWZ_SHARD_JOINT_ROOT"] = "0"
    else:
        os.environ.pop("SWZ_SHARD_JOINT_ROOT", None)
    _init(rank, world, port)
    import schwarzwald_amd as swz
    from schwarzwald_amd import sharded
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = swz.Context(0)
    if root == "spilled" and rank == 0:
        ctx.set_option("SWZ_TILER_SPILL", "host")
    params = swz.TileParams(sampler=sampler, max_points_per_node=max_pts, spacing_at_root=spacing, strategy=strategy,
                            fast_concurrency=concurrency)
    st = sharded.ShardedBatchTiler(ctx, dev, [0, 0, 0], [1, 1, 1], params)
    for b in range(k):
        xyz = torch.from_numpy(_mb_cloud(n, b, rank, world if corner else 0)).to(dev)
        ids = torch.arange(n, dtype=torch.int64) + (b * world * n + rank * n)     # id in the single-process batch order
        attrs = {"intensity": (ids & 0xFFFF).to(torch.int16).to(dev), "point_source_id": (ids >> 16).to(torch.int16).to(dev)}
        bstats = st.add_batch(xyz, attrs)
        if sampler == swz.MIN_DISTANCE and strategy == 0 and bstats["root_mode"] != "local":
            want_mode = {"chain": "chain", "spilled": "chain (a pool lives in host memory)"}.get(root, "joint")
            assert bstats["root_mode"] == want_mode, bstats["root_mode"]
    st.finalize()
    info = st.tiler.info()
    table = st.tiler.node_table()
    ns = int(info["num_stored"])
    d_ids = torch.empty(max(ns, 1), dtype=torch.int32, device=dev)
    st.tiler.export_device(0, d_ids.data_ptr(), 0)
    local = d_ids.cpu().numpy().view(np.uint32)[:ns].astype(np.int64)
    _, pools = st.tiler.pools_device()
    npts = int(info["num_points"])
    lo = ctx.copy_to_host(pools["intensity"], 2 * npts).view(np.uint16).astype(np.int64) if npts else np.zeros(0, np.int64)
    hi = ctx.copy_to_host(pools["point_source_id"], 2 * npts).view(np.uint16).astype(np.int64) if npts else np.zeros(0, np.int64)
    gids = (lo | (hi << 16))[local] if ns else np.zeros(0, np.int64)
    q.put((rank, {kk: np.asarray(v) for kk, v in table.items()}, gids, npts))
    st.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED, "MIN_DISTANCE+chain",
                                     "MIN_DISTANCE+spilled"])
@pytest.mark.parametrize("corner", [False, True])
def test_sharded_multibatch_matches_the_multibatch_oracle(sampler, corner):
    """Two ranks (sharing cuda:0, collectives over gloo), three batches each: the union of the shards' node files must
    be the single-process multi-batch oracle's, file by file and in file order; the root's file is the concatenation of
    the shards' parts in rank order.  corner: rank 1 never owns a point."""
    root = sampler.split("+")[1] if isinstance(sampler, str) else "default"   # (MIN_DISTANCE alone: every batch's root swept jointly)
    sampler = O.MIN_DISTANCE if root != "default" else sampler
    world, n, k, max_pts = 2, 20000, 3, 400
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250 if sampler != O.MIN_DISTANCE else 60)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_mb_worker, args=(r, world, port, n, k, sampler, max_pts, spacing, q, corner, 0, 8, root)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=_QUEUE_TIMEOUT)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    oracle = O.Tiler([0, 0, 0], [1, 1, 1], sampler, max_pts, spacing)
    for b in range(k):
        assert oracle.add_batch(np.vstack([_mb_cloud(n, b, r, world if corner else 0) for r in range(world)])) == 0
    assert oracle.finalize() == 0
    ex = oracle.export()
    want = {(int(l), int(key)): ex["ids"][int(o):int(o + c)].astype(np.int64)
            for l, key, o, c in zip(ex["level"], ex["key"], ex["offset"], ex["count"])}
    have = {}
    for r in range(world):
        table, gids, npts = got[r]
        for l, key, o, c in zip(table["level"], table["key"], table["offset"], table["count"]):
            have.setdefault((int(l), int(key)), []).append(gids[int(o):int(o + c)])
    assert sum(got[r][2] for r in range(world)) == world * n * k
    if corner:
        assert got[1][2] == 0
    assert set(have) == set(want)
    for node, parts in have.items():
        assert node[0] == -1 or len(parts) == 1          # only the root spans shards
        assert np.array_equal(np.concatenate(parts), want[node]), node


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [O.RANDOM_GRID, O.GRID_CENTER, O.MIN_DISTANCE, O.JITTERED])
@pytest.mark.parametrize("k", [1, 3])
def test_sharded_fast_strategy_matches_the_multibatch_oracle(sampler, k):
    """FAST (TilingAlgorithmV3, the reference's default) on two ranks: the start level from the summed prefix histograms of
    the first batch, the levels below it per rank, the skipped levels rebuilt at the end -- the root on rank 0 from the
    level-0 files of both ranks.  Node files = the single-process FAST oracle's, file by file and in file order."""
    world, n, max_pts, conc = 2, 20000, 400, 2
    spacing = O.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250 if sampler != O.MIN_DISTANCE else 60)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_mb_worker, args=(r, world, port, n, k, sampler, max_pts, spacing, q, False, O.FAST, conc)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=_QUEUE_TIMEOUT)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    oracle = O.Tiler([0, 0, 0], [1, 1, 1], sampler, max_pts, spacing, strategy=O.FAST, fast_concurrency=conc)
    for b in range(k):
        assert oracle.add_batch(np.vstack([_mb_cloud(n, b, r, 0) for r in range(world)])) == 0
    assert oracle.finalize() == 0
    ex = oracle.export()
    want = {(int(l), int(key)): ex["ids"][int(o):int(o + c)].astype(np.int64)
            for l, key, o, c in zip(ex["level"], ex["key"], ex["offset"], ex["count"])}
    have = {}
    for r in range(world):
        table, gids, npts = got[r]
        for l, key, o, c in zip(table["level"], table["key"], table["offset"], table["count"]):
            have.setdefault((int(l), int(key)), []).append(gids[int(o):int(o + c)])
    assert set(have) == set(want)
    for node, parts in have.items():
        assert node[0] == -1 or len(parts) == 1          # only the root spans shards
        assert np.array_equal(np.concatenate(parts), want[node]), node
