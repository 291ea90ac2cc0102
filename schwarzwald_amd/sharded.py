"""Multi-GPU driver: one process per GPU, points sharded by their level-0 octant (top 3 Morton bits).

The reference is a single process; sharding is this implementation's own addition.
Every node at level >= 0 lives on exactly one rank, so after ONE exchange step (all-to-all of the point rows,
RCCL over xGMI when the process group is "nccl") each rank tiles its own octants without communication.  Only
the root node spans ranks:
  * its take-all/sample decision uses the global point count (one tiny all-reduce);
  * RANDOM_GRID / GRID_CENTER / JITTERED decide per grid cell and cells never straddle octants, so every rank
    samples its slice of the root locally;
  * MIN_DISTANCE at the root is the greedy sweep in Morton order, which visits the octants in rank order: rank r
    waits for the root samples of the lower ranks ("ghosts", a few MB), sweeps its slice, and passes its own on.
    Index + sort of the local points happen before that on all ranks at once (shard_presort_device), so the
    chain holds the root sweeps only.  Levels >= 0 then run concurrently on all ranks.

torch is used for device memory, index_select and torch.distributed only.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import api


def owner_of_octant(octant, world):
    """Contiguous blocks of octants per rank, so lower ranks own lower Morton codes (world in {1,2,4,8})."""
    return octant * world // 8


def rank_send_counts(octant_counts, world):
    """Points each rank receives from this rank, from this rank's 8 octant counts."""
    out = [0] * world
    for o, c in enumerate(octant_counts):
        out[owner_of_octant(o, world)] += int(c)
    return out


GHOST_HEADROOM = 4 << 20  # rows reserved in front of the received points for MIN_DISTANCE root ghosts


EXCHANGE_CHUNK_BYTES = 64 << 20  # per (source, destination) pair and round of the exchange


def exchange_plan(send_counts, recv_counts, rank, chunk_rows, headroom=0):
    """Rounds of the exchange for one rank: a list of rounds, each a list of (peer, send_slice, recv_slice) with
    slices as (first_row, rows) into the send rows / the receive buffer; at most chunk_rows rows per pair and
    round, empty transfers left out.  The rank's own block is not part of any round (it is copied directly):
    own_copy = (first send row, first buffer row, rows).  All ranks derive the same number of rounds for a pair
    from the same two counts, so sends and receives match up."""
    world = len(send_counts)
    send_off = [sum(send_counts[:r]) for r in range(world)]
    recv_off = [headroom + sum(recv_counts[:r]) for r in range(world)]
    own_copy = (send_off[rank], recv_off[rank], send_counts[rank])
    most = max([max(send_counts[r], recv_counts[r]) for r in range(world) if r != rank] + [0])
    rounds = []
    for k in range((most + chunk_rows - 1) // chunk_rows):
        ops = []
        for r in range(world):
            if r == rank:
                continue
            s_len = min(chunk_rows, max(0, send_counts[r] - k * chunk_rows))
            r_len = min(chunk_rows, max(0, recv_counts[r] - k * chunk_rows))
            if s_len or r_len:
                ops.append((r, (send_off[r] + k * chunk_rows, s_len), (recv_off[r] + k * chunk_rows, r_len)))
        rounds.append(ops)
    return own_copy, rounds


def exchange_rows(rows, send_counts, group=None, headroom=0, chunk_bytes=None):
    """All-to-all of row blocks.  rows: [n, k] tensor already grouped by destination rank; send_counts[r] rows go
    to rank r.  Returns (buffer, recv_counts): the received rows, ordered by source rank, are buffer[headroom:]
    (the first `headroom` rows are left free so that a few rows can later be put right in front).

    The rows that stay on this rank are copied directly; the others travel as grouped point-to-point transfers
    (one RCCL group per round) of at most chunk_bytes per pair and round.  One big all_to_all_single is avoided
    on purpose: RCCL 2.26 (ROCm 7.0) was measured to drop the second half of a single large block (1.9 GB to
    self: the rows from n/2 on never arrive; tools/a2a_probe.py), and bounded messages bound its staging memory."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = rows.device
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    sc = torch.tensor(send_counts, dtype=torch.int64, device=comm_dev)
    rc = torch.empty(world, dtype=torch.int64, device=comm_dev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(v) for v in rc.tolist()]
    send_counts = [int(v) for v in send_counts]
    m = sum(recv_counts)
    buf = torch.empty((headroom + m,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=dev)
    rows = rows.contiguous()
    row_bytes = (rows[0:1].numel() if rows.shape[0] else 1) * rows.element_size()
    chunk = max(1, int(chunk_bytes or EXCHANGE_CHUNK_BYTES) // max(1, row_bytes))
    (s0, r0, cnt), rounds = exchange_plan(send_counts, recv_counts, rank, chunk, headroom)
    buf[r0:r0 + cnt].copy_(rows[s0:s0 + cnt])
    if comm_dev != dev and rounds:  # gloo with device tensors (tests): the transfers go through host memory
        src, dst = rows.to(comm_dev), torch.empty((headroom + m,) + tuple(rows.shape[1:]), dtype=rows.dtype)
    else:
        src, dst = rows, buf
    for ops in rounds:
        p2p = []
        for peer, (ss, sl), (rs, rl) in ops:
            if sl:
                p2p.append(dist.P2POp(dist.isend, src[ss:ss + sl], peer, group))
            if rl:
                p2p.append(dist.P2POp(dist.irecv, dst[rs:rs + rl], peer, group))
        if not p2p:
            continue  # this rank's pairs are done; others may still have rounds
        for req in dist.batch_isend_irecv(p2p):
            req.wait()
    if dst is not buf:
        for r in range(world):
            if r != rank and recv_counts[r]:
                o = headroom + sum(recv_counts[:r])
                buf[o:o + recv_counts[r]].copy_(dst[o:o + recv_counts[r]])
    return buf, recv_counts


class _Stages:
    """Time stamps at the stage boundaries of a sharded batch: events on the stream the library runs on (no
    synchronisation inside the batch), read after it -- what bench.py reports per rank."""

    def __init__(self, dev):
        self.dev, self.marks = dev, []
        self.mark("start")

    def mark(self, name):
        if self.dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.dev))
            self.marks.append((name, ev, time.perf_counter()))
        else:
            self.marks.append((name, None, time.perf_counter()))

    def ms(self):
        """{stage: milliseconds} (device time between the marks; host time where the stage waited on the host)"""
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        out = {}
        for (_, e0, t0), (name, e1, t1) in zip(self.marks[:-1], self.marks[1:]):
            dev_ms = e0.elapsed_time(e1) if e0 is not None else 0.0
            out[name] = round(max(dev_ms, (t1 - t0) * 1e3 if e0 is None else dev_ms), 3)
            out[name + "_host"] = round((t1 - t0) * 1e3, 3)
        return out


def _trace(dev, what, t0):
    """SWZ_DEBUG=1: stage timings on stderr (synchronises the device, debugging only)."""
    if os.environ.get("SWZ_DEBUG"):
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        sys.stderr.write("[swz sharded] %-28s %8.1f ms\n" % (what, (time.perf_counter() - t0) * 1e3))
        sys.stderr.flush()
    return time.perf_counter()


def _all_gather_bytes(mine, group, device, world):
    """every rank's bytes in rank order (equal lengths); the collective behind swz_shard_joint_root_begin / _probe"""
    on_gpu = dist.get_backend(group) == "nccl"
    n = torch.tensor([len(mine)], dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    size = int(n.item())
    t = torch.zeros(max(size, 1), dtype=torch.uint8)
    if mine:
        t[:len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    if on_gpu:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()[:size]) for o in out]


def _joint_root_usable(ctx, rank, world, gather):
    """The collective probe before the first joint root: can every rank map the lower ranks' device memory?"""
    try:
        usable = ctx.shard_joint_root_probe(rank, world, gather)
    except api.SwzError:
        usable = False  # (the probe's own collectives ran on every rank: all of them land here together or not at all)
    if os.environ.get("SWZ_DEBUG"):
        sys.stderr.write("[swz sharded] rank %d: device memory of the other ranks %s be mapped: MIN_DISTANCE root %s\n" % (
            rank, "can" if usable else "cannot", "swept by all ranks at once" if usable else "in turns"))
    return usable


def _joint_root_step(ctx, rank, world, group, gather, begin, failure):
    """The MIN_DISTANCE root of a sharded batch swept by all ranks at once (include/swz_gpu.h, swz_shard_joint_root_*):
    begin() is the library call that samples this rank's root (swz_shard_begin_device / swz_tiler_shard_begin_device,
    without ghosts); a rank whose call never reached the sweep -- no points, or an error -- meets the others here."""
    entered = False
    if not failure:
        try:
            ctx.shard_joint_root_begin(rank, world, gather)
            entered = True
        except api.SwzError as e:
            failure.append(e)
    result = None
    if entered:
        if not failure:
            try:
                result = begin()
            except api.SwzError as e:
                failure.append(e)
        try:  # (a rank without points, or one that failed before its sweep met the others, meets them here)
            ctx.shard_joint_root_meet(ok=not failure)
        except api.SwzError as e:
            failure.append(e)
    else:  # the two exchanges of a batch are collectives: take part (an all-zero blob reads as a shard without points)
        gather(b"")
        gather(b"")
    dist.barrier(group=group)  # the others may read this rank's root arrays until they are done
    # (the mappings are closed one rank at a time: processes that unmap the same allocation at the same moment were seen to
    # abort inside the runtime -- see swz_shard_joint_root_probe)
    for r in range(world):
        if r == rank:
            try:
                ctx.shard_joint_root_end()
            except api.SwzError as e:
                failure.append(e)
        dist.barrier(group=group)
    return result


class ShardedTiler:
    """Tiles one batch whose points are spread over the ranks of a process group."""

    def __init__(self, ctx, device, bmin, bmax, params, group=None):
        self.ctx, self.device, self.bmin, self.bmax, self.params, self.group = ctx, device, bmin, bmax, params, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world not in (1, 2, 4, 8):
            raise ValueError("world size must be 1, 2, 4 or 8 (octants are dealt out in contiguous blocks)")
        self.result = None
        self._keepalive = None
        self._batches = 0
        # The MIN_DISTANCE root swept by all ranks at once (IPC mappings of the lower ranks' root arrays) instead of the
        # chain of ghosts from rank to rank, whose time grows with the number of ranks.  On by default
        # (SWZ_SHARD_JOINT_ROOT=0: the chain); before the first batch that would use it the ranks PROBE whether they can map
        # each other's memory at all -- a collective with a vote -- and keep the chain when one of them cannot.
        self.joint_root = os.environ.get("SWZ_SHARD_JOINT_ROOT", "1") not in ("", "0")
        self._joint_usable = None  # unknown until probed
        self.root_mode = "local"   # what the last batch did: local | chain | joint (| chain (...reason))
        self.stage_ms = {}
        if device.type == "cuda":
            # the context otherwise runs on its own non-blocking stream: torch's kernels (grouping, exchange) and the
            # library's must be ordered, so both use torch's current stream
            ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def _bcast(self, tensor, src):
        backend = dist.get_backend(self.group)
        if backend == "nccl" or tensor.device.type == "cpu":
            dist.broadcast(tensor, src, group=self.group)
            return tensor
        t = tensor.cpu()
        dist.broadcast(t, src, group=self.group)
        tensor.copy_(t)
        return tensor

    def _all_gather_bytes(self, mine):
        return _all_gather_bytes(mine, self.group, self.device, self.world)

    def tile(self, xyz):
        """xyz: [n, 3] float64 tensor on this rank's GPU (any points of the batch).  Returns the tile stats of this
        rank's shard; self.result holds (recv_xyz, keys, perm, level) for the points this rank owns."""
        ctx, dev, world = self.ctx, self.device, self.world
        n = xyz.shape[0]
        # the previous batch's buffers are the caller's no longer
        self.result = None
        self._keepalive = None
        if dev.type == "cuda":
            # keys + perm + send buffer + receive buffer are about 2.5 x the positions; when a previous batch left
            # a workspace that does not leave room for them, give it back first.  Re-growing it costs seconds
            # (about 25 ms per GB), so this is a last resort, not the steady state.
            need = int(2.6 * n * 24)
            cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)  # torch can reuse these
            free, _ = torch.cuda.mem_get_info(dev)
            if free + cached < need:
                torch.cuda.empty_cache()
                free, _ = torch.cuda.mem_get_info(dev)
                if free < need and ctx.workspace_bytes() > 0:
                    ctx.release_workspace()
                    if os.environ.get("SWZ_DEBUG"):
                        sys.stderr.write("[swz sharded] workspace released: %.1f GB were free\n" % (free / 1e9))
        t0 = time.perf_counter()
        stages = _Stages(dev)
        # 1. encode locally, group by destination
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.morton_encode_device(xyz.data_ptr(), n, self.bmin, self.bmax, keys.data_ptr())
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        octant_counts = ctx.partition_by_octant_device(keys.data_ptr(), n, perm.data_ptr())
        del keys
        send_counts = rank_send_counts(octant_counts, world)
        t0 = _trace(dev, "encode + partition", t0)
        send = xyz.index_select(0, perm.long())
        del perm
        t0 = _trace(dev, "group rows by destination", t0)
        stages.mark("encode_partition_group_ms")
        # 2. the one exchange step
        total = torch.tensor([n], dtype=torch.int64, device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(total, group=self.group)
        global_points = int(total.item())
        sequential_root = (self.params.strategy != api.FAST and self.params.sampler == api.MIN_DISTANCE and
                           global_points > self.params.max_points_per_node)
        headroom = GHOST_HEADROOM if (sequential_root and self.rank > 0) else 0
        buf, recv_counts = exchange_rows(send, send_counts, self.group, headroom)
        recv_counts_total = sum(recv_counts)
        del send
        t0 = _trace(dev, "exchange", t0)
        stages.mark("exchange_ms")
        if dev.type == "cuda" and self._batches == 0:
            # first batch: the context is about to grow its workspace with hipMalloc (up to ~160 B per point for
            # MIN_DISTANCE); hand torch's cached send buffers back only when that would not fit otherwise --
            # giving them back costs the next batch ~0.7 s of re-allocation
            free, _ = torch.cuda.mem_get_info(dev)
            if free < int(7.0 * max(n, recv_counts_total) * 24):
                torch.cuda.empty_cache()
        self._batches += 1
        recv = buf[headroom:]
        m = recv.shape[0]
        # A failing library call on ONE rank must not strand the others in the collectives below: the rank keeps
        # taking part (with nothing to contribute) and the failure is raised on all ranks together at the end.
        failure = []

        def guarded(fn, default):
            if failure:
                return default
            try:
                return fn()
            except api.SwzError as e:
                failure.append(e)
                return default

        if self.params.strategy == api.FAST:
            return self._tile_fast(buf, recv, m, global_points, failure, guarded, stages, t0)
        # 3. root node.  m == 0 (this rank's octants are empty, e.g. the upper half of a cubic box around flat
        # terrain) is an ordinary shard: the library takes nothing of the root and reports zero points.
        possible = (sequential_root and world > 1 and self.joint_root and dev.type == "cuda" and
                    ctx.shard_joint_root_possible(self.bmin, self.bmax, self.params))
        joint = possible
        if joint and self._joint_usable is None:
            self._joint_usable = _joint_root_usable(ctx, self.rank, world, self._all_gather_bytes)
        joint = bool(joint and self._joint_usable)
        self.used_joint_root = joint
        self.root_mode = "local" if not sequential_root else ("joint" if joint else ("chain (no IPC mapping)" if possible else "chain"))
        if not sequential_root:
            guarded(lambda: ctx.shard_begin_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params, global_points), 0)
        elif joint:
            # All ranks sweep the root cells of their own octants at once; cells at the face of a lower octant read that
            # rank's records in place through IPC mappings (include/swz_gpu.h, swz_shard_joint_root_*): no ghosts, no turns.
            _joint_root_step(ctx, self.rank, world, self.group, self._all_gather_bytes,
                             lambda: ctx.shard_begin_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params, global_points), failure)
        else:
            if m > 0:
                # everything that does not depend on the ghosts happens on all ranks at once; only the root
                # node itself is left in the chain below
                guarded(lambda: ctx.shard_presort_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params,
                                                         GHOST_HEADROOM), None)
            ghosts = []
            for r in range(world):
                cnt = torch.zeros(1, dtype=torch.int64, device=dev)
                mine = None
                if r == self.rank:
                    g = sum(t.shape[0] for t in ghosts)
                    if g == 0 or m == 0:
                        gptr, g = None, 0
                    elif g <= headroom:  # ghosts go right in front of the received points: no staging copy
                        torch.cat(ghosts, out=buf[headroom - g:headroom])
                        gptr = buf[headroom - g:].data_ptr()
                    else:
                        gbuf = torch.cat(ghosts)
                        gptr = gbuf.data_ptr()
                    taken = guarded(lambda: ctx.shard_begin_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params,
                                                                   global_points, gptr, g), 0)
                    mine = torch.empty((taken, 3), dtype=torch.float64, device=dev)
                    if taken:
                        guarded(lambda: ctx.shard_root_taken_device(mine.data_ptr()), None)
                    cnt[0] = taken
                if r == world - 1:
                    break  # nobody owns higher octants
                self._bcast(cnt, r)
                b = mine if r == self.rank else torch.empty((int(cnt.item()), 3), dtype=torch.float64, device=dev)
                if b.shape[0]:
                    self._bcast(b, r)
                if self.rank > r:
                    ghosts.append(b)
        t0 = _trace(dev, "root node", t0)
        stages.mark("root_ms")
        # 4. everything below the root is local
        okeys = torch.empty(m, dtype=torch.int64, device=dev)
        operm = torch.empty(m, dtype=torch.int32, device=dev)
        olevel = torch.empty(m, dtype=torch.int8, device=dev)
        zero = dict(num_nodes=0, points_visited=0, max_level=-1, fast_start_levels=-1, num_levels=0, min_distance_rounds=0)
        stats = guarded(lambda: ctx.shard_finish_device(okeys.data_ptr(), operm.data_ptr(), olevel.data_ptr()), zero)
        _trace(dev, "levels below the root", t0)
        stages.mark("levels_ms")
        self._stages = stages
        flag = torch.tensor([1 if failure else 0], dtype=torch.int64,
                            device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            if failure:
                raise failure[0]
            raise api.SwzError(api.ERR_PEER_FAILED, "another rank failed to tile its shard of this batch")
        self.result = (recv, okeys, operm, olevel)
        self._keepalive = buf  # the context reads the points until shard_finish returned
        stats["shard_points"] = m
        stats["root_mode"] = self.root_mode
        return stats

    def _tile_fast(self, buf, recv, m, global_points, failure, guarded, stages, t0):
        """FAST (TilingAlgorithmV3, the reference's default, executable/main.cpp:299-301) on this batch: no root step.  The start
        level comes from the distribution of the whole batch (TilingAlgorithms.cpp:1473-1535: the ranks' prefix histograms
        summed), the levels from there down and the skipped levels down to 0 are local, and the root (reconstruct_single_node,
        :1661-1715) samples what the level-0 nodes of ALL ranks hold -- one behind the other in rank order = octant order --
        on rank 0 with AlwaysAdhereToMinSpacing; every rank gets the flags of its candidates back.  self.result holds
        (recv_xyz, keys, perm, level, dup)."""
        ctx, dev, world = self.ctx, self.device, self.world
        on_gpu = dist.get_backend(self.group) == "nccl"
        wire_dev = dev if on_gpu else "cpu"
        wire = (lambda t: t) if on_gpu else (lambda t: t.cpu())
        self.root_mode = "reconstructed on rank 0"

        def vote():
            flag = torch.tensor([1 if failure else 0], dtype=torch.int64, device=wire_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if int(flag.item()):
                if failure:
                    raise failure[0]
                raise api.SwzError(api.ERR_PEER_FAILED, "another rank failed to tile its shard of this batch")

        if global_points < self.params.fast_concurrency:
            failure.append(api.SwzError(api.ERR_BAD_ARG, "FAST: a batch needs at least fast_concurrency points"))
        hist = guarded(lambda: ctx.shard_fast_begin_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params),
                       np.zeros(1 << 18, dtype=np.uint32))
        vote()
        h = torch.from_numpy(hist.astype(np.int64)).to(wire_dev)
        dist.all_reduce(h, group=self.group)
        start = api.fast_start_level_from_counts(h.cpu().numpy().astype(np.uint64), self.params.fast_concurrency)
        ncand = guarded(lambda: ctx.shard_fast_run(start), 0)
        ckeys = torch.empty(max(ncand, 1), dtype=torch.int64, device=dev)
        cxyz = torch.empty((max(ncand, 1), 3), dtype=torch.float64, device=dev)
        if ncand:
            guarded(lambda: ctx.shard_fast_root_candidates_device(ckeys.data_ptr(), cxyz.data_ptr()), None)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        stages.mark("levels_ms")
        vote()
        counts = torch.zeros(world, dtype=torch.int64, device=wire_dev)
        counts[self.rank] = ncand
        dist.all_reduce(counts, group=self.group)
        counts = [int(c) for c in counts.cpu()]
        ckeys, cxyz = ckeys[:ncand], cxyz[:ncand]
        flags = None
        taken_all = None
        if self.rank == 0:
            kparts, xparts = [ckeys], [cxyz]
            for r in range(1, world):
                kb = torch.empty(counts[r], dtype=torch.int64, device=wire_dev)
                xb = torch.empty((counts[r], 3), dtype=torch.float64, device=wire_dev)
                if counts[r]:
                    dist.recv(kb, src=r, group=self.group)
                    dist.recv(xb, src=r, group=self.group)
                kparts.append(kb.to(dev))
                xparts.append(xb.to(dev))
            allk, allx = torch.cat(kparts), torch.cat(xparts).contiguous()
            total = allk.shape[0]
            taken_all = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
            if total:
                # (the ranks own ascending octants: their candidates one behind the other are in key order)
                idx = torch.arange(total, dtype=torch.int32, device=dev)
                guarded(lambda: ctx.sample_points_device(self.params.sampler, self.params.max_points_per_node, allk.data_ptr(),
                                                         idx.data_ptr(), total, allx.data_ptr(), total, 0, -1, self.bmin, self.bmax,
                                                         self.params.spacing_at_root, api.ALWAYS_ADHERE_TO_MIN_SPACING,
                                                         taken_all.data_ptr()), 0)
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
        elif ncand:
            dist.send(wire(ckeys), dst=0, group=self.group)
            dist.send(wire(cxyz), dst=0, group=self.group)
        vote()  # (rank 0 has sampled the root, or failed to: the others learn which before they wait for their flags)
        if self.rank == 0:
            off = 0
            for r in range(world):
                part = taken_all[off:off + counts[r]]
                off += counts[r]
                if r == 0:
                    flags = part.contiguous()
                elif counts[r]:
                    dist.send(wire(part.contiguous()), dst=r, group=self.group)
        elif ncand:
            fb = torch.empty(ncand, dtype=torch.uint8, device=wire_dev)
            dist.recv(fb, src=0, group=self.group)
            flags = fb.to(dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if ncand:
            guarded(lambda: ctx.shard_fast_set_root_device(flags.data_ptr()), None)
        stages.mark("root_ms")
        okeys = torch.empty(m, dtype=torch.int64, device=dev)
        operm = torch.empty(m, dtype=torch.int32, device=dev)
        olevel = torch.empty(m, dtype=torch.int8, device=dev)
        odup = torch.empty(m, dtype=torch.int32, device=dev)
        zero = dict(num_nodes=0, points_visited=0, max_level=-1, fast_start_levels=-1, num_levels=0, min_distance_rounds=0)
        stats = guarded(lambda: ctx.shard_fast_finish_device(okeys.data_ptr(), operm.data_ptr(), olevel.data_ptr(), odup.data_ptr()), zero)
        self._stages = stages
        vote()
        self.result = (recv, okeys, operm, olevel, odup)
        self._keepalive = buf
        stats["shard_points"] = m
        stats["root_mode"] = self.root_mode
        return stats

    def stage_timings(self):
        """Milliseconds per stage of the last batch on this rank (synchronises the device)."""
        return self._stages.ms() if getattr(self, "_stages", None) else {}


class ShardedBatchTiler:
    """A data set that arrives in several batches, tiled by all ranks of a process group (BASELINE config 5): every
    rank runs one api.Tiler for the subtrees of its level-0 octants and its part of the root's file.  Per batch:
    encode, group by destination, ONE exchange step of the point rows and of every attribute column, then the root
    node -- decided from global counts, for MIN_DISTANCE rank by rank with the lower ranks' root files as ghosts --
    and, without communication, the levels below it."""

    def __init__(self, ctx, device, bmin, bmax, params, group=None, capacity_hint=0):
        self.ctx, self.device, self.bmin, self.bmax, self.params, self.group = ctx, device, bmin, bmax, params, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world not in (1, 2, 4, 8):
            raise ValueError("world size must be 1, 2, 4 or 8 (octants are dealt out in contiguous blocks)")
        if device.type == "cuda":
            ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.tiler = api.Tiler(ctx, bmin, bmax, params, capacity_hint)
        self._keep = None
        self.fast_start = -1  # FAST: the start level, known after the first batch
        # the MIN_DISTANCE root of every batch swept by all ranks at once (like ShardedTiler): on by default, after a probe
        self.joint_root = os.environ.get("SWZ_SHARD_JOINT_ROOT", "1") not in ("", "0")
        self._joint_usable = None
        self.root_mode = "local"

    def close(self):
        self.tiler.close()

    def _fold_stages(self):
        # the stage times of the batches of a data set add up (read by bench.py after the last one)
        self._pending_stages = getattr(self, "_pending_stages", [])
        self._pending_stages.append(self._batch_stages)

    def stage_timings(self):
        """Milliseconds per stage summed over the batches added so far on this rank (synchronises the device)."""
        tot = {}
        for st in getattr(self, "_pending_stages", []):
            for k, v in st.ms().items():
                tot[k] = round(tot.get(k, 0.0) + v, 3)
        return tot

    def _all_sum(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64,
                         device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def _bcast(self, tensor, src):
        if dist.get_backend(self.group) == "nccl" or tensor.device.type == "cpu":
            dist.broadcast(tensor, src, group=self.group)
            return tensor
        t = tensor.cpu()
        dist.broadcast(t, src, group=self.group)
        tensor.copy_(t)
        return tensor

    def add_batch(self, xyz, attrs=None):
        """xyz: [n, 3] float64 on this rank's GPU (any points of the batch), attrs: dict name -> tensor with n rows.
        Returns the tile stats of this rank's shard for the batch."""
        ctx, dev, world = self.ctx, self.device, self.world
        attrs = attrs or {}
        n = xyz.shape[0]
        stages = self._batch_stages = _Stages(dev)
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.morton_encode_device(xyz.data_ptr(), n, self.bmin, self.bmax, keys.data_ptr())
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        octant_counts = ctx.partition_by_octant_device(keys.data_ptr(), n, perm.data_ptr())
        del keys
        send_counts = rank_send_counts(octant_counts, world)
        order = perm.long()
        del perm
        global_new = self._all_sum(n)
        send = xyz.index_select(0, order)
        stages.mark("encode_partition_group_ms")
        recv, recv_counts = exchange_rows(send, send_counts, self.group)
        del send
        m = recv.shape[0]
        cols = {}
        for name, t in attrs.items():  # the same exchange, column by column
            rows = t.index_select(0, order).reshape(n, -1)
            got, rc = exchange_rows(rows, send_counts, self.group)
            assert rc == recv_counts
            cols[name] = got.contiguous()
        del order
        stages.mark("exchange_ms")
        if self.params.strategy == api.FAST:
            st = self._add_batch_fast(recv, m, cols, global_new)
            stages.mark("levels_ms")
            self._fold_stages()
            return st
        root_stored = self._all_sum(self.tiler.level_count(-1))
        sample = root_stored > 0 or global_new + root_stored > self.params.max_points_per_node
        sequential_root = self.params.sampler == api.MIN_DISTANCE and sample and global_new > 0
        ptrs = {name: t.data_ptr() for name, t in cols.items()}
        failure = []

        def guarded(fn, default):
            if failure:
                return default
            try:
                return fn()
            except api.SwzError as e:
                failure.append(e)
                return default

        gather = lambda mine: _all_gather_bytes(mine, self.group, dev, world)
        possible = (sequential_root and world > 1 and self.joint_root and dev.type == "cuda" and
                    ctx.shard_joint_root_possible(self.bmin, self.bmax, self.params))
        if possible and self._joint_usable is None:
            self._joint_usable = _joint_root_usable(ctx, self.rank, world, gather)
        joint = bool(possible and self._joint_usable)
        why_chain = "no IPC mapping"
        if joint:
            # The probe exported a small hipMalloc buffer; the real thing exports the tiler's position pool, and a pool that
            # has spilled to page-locked host memory (SWZ_TILER_SPILL, a device budget, out of memory) has no IPC handle
            # (ADVICE r4): reserve this batch's room first, then every rank says where its pool lives -- one vote per batch.
            exportable = 1
            try:
                self.tiler.reserve(int(self.tiler.info()["num_points"]) + m)
                exportable = int(self.tiler.pool_residency()[1] == 0)
            except api.SwzError:
                exportable = 0  # (the batch fails on its own below; the others must not wait for this rank's export)
            if self._all_sum(exportable) != world:
                joint, why_chain = False, "a pool lives in host memory"
        self.root_mode = "local" if not sequential_root else ("joint" if joint else ("chain (%s)" % why_chain if possible else "chain"))
        if not sequential_root:
            guarded(lambda: self.tiler.shard_begin_device(recv.data_ptr(), m, ptrs, global_new, root_stored), 0)
        elif joint:
            # the batch's points merged with this rank's part of the root's file, swept together with the other ranks' parts
            _joint_root_step(ctx, self.rank, world, self.group, gather,
                             lambda: self.tiler.shard_begin_device(recv.data_ptr(), m, ptrs, global_new, root_stored), failure)
        else:
            ghosts = []
            for r in range(world):
                cnt = torch.zeros(1, dtype=torch.int64, device=dev)
                mine = None
                if r == self.rank:
                    gbuf = torch.cat(ghosts) if ghosts else None
                    g = gbuf.shape[0] if gbuf is not None else 0
                    have = guarded(lambda: self.tiler.shard_begin_device(recv.data_ptr(), m, ptrs, global_new, root_stored,
                                                                         gbuf.data_ptr() if g else None, g), 0)
                    mine = torch.empty((have, 3), dtype=torch.float64, device=dev)
                    if have:
                        guarded(lambda: self.tiler.level_positions_device(-1, mine.data_ptr()), None)
                    cnt[0] = have
                if r == world - 1:
                    break
                self._bcast(cnt, r)
                b = mine if r == self.rank else torch.empty((int(cnt.item()), 3), dtype=torch.float64, device=dev)
                if b.shape[0]:
                    self._bcast(b, r)
                if self.rank > r:
                    ghosts.append(b)
        stages.mark("root_ms")
        zero = dict(num_nodes=0, points_visited=0, max_level=-1, fast_start_levels=-1, num_levels=0, min_distance_rounds=0)
        stats = guarded(lambda: self.tiler.shard_finish(), zero)
        stages.mark("levels_ms")
        self._fold_stages()
        if self._all_sum(1 if failure else 0):
            # the ranks that did not fail have committed this batch and the failing one has not: no rank may go on
            try:
                self.tiler.poison("a rank of the sharded run failed to tile this batch")
            except api.SwzError:
                pass
            if failure:
                raise failure[0]
            raise api.SwzError(api.ERR_PEER_FAILED, "another rank failed to tile its shard of this batch")
        self._keep = (recv, cols)
        stats["shard_points"] = m
        stats["root_mode"] = self.root_mode
        return stats

    # -- FAST (TilingAlgorithmV3, the reference's default): no root step per batch.  The start level comes from the FIRST
    # batch's distribution over all shards (TilingAlgorithms.cpp:1473-1535) and is kept; the skipped levels are rebuilt at
    # the end of the data set (finalize below).
    def _vote(self, failure):
        if self._all_sum(1 if failure else 0):
            try:
                self.tiler.poison("a rank of the sharded run failed to tile this batch")
            except api.SwzError:
                pass
            if failure:
                raise failure[0]
            raise api.SwzError(api.ERR_PEER_FAILED, "another rank failed to tile its shard of this batch")

    def _add_batch_fast(self, recv, m, cols, global_new):
        failure = []
        ptrs = {name: t.data_ptr() for name, t in cols.items()}
        try:
            if global_new < self.params.fast_concurrency:
                raise api.SwzError(api.ERR_BAD_ARG, "FAST: a batch needs at least fast_concurrency points")
            self.tiler.shard_begin_device(recv.data_ptr(), m, ptrs, global_new, 0)
        except api.SwzError as e:
            failure.append(e)
        if self.fast_start < 0:
            hist = np.zeros(1 << 18, dtype=np.int64)
            if not failure:
                try:
                    hist = self.tiler.shard_fast_histogram().astype(np.int64)
                except api.SwzError as e:
                    failure.append(e)
            t = torch.from_numpy(hist)
            if dist.get_backend(self.group) == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, group=self.group)
            self._vote(failure)
            self.fast_start = api.fast_start_level_from_counts(t.cpu().numpy().astype(np.uint64), self.params.fast_concurrency)
        stats = dict(num_nodes=0, points_visited=0, max_level=-1, fast_start_levels=-1, num_levels=0, min_distance_rounds=0)
        if not failure:
            try:
                self.tiler.shard_set_start_level(self.fast_start)
                stats = self.tiler.shard_finish()
            except api.SwzError as e:
                failure.append(e)
        self._vote(failure)
        self._keep = (recv, cols)
        stats["shard_points"] = m
        stats["root_mode"] = self.root_mode
        return stats

    def finalize(self):
        """Ends the data set.  ACCURATE: nothing is left to do beyond the tiler's own finalize.  FAST: every rank rebuilds the
        skipped levels of its octants down to level 0; the root (reconstruct_single_node, TilingAlgorithms.cpp:1661-1715)
        samples what its eight children hold, and they lie on different ranks: their level-0 files come together on rank 0
        in rank order (= octant order, the order the reference appends them in), are indexed against the root bounds and
        sampled with AlwaysAdhereToMinSpacing there; every rank keeps the part of the root's file that comes from its points."""
        if self.params.strategy != api.FAST or self.fast_start <= 0:
            return self.tiler.finalize()
        dev, world = self.device, self.world
        on_gpu = dist.get_backend(self.group) == "nccl"
        # Every collective phase below is preceded by a vote (like add_batch): a rank whose local step raised must not leave
        # the others waiting in all_reduce / recv / send for a peer that is gone.
        failure = []
        stats, cnt = None, 0
        try:
            stats = self.tiler.shard_fast_finalize_local()
            cnt = self.tiler.level_count(0)
        except api.SwzError as e:
            failure.append(e)
        self._vote(failure)
        counts = torch.zeros(world, dtype=torch.int64, device=dev if on_gpu else "cpu")
        counts[self.rank] = cnt
        dist.all_reduce(counts, group=self.group)
        counts = [int(c) for c in counts.cpu()]
        mine = torch.empty((max(cnt, 1), 3), dtype=torch.float64, device=dev)
        try:
            if cnt:
                self.tiler.level_positions_device(0, mine.data_ptr())
                torch.cuda.synchronize(dev) if dev.type == "cuda" else None
        except api.SwzError as e:
            failure.append(e)
        self._vote(failure)
        mine = mine[:cnt]
        wire = (lambda t: t) if on_gpu else (lambda t: t.cpu())
        flags = None
        if self.rank == 0:
            parts = [mine.cpu()]
            for r in range(1, world):
                buf = torch.empty((counts[r], 3), dtype=torch.float64, device=dev if on_gpu else "cpu")
                if counts[r]:
                    dist.recv(buf, src=r, group=self.group)
                parts.append(buf.cpu())
            allp = torch.cat(parts).numpy()
            taken = np.zeros(allp.shape[0], dtype=np.uint8)
            try:
                if allp.shape[0]:
                    keys, clamped = self.ctx.morton_encode(allp, self.bmin, self.bmax)
                    perm, skeys = self.ctx.sort_by_key(keys)
                    t_sorted = self.ctx.sample_points(self.params.sampler, self.params.max_points_per_node, skeys, perm, clamped, 0, -1,
                                                      self.bmin, self.bmax, self.params.spacing_at_root, api.ALWAYS_ADHERE_TO_MIN_SPACING)
                    taken[perm] = t_sorted
            except api.SwzError as e:
                failure.append(e)
        else:
            if cnt:
                dist.send(wire(mine), dst=0, group=self.group)
        # (rank 0 has sampled the root, or failed to: the others learn which before they wait for their flags)
        self._vote(failure)
        if self.rank == 0:
            off = 0
            for r in range(world):
                part = torch.from_numpy(taken[off:off + counts[r]].copy())
                off += counts[r]
                if r == 0:
                    flags = part.to(dev)
                elif counts[r]:
                    dist.send(wire(part.to(dev)), dst=r, group=self.group)
        elif cnt:
            buf = torch.empty(cnt, dtype=torch.uint8, device=dev if on_gpu else "cpu")
            dist.recv(buf, src=0, group=self.group)
            flags = buf.to(dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        try:
            self.tiler.shard_fast_set_root(flags.data_ptr() if (flags is not None and cnt) else None)
        except api.SwzError as e:
            failure.append(e)
        self._vote(failure)
        return stats

