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


def exchange_rows(rows, send_counts, group=None, headroom=0):
    """all_to_all_single of row blocks.  rows: [n, k] tensor already grouped by destination rank;
    send_counts[r] rows go to rank r.  Returns (buffer, recv_counts): the received rows are buffer[headroom:]
    (the first `headroom` rows are left free so that a few rows can later be put right in front)."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = rows.device
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    sc = torch.tensor(send_counts, dtype=torch.int64, device=comm_dev)
    rc = torch.empty(world, dtype=torch.int64, device=comm_dev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(v) for v in rc.tolist()]
    m = sum(recv_counts)
    buf = torch.empty((headroom + m,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=dev)
    if comm_dev == dev:
        dist.all_to_all_single(buf[headroom:], rows.contiguous(), recv_counts, [int(v) for v in send_counts],
                               group=group)
    else:
        recv = torch.empty((m,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=comm_dev)
        dist.all_to_all_single(recv, rows.to(comm_dev).contiguous(), recv_counts, [int(v) for v in send_counts],
                               group=group)
        buf[headroom:].copy_(recv)
    return buf, recv_counts


class ShardedTiler:
    """Tiles one batch whose points are spread over the ranks of a process group."""

    def __init__(self, ctx, device, bmin, bmax, params, group=None):
        self.ctx, self.device, self.bmin, self.bmax, self.params, self.group = ctx, device, bmin, bmax, params, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world not in (1, 2, 4, 8):
            raise ValueError("world size must be 1, 2, 4 or 8 (octants are dealt out in contiguous blocks)")
        self.result = None

    def _bcast(self, tensor, src):
        backend = dist.get_backend(self.group)
        if backend == "nccl" or tensor.device.type == "cpu":
            dist.broadcast(tensor, src, group=self.group)
            return tensor
        t = tensor.cpu()
        dist.broadcast(t, src, group=self.group)
        tensor.copy_(t)
        return tensor

    def tile(self, xyz):
        """xyz: [n, 3] float64 tensor on this rank's GPU (any points of the batch).  Returns the tile stats of this
        rank's shard; self.result holds (recv_xyz, keys, perm, level) for the points this rank owns."""
        ctx, dev, world = self.ctx, self.device, self.world
        n = xyz.shape[0]
        if dev.type == "cuda":
            # keys + perm + send buffer + receive buffer are about 3 x the positions; when a previous batch left a
            # workspace that does not leave room for them, give it back first (it is re-grown on demand)
            free, _ = torch.cuda.mem_get_info(dev)
            if free < int(3.2 * n * 24) and ctx.workspace_bytes() > 0:
                ctx.release_workspace()
        # 1. encode locally, group by destination
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.morton_encode_device(xyz.data_ptr(), n, self.bmin, self.bmax, keys.data_ptr())
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        octant_counts = ctx.partition_by_octant_device(keys.data_ptr(), n, perm.data_ptr())
        del keys
        send_counts = rank_send_counts(octant_counts, world)
        send = xyz.index_select(0, perm.long())
        del perm
        # 2. the one exchange step
        total = torch.tensor([n], dtype=torch.int64, device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(total, group=self.group)
        global_points = int(total.item())
        sequential_root = self.params.sampler == api.MIN_DISTANCE and global_points > self.params.max_points_per_node
        headroom = GHOST_HEADROOM if (sequential_root and self.rank > 0) else 0
        buf, _ = exchange_rows(send, send_counts, self.group, headroom)
        del send
        if dev.type == "cuda":
            torch.cuda.empty_cache()  # hand the freed send buffers back: the context allocates with hipMalloc
        recv = buf[headroom:]
        m = recv.shape[0]
        # 3. root node
        if not sequential_root:
            ctx.shard_begin_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params, global_points)
        else:
            if m > 0:
                # everything that does not depend on the ghosts happens on all ranks at once; only the root
                # node itself is left in the chain below
                ctx.shard_presort_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params, GHOST_HEADROOM)
            ghosts = []
            for r in range(world):
                cnt = torch.zeros(1, dtype=torch.int64, device=dev)
                mine = None
                if r == self.rank:
                    g = sum(t.shape[0] for t in ghosts)
                    if g == 0:
                        gptr = None
                    elif g <= headroom:  # ghosts go right in front of the received points: no staging copy
                        torch.cat(ghosts, out=buf[headroom - g:headroom])
                        gptr = buf[headroom - g:].data_ptr()
                    else:
                        gbuf = torch.cat(ghosts)
                        gptr = gbuf.data_ptr()
                    taken = ctx.shard_begin_device(recv.data_ptr(), m, self.bmin, self.bmax, self.params, global_points,
                                                   gptr, g)
                    mine = torch.empty((taken, 3), dtype=torch.float64, device=dev)
                    if taken:
                        ctx.shard_root_taken_device(mine.data_ptr())
                    cnt[0] = taken
                if r == world - 1:
                    break  # nobody owns higher octants
                self._bcast(cnt, r)
                b = mine if r == self.rank else torch.empty((int(cnt.item()), 3), dtype=torch.float64, device=dev)
                if b.shape[0]:
                    self._bcast(b, r)
                if self.rank > r:
                    ghosts.append(b)
        # 4. everything below the root is local
        okeys = torch.empty(m, dtype=torch.int64, device=dev)
        operm = torch.empty(m, dtype=torch.int32, device=dev)
        olevel = torch.empty(m, dtype=torch.int8, device=dev)
        stats = ctx.shard_finish_device(okeys.data_ptr(), operm.data_ptr(), olevel.data_ptr())
        self.result = (recv, okeys, operm, olevel)
        self._keepalive = buf  # the context reads the points until shard_finish returned
        stats["shard_points"] = m
        return stats
