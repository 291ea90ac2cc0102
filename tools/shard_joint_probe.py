#!/usr/bin/env python3
"""Times the sharded single-batch tiler with W processes sharing GPU 0 (collectives over gloo), the MIN_DISTANCE root taken
in turns and swept by all ranks at once (SWZ_SHARD_JOINT_ROOT=1: IPC mappings).  usage: shard_joint_probe.py WORLD POINTS_PER_RANK"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, n, joint, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SWZ_SHARD_JOINT_ROOT="1" if joint else "0")
    import schwarzwald_amd as swz
    from schwarzwald_amd import sharded
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
    ctx.generate_uniform_device(0x5C4A72A1D + 3, rank * n, n, xyz.data_ptr())
    p = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=20000, spacing_at_root=swz.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250))
    t = sharded.ShardedTiler(ctx, dev, [0, 0, 0], [1, 1, 1], p)
    times = []
    for rep in range(3):
        dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        t.tile(xyz)
        torch.cuda.synchronize(); dist.barrier(); times.append((time.perf_counter() - t0) * 1e3)
    q.put((rank, times, t.used_joint_root))
    ctx.close(); dist.destroy_process_group()


if __name__ == "__main__":
    world, n = int(sys.argv[1]), int(sys.argv[2])
    for joint in (False, True):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        c = mp.get_context("spawn"); q = c.Queue()
        ps = [c.Process(target=worker, args=(r, world, port, n, joint, q)) for r in range(world)]
        [p.start() for p in ps]
        got = [q.get(timeout=600) for _ in range(world)]
        [p.join(60) for p in ps]
        r0 = [g for g in got if g[0] == 0][0]
        print("%d ranks x %d points on one GPU, root %s: %s ms per batch (rank 0, three batches)" % (
            world, n, "swept by all ranks at once" if r0[2] else "in turns", ", ".join("%.0f" % x for x in r0[1])), flush=True)
