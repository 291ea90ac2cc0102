import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
for n in (10_000_000, 80_000_000, 100_000_000, 200_000_000):
    rows = torch.arange(n * 3, dtype=torch.float64, device=dev).reshape(n, 3)
    out = torch.zeros_like(rows)
    dist.all_to_all_single(out, rows, [n], [n])
    torch.cuda.synchronize()
    bad = (out != rows).any(dim=1)
    nb = int(bad.sum())
    first = int(torch.nonzero(bad)[0]) if nb else -1
    print("n=%d rows (%.2f GB): mismatching rows %d, first %d" % (n, n * 24 / 1e9, nb, first), flush=True)
    del rows, out, bad
dist.destroy_process_group()
