#!/usr/bin/env python3
"""bench.py -- Mpoints/s of the tiler hot path (Morton encode + radix sort + octree partition + LOD
sampling) on synthetic uniform points, per BASELINE.json.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic points that already sit in HBM when
the timed region starts.  N=1 runs BASELINE.json's headline configuration (1 B points, MIN_DISTANCE,
spacing = diagonal/250, ACCURATE strategy) on one MI355X.  For N>1 (launched by torch.distributed.run,
one rank per GPU) every rank owns the same number of points (weak scaling); points are exchanged once
after the encode by their top Morton bits (RCCL all-to-all) and every rank then tiles its own octants.

Rank 0 prints ONE JSON line with the metric, the roofline of the dominant kernel (HIP-event timing taken
inside the timed region) and, at N=1, a CPU baseline (the oracle, timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEED = 0x5C4A72A1D


def check_shard_stamps(rows, step_ms):
    """Per-shard stage stamps are milliseconds inside ONE step: a value that is negative, not finite or larger than the
    step it belongs to is a bookkeeping error (a stamp taken against an unset clock once printed 1.85e8 ms) -- the line
    then carries the reason instead of the numbers.  Returns None when the rows are plausible."""
    import math
    for r in rows:
        for k, v in r.items():
            if k == "shard" or v is None:
                continue
            if not math.isfinite(v) or v < 0.0 or (step_ms > 0.0 and v > step_ms * 1.05 + 1.0):
                return "shard %s: %s = %r ms does not fit into a step of %.3f ms" % (r.get("shard"), k, v, step_ms)
    return None


def shard_stamp_error(report, step_ms):
    """the same check for the per-rank stage times of the torch driver (numeric *_ms fields of every rank's row)"""
    rows = [{"shard": r.get("rank", i), **{k: v for k, v in r.items() if k.endswith("_ms") and isinstance(v, (int, float))}}
            for i, r in enumerate(report or [])]
    return check_shard_stamps(rows, step_ms)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=1_000_000_000, help="points per GPU")
    ap.add_argument("--sampler", default="MIN_DISTANCE", choices=["RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED"])
    ap.add_argument("--diagonal-fraction", type=float, default=250.0)
    ap.add_argument("--max-points-per-node", type=int, default=20000)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="points of the CPU-baseline sample (0 = skip); "
                    "100 M is what SURVEY.md section 8(d) asks for (about a minute on the GPU box's host cores)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all host cores available)")
    ap.add_argument("--strategy", default="ACCURATE", choices=["ACCURATE", "FAST"],
                    help="tiling strategy (TilingAlgorithmV1 / V3); the headline is ACCURATE, the canonical top-down semantics")
    ap.add_argument("--fast-concurrency", type=int, default=8, help="FAST: the thread count its start level depends on")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--bounds-scale", type=float, default=1.0, help="experiment: tile the unit-cube points inside a "
                    "cube this many times larger (2 = the points fill one octant of the root, which is what one rank "
                    "of an 8-GPU run sees at the root level)")
    ap.add_argument("--payload", default="", help="comma separated attribute columns (e.g. rgb,intensity): after the "
                    "timed region also build the node lists and gather the node payload on the device; reported "
                    "separately under \"payload\", never part of `value`")
    ap.add_argument("--md-mode", default="both", choices=["exact", "property", "both"], help="MIN_DISTANCE: exact = the "
                    "reference's Morton-order greedy set, bit-identical (the headline `value`); property = "
                    "SWZ_FLAG_MIN_DISTANCE_PROPERTY (same spacing + maximality guarantees, different priority order); "
                    "both = `value` is exact and the property-mode timing of the same workload is reported beside it "
                    "under \"min_distance_property\"")
    ap.add_argument("--batches", type=int, default=1, help="tile the points in this many batches through the multi-batch "
                    "tiler (swz_tiler_*: cached-point re-read + merge like the reference with internal_cache_size < N); "
                    "a step is then the whole data set, batch after batch, into a fresh tiler")
    ap.add_argument("--batch-order", default="uniform", choices=["uniform", "tiles"], help="with --batches k: uniform = every "
                    "batch is cut out of the whole cloud (each batch reaches every node: the worst case for a multi-batch "
                    "tiler); tiles = batch i holds the points of the i-th tile of a regular x-y grid over the bounds, the "
                    "way a data set arrives as LAS tiles (spatially coherent: a batch reaches a small part of the tree)")
    ap.add_argument("--staged", action="store_true", help="with --batches: additionally time the batches coming from "
                    "PINNED HOST memory through swz_tiler_stage_batch / swz_tiler_tile_staged (hipMemcpyAsync of batch "
                    "k+1 under the kernels of batch k; attribute columns of --payload travel along); reported under "
                    "\"staged\" (PCIe-inclusive, never `value`)")
    ap.add_argument("--driver", default="torch", choices=["torch", "group"], help="with --gpus N: torch = one process per "
                    "GPU (torch.distributed over RCCL, schwarzwald_amd/sharded.py; what the round's scaling run uses); group = "
                    "ONE C++ process for all GPUs (swz_group_*, tools/group_bench.cpp: the shape of the reference's own host), "
                    "started as a child process; with --batches k through swz_group_add_batch / swz_group_finalize")
    ap.add_argument("--group-transport", default="rccl", choices=["rccl", "peer"], help="--driver group: exchange by grouped "
                    "ncclSend / ncclRecv or by hipMemcpyPeerAsync (peer copies also work with all shards on ONE device)")
    ap.add_argument("--group-devices", type=int, default=0, help="--driver group: devices the shards are dealt to (0 = as "
                    "many as there are shards; 1 = all shards share device 0, the only way to run it on a one-GPU box)")
    ap.add_argument("--also", default="GRID_CENTER,GRID_CENTER@100000000,JITTERED,MIN_DISTANCE_FAST", help="N = 1, one batch, headline workload only: short extra legs "
                    "after the timed region, reported under \"also\" and never part of `value` -- a sampler name runs the same points "
                    "through that sampler (GRID_CENTER: BASELINE configs[1]'s sampler at the headline size), <sampler>_FAST through the "
                    "FAST strategy (the reference's default, executable/main.cpp:299-301), <leg>@POINTS on the first POINTS points of the cloud "
                    "(GRID_CENTER@100000000 = BASELINE configs[1]); a leg that fails reports its error and costs nothing else; \"\" = none")
    ap.add_argument("--config", type=int, default=0, choices=[0, 4, 5], help="BASELINE.json's multi-GPU configurations as one "
                    "command each: 4 = 1 B points IN TOTAL, JITTERED, sharded over --gpus ranks by the top Morton bits (strong "
                    "scaling); 5 = 4 B points in total with RGB + intensity, MIN_DISTANCE, every rank's share staged from pinned "
                    "host memory in batches of about 50 M points (strong scaling).  --total-points scales either down for a dry "
                    "run; --driver group runs the same shape from ONE C++ process")
    ap.add_argument("--total-points", type=int, default=0, help="points of the whole job, split evenly over the ranks (strong "
                    "scaling: `scaling` says so); overrides --points")
    ap.add_argument("--one-device", action="store_true", help="dry run of an N-rank configuration on ONE GPU: all ranks share "
                    "cuda:0 and talk over gloo (RCCL does not put two ranks on one device); the line is labelled a dry run")
    args = ap.parse_args()
    if args.config == 4:
        args.sampler, args.strategy, args.batches = "JITTERED", "ACCURATE", 1
        args.total_points = args.total_points or 1_000_000_000
    elif args.config == 5:
        args.sampler, args.staged = "MIN_DISTANCE", True
        args.payload = args.payload or "rgb,intensity"
        args.total_points = args.total_points or 4_000_000_000
        if args.md_mode == "both":
            args.md_mode = "exact"
    if args.total_points:
        args.points = args.total_points // max(args.gpus, 1)
    if args.config == 5 and args.batches <= 1:
        args.batches = max(2, (args.points + 49_999_999) // 50_000_000)  # (2 at least: the path under test is the batch tiler)
    return args


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching HIP (the ranks are started as children before anything in this
    process initialises a GPU): HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set, else the KFD topology's GPU nodes.
    None when neither tells."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        n = 0
        base = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(base):
            try:
                props = open(os.path.join(base, node, "properties")).read()
            except OSError:
                continue
            for line in props.splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        return n or None  # (no GPU node listed: rather let the ranks find out than refuse wrongly)
    except OSError:
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks as CHILD processes (one per GPU, RCCL) before this
    process touches a GPU, relay their output, exit with their code."""
    import socket
    import subprocess
    have = visible_gpus()
    if have is not None and have < args.gpus and not args.one_device:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (--one-device: a dry run with all ranks on one)\n" % (args.gpus, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def run_group_driver(args):
    """`--driver group`: the sharded batch driven from ONE C++ process (swz_group_*).  Builds tools/group_bench.cpp against
    the library, runs it as a child (this process never touches a GPU) and reports its wall times in the usual line."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    shards = args.gpus
    if shards not in (1, 2, 4, 8):
        sys.stderr.write("bench.py --driver group: 1, 2, 4 or 8 shards\n")
        return 2
    devices = args.group_devices or shards
    have = visible_gpus()
    if have is not None and have < devices:
        sys.stderr.write("bench.py: --driver group wants %d device(s) but only %d GPU(s) visible (--group-devices 1 puts all "
                         "shards on one)\n" % (devices, have))
        return 2
    transport = 1 if (args.group_transport == "rccl" and devices == shards) else 0  # RCCL needs one device per shard
    lib_dir = os.path.join(root, "schwarzwald_amd", "lib")
    exe = os.path.join(tempfile.mkdtemp(prefix="swz_group_bench_"), "group_bench")
    subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(root, "tools", "group_bench.cpp"), "-o", exe, "-L" + lib_dir,
                    "-lswz_gpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    sampler = ["RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED"].index(args.sampler)
    cmd = [exe, str(shards), str(args.points), str(args.warmup + args.steps), str(devices), str(transport), str(args.batches),
           str(sampler), "1" if args.strategy == "FAST" else "0", str(args.warmup), "1" if (args.staged and args.batches > 1) else "0"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    sys.stderr.write(r.stderr)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        return r.returncode
    ms = [float(line.split(" points: ")[1].split(" ms")[0]) for line in r.stdout.splitlines() if line.startswith("rep ")]
    timed = ms[args.warmup:]
    if len(timed) != args.steps:
        sys.stderr.write("bench.py --driver group: expected %d timed steps, the driver reported %d\n" % (args.steps, len(timed)))
        return 1
    total_ms = sum(timed)
    # shard 0's kernel classes over the timed steps (HIP events on its stream) -> the roofline object; every shard's
    # critical-path stamps of the last step
    prof = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("class "):
            _, name, launches, ms_, nbytes = ln.split()
            prof[name] = {"launches": int(launches), "total_ms": float(ms_), "algorithmic_bytes": int(nbytes)}
    roofline = None
    if prof:
        name, k = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        avg_ms = k["total_ms"] / max(k["launches"], 1)
        achieved = k["algorithmic_bytes"] / max(k["launches"], 1) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": name, "shard": 0, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "launches": k["launches"], "avg_launch_ms": round(avg_ms, 4)}
    last_rep = args.warmup + args.steps - 1
    shard_rows = []
    for ln in r.stdout.splitlines():
        f = ln.split()
        if ln.startswith("shard ") and int(f[1]) == last_rep:
            shard_rows.append({"shard": int(f[2]), "exchange_done_ms": float(f[3]), "root_begun_ms": float(f[4]),
                               "root_done_ms": float(f[5]), "levels_done_ms": float(f[6])})
        if ln.startswith("shardsum ") and int(f[1]) == last_rep:  # batches: per stage, summed over the batches of the last data set
            shard_rows.append({"shard": int(f[2]), "exchange_ms": float(f[3]), "root_ms": float(f[4]), "levels_ms": float(f[5])})
    stamp_error = check_shard_stamps(shard_rows, timed[-1] if timed else 0.0)
    line = {
        "metric": "Mpoints/s end-to-end tile (Morton+sort+sample)", "value": round(shards * args.points * args.steps / total_ms / 1e3, 3),
        "unit": "Mpoints/s", "n_gpus": shards, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(total_ms / args.steps, 3),
        "higher_is_better": True, "scaling": "strong" if args.total_points else "weak", "vs_baseline": None, "dtype": "u64 keys / f64 positions", "data": "synthetic",
        "config": {"workload": "%s%d uniform points per shard in the unit cube, %s sampling, spacing = diagonal/250, "
                               "max_points_per_node=20000, %s strategy, %d batch(es)%s" % (
                                   ("BASELINE config %d%s: %d points in total, " % (args.config, " (scaled down)" if args.total_points != {
                                       4: 1_000_000_000, 5: 4_000_000_000}[args.config] else "", args.total_points)) if args.config else "",
                                   args.points, args.sampler, args.strategy, args.batches,
                                   " staged from pinned host memory with RGB + intensity (29 B/pt; the wall time includes the host link)"
                                   if (args.staged and args.batches > 1) else ""),
                   "baseline_config": args.config or None, "total_points": args.total_points or None,
                   "staged_from_pinned_host": bool(args.staged and args.batches > 1) or None,
                   "points_per_gpu": args.points, "sampler": args.sampler, "strategy": args.strategy, "batches": args.batches,
                   "parallelism": "%d shards on %d device(s), ONE C++ process (swz_group_%s), exchange by %s" % (
                       shards, devices, "add_batch" if args.batches > 1 else "tile", "RCCL send/recv" if transport else "peer copies")},
        "driver": "group", "timed_region": "wall clock of the whole call(s) inside the child process: encode, partition, exchange, "
                                           "sort, root, levels (inputs resident on the shards' devices)",
        "steps_ms": [round(x, 3) for x in timed], "roofline": roofline,
        "kernels_ms_per_step_shard0": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())},
        "shards": None if stamp_error else shard_rows, "cpu_baseline": None, "cpu_baseline_reference": committed_cpu_baseline(),
        "ranks_in_process_group": 1, "root_mode": os.environ.get("SWZ_GROUP_JOINT_ROOT", "1") not in ("", "0") and "joint" or "chain",
        "exchange_ms": None if stamp_error else max((r.get("exchange_done_ms", r.get("exchange_ms", 0.0)) for r in shard_rows), default=None),
        "stamp_error": stamp_error,
        "driver_output": r.stdout.splitlines()[-min(len(ms), 3):],
    }
    print(json.dumps(line))
    return 0


def implemented_sort_frac(prof, args, visit_factor, points, seconds_per_step):
    """hbm_frac_end_to_end with the sort priced at what the IMPLEMENTED sort has to move instead of SURVEY's 8 + 8 x 24 =
    200 B/pt (an eight-pass LSD sort): one histogram pass over the keys (8) + 24 B per scatter pass actually launched
    (four over the top digits at 1 B uniform points) + 24 for the pass that orders the runs of equal top bits, if it ran.
    None without the kernel profile."""
    k = prof.get("radix_scatter") if prof else None
    if not k or not seconds_per_step:
        return None
    # (full passes over the data: the bracket also holds the eight tiny passes that sort the sample which picks the number
    # of top digits, so the launches are not the passes -- the bytes are)
    passes = k["algorithmic_bytes"] / float(args.steps) / (24.0 * points)
    sort_bytes = 8.0 + 24.0 * passes + (24.0 if "radix_runs" in prof else 0.0)
    per_level = 33.0 if args.sampler == "RANDOM_GRID" else 57.0
    alg = 32.0 + sort_bytes + per_level * visit_factor
    return {"frac": round(alg * points / seconds_per_step / (HBM_PEAK_GBS * 1e9), 5), "bytes_per_point": round(alg, 1),
            "sort_bytes_per_point": round(sort_bytes, 1), "scatter_passes": round(passes, 2)}


def algorithmic_bytes_per_point(sampler, visit_factor):
    """encode 32 + sort 200 + per visited level 33 (RANDOM_GRID) or 57."""
    per_level = 33.0 if sampler == "RANDOM_GRID" else 57.0
    return 32.0 + 200.0 + per_level * visit_factor


def library_source_sha16():
    """What the committed traffic numbers are stamped with: a hash of the kernel sources the library is built from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "schwarzwald_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "schwarzwald_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "schwarzwald_amd", "csrc", "*.inc"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_rounds():
    """profiles/rNN, newest first"""
    try:
        names = [d for d in os.listdir(os.path.join(ROOT, "profiles")) if len(d) == 3 and d[0] == "r" and d[1:].isdigit()]
    except OSError:
        return []
    return sorted(names, reverse=True)


def measured_traffic(kernel_class, n, sampler, table="bytes_per_launch"):
    """HBM bytes per launch of the dominant kernel class from the committed rocprofv3 PMC passes
    (profiles/rNN/traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this very
    command, corrected as MI355X_MICROARCH.md prescribes) -- only when that profile was taken on the kernel sources this
    run uses (traffic.json carries their hash).  Returns (bytes or None, where the number comes from / why there is none)."""
    sha = library_source_sha16()
    for rnd in profile_rounds():  # the newest committed profile of this configuration
        path = os.path.join(ROOT, "profiles", rnd, "traffic.json")
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get("points") != n or t.get("sampler") != sampler:
            return None, "profiles/%s/traffic.json is for another workload" % rnd
        if t.get("source_sha16") != sha:
            return None, "stale: profiles/%s/traffic.json was measured on kernel sources %s, this run uses %s" % (
                rnd, t.get("source_sha16", "(unstamped)"), sha)
        return t.get(table, {}).get(kernel_class), "profiles/%s/traffic.json (rocprofv3 --pmc, kernel sources %s)" % (rnd, sha)
    return None, "no committed profile"


def committed_cpu_baseline():
    """N > 1 lines carry no CPU timing of their own (rank 0 at N = 1 measures it): the newest committed N = 1 line's."""
    for rnd in profile_rounds():
        path = os.path.join(ROOT, "profiles", rnd, "bench_1B_min_distance.json")
        try:
            cb = json.load(open(path)).get("cpu_baseline")
        except Exception:
            continue
        if cb:
            return dict(cb, source="profiles/%s/bench_1B_min_distance.json" % rnd)
    return None


def cpu_baseline(args, spacing):
    """The oracle (CPU restatement of the reference path, single thread) on a bounded sample of the same
    workload: same generator, bounds, spacing and parameters, fewer points."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    n = args.cpu_sample
    threads = args.cpu_threads or len(os.sched_getaffinity(0))
    xyz = O.generate_uniform(SEED + 3, n)
    t0, c0 = time.perf_counter(), time.process_time()
    r = O.tile(xyz, [0, 0, 0], [args.bounds_scale] * 3, getattr(O, args.sampler), args.max_points_per_node, spacing,
               strategy=getattr(O, args.strategy), fast_concurrency=args.fast_concurrency, threads=threads)
    dt, cpu = time.perf_counter() - t0, time.process_time() - c0
    assert r["status"] == 0
    st = r["stage_seconds"]
    return {"value": round(n / dt / 1e6, 4), "unit": "Mpoints/s", "cores": threads, "kind": "port",
            "effective_parallelism": round(cpu / dt, 2),
            "stages_Mpoints_per_s": {k: round(n / v / 1e6, 3) for k, v in st.items() if v > 0},
            "sample": "%d uniform points, %s, d=%g, max_points_per_node=%d, %s, one batch (%.1f s wall, %.1f s CPU); threaded "
                      "like the reference: chunked encode on all threads, sort and root node on one, nodes >= 100000 points "
                      "as tasks" % (n, args.sampler, args.diagonal_fraction, args.max_points_per_node, args.strategy, dt, cpu)}


def payload_leg(args, ctx, swz, torch, dev, xyz, keys, perm, level, n):
    """node lists + permuted gather of positions and attribute columns into node order
    (what a lossless persistence writes).  Algorithmic bytes per point: 8 (perm + order) + 4 (composed index) +
    4 + 2 x 24 (positions) + 4 + 2 x row bytes per attribute."""
    names = [a for a in args.payload.split(",") if a]
    ws = ctx.workspace_bytes()
    ctx.release_workspace()  # the sampling workspace (MIN_DISTANCE: most of the HBM) is not needed any more
    cols = {}
    for a in names:
        idx, dt, width = swz.ATTRIBUTES[a]
        t = torch.empty((n, width) if width > 1 else (n,), dtype=getattr(torch, np.dtype(dt).name), device=dev)
        t.random_(0, 200) if not t.is_floating_point() else t.normal_()
        cols[a] = t
    order = torch.empty(n, dtype=torch.int32, device=dev)
    out_xyz = torch.empty_like(xyz)
    out_cols = {a: torch.empty_like(t) for a, t in cols.items()}
    res = {}
    for rep in range(2):  # first pass warms the workspace
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nodes = ctx.build_node_lists_device(keys.data_ptr(), level.data_ptr(), n, order.data_ptr())
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        ctx.gather_payload_device(perm.data_ptr(), order.data_ptr(), n, xyz.data_ptr(), {a: t.data_ptr() for a, t in cols.items()},
                                  out_xyz.data_ptr(), {a: t.data_ptr() for a, t in out_cols.items()})
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        row = sum(np.dtype(swz.ATTRIBUTES[a][1]).itemsize * swz.ATTRIBUTES[a][2] for a in names)
        alg = 12 + 52 + sum(4 + 2 * np.dtype(swz.ATTRIBUTES[a][1]).itemsize * swz.ATTRIBUTES[a][2] for a in names)
        res = {"tile_workspace_GB": round(ws / 1e9, 1), "columns": ["position"] + names, "attribute_bytes_per_point": row, "nodes": int(nodes["level"].shape[0]),
               "node_lists_ms": round((t1 - t0) * 1e3, 3), "gather_ms": round((t2 - t1) * 1e3, 3),
               "gather_Mpoints_per_s": round(n / (t2 - t1) / 1e6, 1), "algorithmic_bytes_per_point": alg,
               "gather_hbm_frac": round(alg * n / (t2 - t1) / (HBM_PEAK_GBS * 1e9), 4)}
    return res


def also_leg(args, ctx, swz, torch, dev, xyz, keys, perm, level, n, bmin, bmax, params, sname, fast):
    """One short leg beside the headline (after the timed region): another sampler or strategy on the first n points of the
    same cloud, with its own roofline object.  Never part of `value`."""
    import dataclasses
    lp = dataclasses.replace(params, sampler=swz.SAMPLERS[sname], strategy=swz.FAST if fast else swz.ACCURATE, flags=0)

    def lstep():
        return ctx.tile_device(xyz.data_ptr(), n, bmin, bmax, lp, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
    lstep()
    ctx.profile_reset()
    torch.cuda.synchronize(dev)
    lsteps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(lsteps):
        lstats = lstep()
    torch.cuda.synchronize(dev)
    ldt = (time.perf_counter() - t0) / lsteps
    lprof = ctx.profile_get()
    lvisit = lstats["points_visited"] / float(n)
    lalg = algorithmic_bytes_per_point(sname, lvisit)
    lroof = None
    if lprof:
        lname, lk = max(lprof.items(), key=lambda kv: kv[1]["total_ms"])
        lavg = lk["total_ms"] / max(lk["launches"], 1)
        lach = lk["algorithmic_bytes"] / max(lk["launches"], 1) / (lavg * 1e-3) / 1e9 if lavg > 0 else 0.0
        lroof = {"bound": "hbm", "kernel": lname, "achieved": round(lach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(lach / HBM_PEAK_GBS, 5), "launches": lk["launches"], "avg_launch_ms": round(lavg, 4)}
    largs = argparse.Namespace(steps=lsteps, sampler=sname)
    return {"workload": "the %s %d points, %s sampling, %s strategy" % ("same" if n == xyz.shape[0] else "first", n, sname, "FAST" if fast else "ACCURATE"),
            "points": n, "ms_per_step": round(ldt * 1e3, 3), "Mpoints_per_s": round(n / ldt / 1e6, 3), "visit_factor": round(lvisit, 4),
            "hbm_frac_end_to_end": round(lalg * n / ldt / (HBM_PEAK_GBS * 1e9), 5),
            "hbm_frac_end_to_end_implemented_sort": implemented_sort_frac(lprof, largs, lvisit, n, ldt),
            "roofline": lroof, "steps": lsteps,
            "kernels_ms_per_step": {k: round(v["total_ms"] / lsteps, 3) for k, v in sorted(lprof.items())}}


def multibatch_leg(args, ctx, swz, torch, dev, xyz, n, bmin, bmax, params):
    """--batches k: the same points through the multi-batch tiler, device resident (the timed `value`) and, with
    --staged, from pinned host memory with the copy of batch i+1 under the kernels of batch i."""
    k = args.batches
    bounds = [(i * n) // k for i in range(k + 1)]
    names = [a for a in args.payload.split(",") if a]

    def run_device():
        stats = []
        with swz.Tiler(ctx, bmin, bmax, params, capacity_hint=n) as t:
            for i in range(k):
                lo, hi = bounds[i], bounds[i + 1]
                stats.append(t.add_batch_device(xyz[lo:hi].data_ptr(), hi - lo))
            t.finalize()
            info = t.info()
        return stats, info

    out = {"run_device": run_device}
    if args.staged:
        host = swz.pinned_empty((n, 3), np.float64)
        torch.from_numpy(host).copy_(xyz)  # D2H once, outside every timed region
        cols = {}
        rng = np.random.default_rng(1)
        for a in names:
            idx, dt, width = swz.ATTRIBUTES[a]
            arr = swz.pinned_empty((n, width) if width > 1 else (n,), dt)
            arr[...] = rng.integers(0, 200, arr.shape).astype(dt)
            cols[a] = arr

        def run_staged():
            with swz.Tiler(ctx, bmin, bmax, params, capacity_hint=n) as t:
                def stage(i):
                    lo, hi = bounds[i], bounds[i + 1]
                    t.stage_batch(host[lo:hi], {a: c[lo:hi] for a, c in cols.items()})
                stage(0)
                for i in range(k):
                    if i + 1 < k:
                        stage(i + 1)
                    t.tile_staged()
                t.finalize()
                return t.info()
        out["run_staged"] = run_staged
        out["staged_bytes_per_point"] = 24 + sum(np.dtype(swz.ATTRIBUTES[a][1]).itemsize * swz.ATTRIBUTES[a][2] for a in names)
    return out


def main():
    args = parse_args()
    if args.driver == "group":
        sys.exit(run_group_driver(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("SWZ_BENCH_FORCE_SHARDED") != "1":
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE=%s\n" % (args.gpus, os.environ.get("WORLD_SIZE")))
        sys.exit(2)
    import torch

    import schwarzwald_amd as swz

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SWZ_BENCH_FORCE_SHARDED=1: run the sharded driver (RCCL init, exchange, shard API) even with one rank
    distributed = world > 1 or os.environ.get("SWZ_BENCH_FORCE_SHARDED") == "1"
    if distributed and "RANK" not in os.environ:  # SWZ_BENCH_FORCE_SHARDED=1 outside torchrun: a group of one
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # a free port: two such benches on one box must not collide
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
            sock.close()
    if distributed and args.one_device:
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo")
    elif distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if (distributed and not args.one_device) else 0)

    n = args.points
    bmin, bmax = [0.0, 0.0, 0.0], [args.bounds_scale] * 3
    spacing = swz.spacing_from_diagonal(bmin, bmax, args.diagonal_fraction)
    params = swz.TileParams(sampler=swz.SAMPLERS[args.sampler], max_points_per_node=args.max_points_per_node,
                            spacing_at_root=spacing, max_depth=100, strategy=getattr(swz, args.strategy),
                            fast_concurrency=args.fast_concurrency,
                            flags=swz.FLAG_MIN_DISTANCE_PROPERTY if args.md_mode == "property" else 0)
    ctx = swz.Context(dev.index)
    if os.environ.get("SWZ_BENCH_OWN_STREAM") != "1":  # default: share torch's current stream (ordered with its kernels)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    # synthetic input, resident in HBM before the timed region
    xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
    ctx.generate_uniform_device(SEED + 3, rank * n, n, xyz.data_ptr())
    if args.batches > 1 and args.batch_order == "tiles":
        # the same uniform points, batch i squeezed into tile i of a gx x gy grid over x and y (the whole z range): the
        # union is still uniform in the cube when the batches have equal sizes
        k = args.batches
        gx = int(np.ceil(np.sqrt(k)))
        gy = (k + gx - 1) // gx
        for i in range(k):
            lo, hi = (i * n) // k, ((i + 1) * n) // k
            tx, ty = i % gx, i // gx
            xyz[lo:hi, 0].mul_(args.bounds_scale / gx).add_(tx * args.bounds_scale / gx)
            xyz[lo:hi, 1].mul_(args.bounds_scale / gy).add_(ty * args.bounds_scale / gy)
        torch.cuda.synchronize(dev)

    mb = None
    if distributed and (args.batches > 1 or args.strategy == "FAST"):
        # (FAST on several GPUs goes through the batch tiler also with ONE batch: the single-batch sharded call is ACCURATE only)
        # BASELINE config 5's shape: every rank feeds its points in `batches` batches (attribute columns of --payload
        # along) to a sharded multi-batch tiler; with --staged the batches come from pinned host memory, the copy of
        # batch i+1 on a side stream under the tiling of batch i
        from schwarzwald_amd import sharded
        k = args.batches
        cuts = [(i * n) // k for i in range(k + 1)]
        names = [a for a in args.payload.split(",") if a]
        cols = {}
        for a in names:
            idx, dt, width = swz.ATTRIBUTES[a]
            tdt = {"uint8": torch.uint8, "int8": torch.int8, "uint16": torch.int16, "float32": torch.float32,
                   "float64": torch.float64}[np.dtype(dt).name]
            t = torch.empty((n, width) if width > 1 else (n,), dtype=tdt, device=dev)
            t.random_(0, 100) if not t.is_floating_point() else t.normal_()
            cols[a] = t
        host_xyz = host_cols = None
        if args.staged:
            host_xyz = torch.empty((n, 3), dtype=torch.float64, pin_memory=True)
            host_xyz.copy_(xyz)
            host_cols = {a: torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t) for a, t in cols.items()}
            torch.cuda.synchronize(dev)
        copy_stream = torch.cuda.Stream(device=dev)

        def fetch(i):
            lo, hi = cuts[i], cuts[i + 1]
            if not args.staged:
                return xyz[lo:hi], {a: t[lo:hi] for a, t in cols.items()}, None
            with torch.cuda.stream(copy_stream):
                bx = host_xyz[lo:hi].to(dev, non_blocking=True)
                bc = {a: t[lo:hi].to(dev, non_blocking=True) for a, t in host_cols.items()}
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return bx, bc, ev

        last = {}

        def step():
            runner = sharded.ShardedBatchTiler(ctx, dev, bmin, bmax, params, capacity_hint=int(1.3 * n) + 1024)
            last["runner"] = runner
            tot = dict(num_nodes=0, points_visited=0, max_level=-1, fast_start_levels=-1, num_levels=0, min_distance_rounds=0)
            nxt = fetch(0)
            for i in range(k):
                bx, bc, ev = nxt
                if i + 1 < k:
                    nxt = fetch(i + 1)
                if ev is not None:
                    torch.cuda.current_stream(dev).wait_event(ev)
                st = runner.add_batch(bx, bc)
                for key in ("points_visited", "num_levels", "min_distance_rounds"):
                    tot[key] += st[key]
                tot["max_level"] = max(tot["max_level"], st["max_level"])
            runner.finalize()
            tot["num_nodes"] = int(runner.tiler.info()["num_nodes"])
            tot["shard_points"] = int(runner.tiler.info()["num_points"])
            last["stage_ms"] = runner.stage_timings()
            last["root_mode"] = runner.root_mode
            runner.close()
            return tot
    elif distributed:
        from schwarzwald_amd import sharded
        runner = sharded.ShardedTiler(ctx, dev, bmin, bmax, params)

        def step():
            return runner.tile(xyz)
    elif args.batches > 1:
        mb = multibatch_leg(args, ctx, swz, torch, dev, xyz, n, bmin, bmax, params)
        mb_info = {}

        def step():
            per_batch, info = mb["run_device"]()
            mb_info.update(info)
            return dict(num_nodes=int(info["num_nodes"]), points_visited=sum(b["points_visited"] for b in per_batch),
                        max_level=max(b["max_level"] for b in per_batch), fast_start_levels=int(info["fast_start_levels"]),
                        num_levels=sum(b["num_levels"] for b in per_batch),
                        min_distance_rounds=sum(b["min_distance_rounds"] for b in per_batch))
    else:
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        level = torch.empty(n, dtype=torch.int8, device=dev)

        def step():
            return ctx.tile_device(xyz.data_ptr(), n, bmin, bmax, params, keys.data_ptr(), perm.data_ptr(),
                                   level.data_ptr())

    def barrier():
        if distributed:
            torch.cuda.synchronize(dev)
            dist.barrier()
        torch.cuda.synchronize(dev)

    extra_warmups = 0
    first_data_set_ms = None
    warm_left = args.warmup
    if mb is not None:
        # what a one-shot user pays: the FIRST data set of this context, workspace allocation and all (a real `Schwarzwald
        # --tiler` run is one data set, executable/main.cpp:233-301); it doubles as the first warm-up step.  (The driver
        # wipes the memory of a process that has exited, and the allocations of the next one wait for it: 0.95 s on an
        # idle device, 4.2 s when this process starts right after one that held 100 GB -- tools/lists/first_data_set.txt.)
        torch.cuda.synchronize(dev)
        tf = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        first_data_set_ms = (time.perf_counter() - tf) * 1e3
        warm_left = max(0, warm_left - 1)
    for _ in range(warm_left):
        step()
    if mb is not None:
        # The multi-batch tiler's workspace (node store sides, merge buffers) is sized by what the data sets before have asked
        # for and may still grow in the second or third one (the store's compaction points move with the capacities it
        # finds); a multi-GB hipMalloc in the middle of a data set stalls it for 0.25-2 s (tools/default_op_probe.py).  The
        # timed steps are the steady state: warm up until a whole data set leaves the workspace as it found it -- at most two
        # more (in repeated data sets on one context the workspace keeps creeping up by ~0.5 GB per data set: the
        # log-structured store's compaction points move with the capacities it finds).
        held = ctx.workspace_bytes()
        while extra_warmups < 2:
            step()
            extra_warmups += 1
            now = ctx.workspace_bytes()
            grew = now > held
            held = now
            if not grew:
                break
    if not args.no_profile:
        ctx.profile_enable(True)
        ctx.profile_reset()
    barrier()
    held_before = ctx.workspace_bytes()
    t0 = time.perf_counter()
    stats = None
    for _ in range(args.steps):
        stats = step()
    barrier()
    elapsed = time.perf_counter() - t0
    workspace_grew = ctx.workspace_bytes() > held_before  # (an allocation inside the timed steps: the line says so)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = {} if args.no_profile else ctx.profile_get()
    # the sharded single-batch driver: what every rank's last step spent where (events on the launch stream), which way the
    # MIN_DISTANCE root went, how many points the rank ended up owning
    shard_report = None
    if distributed:
        batch_path = args.batches > 1 or args.strategy == "FAST"
        mine = {"rank": rank, "root_mode": last.get("root_mode") if batch_path else getattr(runner, "root_mode", None),
                "shard_points": int(stats.get("shard_points", 0)) if stats else 0}
        mine.update(last.get("stage_ms", {}) if batch_path else runner.stage_timings())  # (batch path: summed over the batches of the last step)
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        shard_report = gathered
    total_points = n * world
    ms_per_step = elapsed * 1e3 / args.steps
    value = total_points / (elapsed / args.steps) / 1e6
    visit = stats["points_visited"] / float(n) if stats else 0.0

    if rank == 0:
        roofline = None
        if prof:
            name, k = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
            avg_ms = k["total_ms"] / max(k["launches"], 1)
            alg_per_launch = k["algorithmic_bytes"] / max(k["launches"], 1)
            achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": measured_traffic(name, n, args.sampler)[0],
                        "traffic_source": measured_traffic(name, n, args.sampler)[1],
                        "launches": k["launches"], "avg_launch_ms": round(avg_ms, 4)}
        alg = algorithmic_bytes_per_point(args.sampler, visit)
        out = {
            "metric": "Mpoints/s end-to-end tile (Morton+sort+sample)", "value": round(value, 3),
            "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.total_points else "weak", "vs_baseline": None,
            "dtype": "u64 keys / f64 positions", "data": "synthetic",
            "config": {"workload": "%s%d uniform points per GPU in the unit cube, %s sampling, spacing = diagonal/%g, "
                                   "max_points_per_node=%d, %s strategy, %s" % (
                                       ("BASELINE config %d%s: %d points in total, " % (args.config, " (scaled down)" if (
                                           args.total_points != {4: 1_000_000_000, 5: 4_000_000_000}[args.config]) else "", args.total_points))
                                       if args.config else ("%d points in total, " % args.total_points if args.total_points else ""),
                                       n, args.sampler, args.diagonal_fraction, args.max_points_per_node, args.strategy,
                                       "one batch" if args.batches <= 1 else "%d batches through the multi-batch tiler (%s)" % (
                                           args.batches, "cut out of the whole cloud" if args.batch_order == "uniform" else "one x-y tile each")),
                       "points_per_gpu": n, "sampler": args.sampler, "strategy": args.strategy,
                       "min_distance_mode": ("property" if args.md_mode == "property" else "exact") if args.sampler == "MIN_DISTANCE" else None,
                       "batches": args.batches, "batch_order": args.batch_order if args.batches > 1 else None,
                       "staged_from_pinned_host": bool(args.staged and distributed and args.batches > 1) or None,
                       "payload_columns": [a for a in args.payload.split(",") if a] if (distributed and args.batches > 1) else None,
                       "baseline_config": args.config or None, "total_points": args.total_points or None,
                       "parallelism": "1 GPU" if world == 1 else ("%d ranks sharded by top Morton bits, one all-to-all%s" % (
                           world, " -- DRY RUN: all ranks on ONE GPU over gloo" if args.one_device else ""))},
            "extra_warmup_data_sets": extra_warmups if mb is not None else None,
            "first_data_set_ms": round(first_data_set_ms, 3) if first_data_set_ms is not None else None,
            "workspace_grew_in_timed_steps": bool(workspace_grew),
            "ranks_in_process_group": dist.get_world_size() if distributed else 1,
            "root_mode": shard_report[0]["root_mode"] if shard_report else None,
            "exchange_ms": max((r.get("exchange_ms", 0.0) for r in shard_report), default=None) if shard_report and not shard_stamp_error(shard_report, ms_per_step) else None,
            "stamp_error": shard_stamp_error(shard_report, ms_per_step) if shard_report else None,
            "shards": shard_report,
            "visit_factor": round(visit, 4),
            "hbm_frac_end_to_end": round(alg * total_points / world / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9), 5),
            "algorithmic_bytes_per_point": round(alg, 1),
            "hbm_frac_end_to_end_implemented_sort": implemented_sort_frac(prof, args, visit, total_points / world, elapsed / args.steps),
            "tile_stats": stats,
            "roofline": roofline,
            "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())},
        }
        if args.md_mode == "both" and args.sampler == "MIN_DISTANCE" and not distributed and mb is None:
            import dataclasses
            pp = dataclasses.replace(params, flags=swz.FLAG_MIN_DISTANCE_PROPERTY)

            def pstep():
                return ctx.tile_device(xyz.data_ptr(), n, bmin, bmax, pp, keys.data_ptr(), perm.data_ptr(), level.data_ptr())
            pstep()
            ctx.profile_reset()
            torch.cuda.synchronize(dev)
            psteps = max(1, min(args.steps, 5))  # (beside the headline, after the timed region: a handful of steps is enough)
            t0 = time.perf_counter()
            for _ in range(psteps):
                pstats = pstep()
            torch.cuda.synchronize(dev)
            pdt = (time.perf_counter() - t0) / psteps
            pprof = ctx.profile_get()
            pvisit = pstats["points_visited"] / float(n)
            palg = algorithmic_bytes_per_point(args.sampler, pvisit)
            proof = None
            if pprof:  # the roofline object of the property-mode run: its own dominant kernel class
                pname, pk = max(pprof.items(), key=lambda kv: kv[1]["total_ms"])
                pavg = pk["total_ms"] / max(pk["launches"], 1)
                pach = pk["algorithmic_bytes"] / max(pk["launches"], 1) / (pavg * 1e-3) / 1e9 if pavg > 0 else 0.0
                ptraffic = measured_traffic(pname, n, args.sampler, "bytes_per_launch_property_mode")
                proof = {"bound": "hbm", "kernel": pname, "achieved": round(pach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(pach / HBM_PEAK_GBS, 5), "traffic": ptraffic[0], "traffic_source": ptraffic[1],
                         "launches": pk["launches"], "avg_launch_ms": round(pavg, 4)}
            out["min_distance_property"] = {
                "roofline": proof,
                "ms_per_step": round(pdt * 1e3, 3), "Mpoints_per_s": round(n / pdt / 1e6, 3), "visit_factor": round(pvisit, 4),
                "hbm_frac_end_to_end": round(palg * n / pdt / (HBM_PEAK_GBS * 1e9), 5), "tile_stats": pstats,
                "steps": psteps,
                "kernels_ms_per_step": {k: round(v["total_ms"] / psteps, 3) for k, v in sorted(pprof.items())},
                "note": "same workload with SWZ_FLAG_MIN_DISTANCE_PROPERTY: spacing and maximality guaranteed "
                        "(tests/test_min_distance_property.py), taken set differs from the reference's"}
        if (args.also and world == 1 and not distributed and mb is None and args.md_mode != "property" and args.sampler == "MIN_DISTANCE"
                and args.strategy == "ACCURATE" and not args.payload):
            import dataclasses
            out["also"] = {}
            for leg in [x for x in args.also.split(",") if x]:
                # NAME[_FAST][@POINTS]: POINTS < n runs the leg on the first POINTS points of the same cloud (BASELINE
                # configs[1] is GRID_CENTER at 100 M points)
                spec, _, at = leg.partition("@")
                fast = spec.endswith("_FAST")
                sname = spec[:-5] if fast else spec
                ln = min(n, int(at)) if at else n
                if sname not in swz.SAMPLERS or (sname == args.sampler and (args.strategy == "FAST") == fast and ln == n):
                    continue
                try:  # an optional leg must never cost the line its headline (out of memory for a grid table, ...)
                    out["also"][leg] = also_leg(args, ctx, swz, torch, dev, xyz, keys, perm, level, ln, bmin, bmax, params, sname, fast)
                except Exception as e:  # noqa: BLE001
                    out["also"][leg] = {"error": "%s: %s" % (type(e).__name__, e)}
        if mb is not None and "run_staged" in mb:
            mb["run_staged"]()  # warm-up: pools and workspace sized
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            info = mb["run_staged"]()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            copied = int(info["staged_bytes"])
            out["staged"] = {"ms": round(dt * 1e3, 3), "Mpoints_per_s_pcie_inclusive": round(n / dt / 1e6, 3),
                             "h2d_bytes": copied, "h2d_bytes_per_point": mb["staged_bytes_per_point"],
                             "h2d_GBs_if_serial": round(copied / dt / 1e9, 2),
                             "copy_wait_ms": round(float(info["staged_wait_ms"]), 3),
                             "device_resident_ms": round(ms_per_step, 3),
                             "note": "batch i+1 is copied from pinned host memory (hipMemcpyAsync, copy stream) while "
                                     "batch i is tiled; copy_wait_ms is the time tiling had to wait for a copy"}
        if world == 1 and not distributed and args.payload and mb is None:
            out["payload"] = payload_leg(args, ctx, swz, torch, dev, xyz, keys, perm, level, n)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args, spacing)
        elif world > 1:
            out["cpu_baseline_reference"] = committed_cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
