#!/usr/bin/env python3
"""Derives profiles/rNN/traffic.json from the two per-kernel PMC summaries tools/profile_round.sh writes
(FETCH_SIZE and WRITE_SIZE, KiB, separate --pmc passes of `bench.py --steps 1 --warmup 0`).
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE counts half the bytes on gfx950
(/opt/skills/guides/MI355X_MICROARCH.md); the factor is re-checked on radix_hist_kernel, which reads
8 passes x 8 B x points and nothing else."""
import csv, json, sys, os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01"
points = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 5  # sampled levels per step (root .. 3)


def load(name):
    out = {}
    with open(os.path.join(d, name)) as f:
        next(f)
        for line in f:
            k, n, v = line.rstrip("\n").rsplit(",", 2)
            out[k] = (int(n), float(v))
    return out


fetch, write = load("pmc_FETCH_SIZE_by_kernel.csv"), load("pmc_WRITE_SIZE_by_kernel.csv")


def bytes_of(pred):
    ks = sorted(k for k in set(fetch) | set(write) if pred(k))
    b = sum((2.0 * fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1]) * 1024.0 for k in ks)
    return ks, b


# calibration on a kernel whose reads are known exactly: the one-sweep histogram reads every key once (8 B x points);
# (round 1: radix_hist_kernel, eight passes over the keys)
if "swz::radix_ghist_kernel" in fetch:
    calib = 2.0 * fetch["swz::radix_ghist_kernel"][1] * 1024.0 / (8.0 * points)
else:
    calib = 2.0 * fetch["swz::radix_hist_kernel"][1] * 1024.0 / (8 * 8.0 * points)
# (templated kernels are listed as "void swz::md_sweep_kernel<1, false>"; the fused cell scan belongs to the class too)
# (round 6: the sparse levels run sb_block_kernel / sb_table_kernel, swz_mdblock.hip)
md_k, md_b = bytes_of(lambda k: "swz::md_" in k or "swz::sp_" in k or "swz::sb_" in k or "swz::mq_" in k or "swz::CellHeadF" in k or "swz::MqHeadF" in k)
rs_k, rs_b = bytes_of(lambda k: k in ("swz::radix_scatter_kernel", "swz::radix_onesweep_kernel"))
# launches of the scatter kernel in the profiled step: the passes over the whole input (the eight tiny passes that sort
# the sample which picks the number of top digits do not count)
rs_disp = sum(fetch.get(k, (0, 0))[0] for k in rs_k)
rs_launches = rs_disp - 8 if rs_disp > 8 else max(1, rs_disp)
md_lo = sum((fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1]) * 1024.0 for k in md_k)
import bench  # library_source_sha16(): bench.py only quotes these numbers for the kernel sources they were measured on
# (the hash tools/profile_round.sh took on the GPU box next to the counters, if it did; else the sources as they are now)
sha_file = os.path.join(d, "source_sha16.txt")
source_sha16 = open(sha_file).read().strip() if os.path.exists(sha_file) else bench.library_source_sha16()
out = {
    "points": points, "sampler": "MIN_DISTANCE", "source_sha16": source_sha16,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 0`; "
              "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md); "
              "check: corrected reads of the radix histogram kernel / its exact key bytes = %.4f.  The factor 2 also holds for "
              "scattered 8-byte loads (tools/fetch_calib.hip: 268 M loads in 268 M different lines -> FETCH_SIZE 16 GiB, "
              "TCC_EA0_RDREQ_128B one per load: the L2 reads whole 128-byte lines, also for non-temporal and agent-scope "
              "atomic loads), so bytes_per_launch is the traffic, not an upper bound; the uncorrected sum is kept as "
              "bytes_per_launch_lower_bound for comparison with round 1" % calib,
    "bytes_per_launch": {"sample_min_distance": md_b / levels, "radix_scatter": rs_b / rs_launches},
    "bytes_per_launch_lower_bound": {"sample_min_distance": md_lo / levels},
    "detail": {
        "sample_min_distance": {"kernels": md_k, "launches_per_step": levels, "bytes_per_step": md_b},
        "radix_scatter": {"kernels": rs_k, "launches_per_step": rs_launches, "bytes_per_step": rs_b},
    },
}
# the property-mode run of the same workload (its own two --pmc passes): the rounds on the dense levels (pr_* kernels and
# their alive scans: class sample_min_distance_property, one launch per dense level) and the exact sparse levels below
try:
    fetch, write = load("pmc_FETCH_SIZE_by_kernel_property_mode.csv"), load("pmc_WRITE_SIZE_by_kernel_property_mode.csv")
    pr_k, pr_b = bytes_of(lambda k: "swz::pr_" in k or "swz::PrAlive" in k or "swz::PrList" in k)
    sp_k, sp_b = bytes_of(lambda k: "swz::sp_" in k or "swz::sb_" in k)
    dense_levels = max(1, levels - 2)
    out["bytes_per_launch_property_mode"] = {"sample_min_distance_property": pr_b / dense_levels, "sample_min_distance": sp_b / 2.0}
    out["detail"]["property_mode"] = {"sample_min_distance_property": {"kernels": pr_k, "launches_per_step": dense_levels, "bytes_per_step": pr_b},
                                      "sample_min_distance": {"kernels": sp_k, "launches_per_step": 2, "bytes_per_step": sp_b}}
except FileNotFoundError:
    pass
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps(out["bytes_per_launch"]), "calibration", calib)
