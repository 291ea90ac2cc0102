#!/usr/bin/env python3
"""Per-level breakdown of the MIN_DISTANCE round kernels from a rocprofv3 kernel trace (csv)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("swz::", "")))
rows.sort()
level = -1
cur = None
levels = []
for s, e, k in rows:
    if k in ("md_fill_queue_kernel", "md_lazy_start_kernel"):
        cur = {"k": collections.defaultdict(lambda: [0, 0]), "gap": 0, "last": None, "rounds": [], "t0": s}
        levels.append(cur)
    if cur is None or k not in ("md_sweep_kernel", "md_commit_kernel", "md_requeue_kernel"):
        if k.startswith("md_") or k.startswith("sp_"):
            continue
        cur = None if k in ("md_gather_active_kernel",) else cur
        continue
    d = cur["k"][k]
    d[0] += 1
    d[1] += e - s
    if cur["last"] is not None:
        cur["gap"] += max(0, s - cur["last"])
    cur["last"] = e
    if k == "md_sweep_kernel":
        cur["rounds"].append([s, e - s, 0, 0])
    elif k == "md_commit_kernel":
        cur["rounds"][-1][2] = e - s
    else:
        cur["rounds"][-1][3] = e - s
        cur["t1"] = e
for i, L in enumerate(levels):
    if not L["rounds"]:
        continue
    n = len(L["rounds"])
    tot = (L["t1"] - L["t0"]) / 1e6
    print("level#%d: %d rounds, wall %.1f ms; sweep %.1f commit %.1f requeue %.1f gaps %.1f ms" % (
        i, n, tot, L["k"]["md_sweep_kernel"][1] / 1e6, L["k"]["md_commit_kernel"][1] / 1e6,
        L["k"]["md_requeue_kernel"][1] / 1e6, L["gap"] / 1e6))
    # per 10% of rounds
    for b in range(10):
        seg = L["rounds"][b * n // 10:(b + 1) * n // 10]
        if not seg:
            continue
        print("   rounds %5d..: sweep %6.1f us  commit %5.1f us  requeue %5.1f us (avg per round)" % (
            b * n // 10, sum(x[1] for x in seg) / len(seg) / 1e3, sum(x[2] for x in seg) / len(seg) / 1e3,
            sum(x[3] for x in seg) / len(seg) / 1e3))
