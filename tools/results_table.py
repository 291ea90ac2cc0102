#!/usr/bin/env python3
"""The results table of DESIGN.md section 6 from the committed lines of a profile round: results_table.py [profiles/r06]"""
import json, os, re, sys
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06")


def line(name):
    try:
        return json.loads(open(os.path.join(D, name)).read().strip().splitlines()[-1])
    except Exception:
        return None


def t(ms):
    return "%.2f s" % (ms / 1e3) if ms >= 1500 else ("%.0f ms" % ms if ms >= 200 else "%.1f ms" % ms)


def r(x):
    return "{:,.0f}".format(x).replace(",", " ")


rows = []
h = line("bench_1B_min_distance.json")
if h:
    rows.append(("1 B, MIN_DISTANCE exact (headline)", "**%s**" % t(h["ms_per_step"]), r(h["value"])))
    p = h.get("min_distance_property")
    if p:
        rows.append(("1 B, MIN_DISTANCE property mode", "**%s**" % t(p["ms_per_step"]), r(p["Mpoints_per_s"])))
g = [line("bench_1B_%s.json" % s) for s in ("RANDOM_GRID", "GRID_CENTER", "JITTERED")]
if all(g):
    rows.append(("1 B, RANDOM_GRID / GRID_CENTER / JITTERED", " / ".join(t(x["ms_per_step"]) for x in g), " / ".join(r(x["value"]) for x in g)))
x = line("bench_100M_grid_center.json")
if x:
    rows.append(("100 M, GRID_CENTER (BASELINE configs[1])", t(x["ms_per_step"]), r(x["value"])))
x = line("bench_1B_min_distance_FAST.json")
if x:
    rows.append(("1 B, MIN_DISTANCE exact, FAST strategy, one batch", t(x["ms_per_step"]), r(x["value"])))
for order, what in (("tiles", "batches as x-y tiles"), ("uniform", "every batch cut out of the whole cloud")):
    a, b = line("bench_1B_100batches_MIN_DISTANCE_FAST_%s.json" % order), line("bench_1B_100batches_MIN_DISTANCE_FAST_%s_property.json" % order)
    if a and b:
        rows.append(("the reference's default operating point (MIN_DISTANCE, FAST, 1 B points in 100 batches of 10 M), %s: exact / property; first data set of the context" % what,
                     "**%s / %s**; %s / %s" % (t(a["ms_per_step"]), t(b["ms_per_step"]), t(a["first_data_set_ms"]), t(b["first_data_set_ms"])),
                     "%s / %s" % (r(a["value"]), r(b["value"]))))
a, b = line("bench_1B_100batches_MIN_DISTANCE_tiles.json"), line("bench_1B_100batches_MIN_DISTANCE_uniform.json")
if a and b:
    rows.append(("the same in ACCURATE: tiles / uniform, exact", "%s / %s" % (t(a["ms_per_step"]), t(b["ms_per_step"])), "%s / %s" % (r(a["value"]), r(b["value"]))))
a, b = line("bench_1B_100batches_RANDOM_GRID_tiles.json"), line("bench_1B_100batches_RANDOM_GRID_uniform.json")
if a and b:
    rows.append(("1 B in 100 batches, RANDOM_GRID, ACCURATE: tiles / uniform", "%s / %s" % (t(a["ms_per_step"]), t(b["ms_per_step"])), "%s / %s" % (r(a["value"]), r(b["value"]))))
x = line("bench_500M_5batches_staged.json")
if x:
    rows.append(("500 M in 5 batches, MIN_DISTANCE exact, staged from pinned host memory with RGB + intensity", t(x["ms_per_step"]), r(x["value"])))
x = line("bench_group_driver_8shards_1device.json")
if x:
    rows.append(("8 shards x 25 M on ONE device through the C++ group driver, one batch", t(x["ms_per_step"]), r(x["value"])))
try:
    c = open(os.path.join(D, "clustered_100M.txt")).read()
    by = {}
    for m in re.finditer(r"N=100000000 (\S+)\s*(property)?\s*: ([0-9.]+) ms = ([0-9]+) Mpts/s", c):
        by.setdefault(m.group(1) + (" property" if m.group(2) else ""), []).append((float(m.group(3)), int(m.group(4))))
    if by:
        ks = [k for k in ("MIN_DISTANCE", "MIN_DISTANCE property", "GRID_CENTER") if k in by]
        rows.append(("100 M surface-like clustered points: " + " / ".join(ks),
                     " / ".join("%s" % "–".join(sorted({"%.0f" % v[0] if v[0] >= 100 else "%.1f" % v[0] for v in by[k]})) for k in ks) + " ms",
                     " / ".join("–".join(sorted({r(v[1]) for v in by[k]})) for k in ks)))
except OSError:
    pass
if h and h.get("cpu_baseline"):
    cb = h["cpu_baseline"]
    rows.append(("CPU oracle threaded like the reference, %d threads of the GPU box: %s" % (cb["cores"], cb["sample"].split(" (")[0]),
                 re.search(r"\(([0-9.]+ s) wall", cb["sample"]).group(1) if re.search(r"\(([0-9.]+ s) wall", cb["sample"]) else "", "%.2f" % cb["value"]))
for a, b, c in rows:
    print("| %s | %s | %s |" % (a, b, c))
if h:
    ro = h["roofline"]
    print()
    print("roofline: launches/step %d, avg launch %.2f ms, achieved %.1f GB/s, frac %.4f, traffic %s" % (
        ro["launches"] // max(1, h["steps"]), ro["avg_launch_ms"], ro["achieved"], ro["frac"], ro.get("traffic")))
    print("e2e %.4f, implemented sort %.4f; classes %s" % (h["hbm_frac_end_to_end"], h["hbm_frac_end_to_end_implemented_sort"]["frac"],
                                                         {k: round(v, 1) for k, v in h["kernels_ms_per_step"].items()}))
    if h.get("min_distance_property"):
        print("property classes", {k: round(v, 1) for k, v in h["min_distance_property"]["kernels_ms_per_step"].items()})
