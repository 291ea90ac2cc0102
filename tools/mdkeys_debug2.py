"""Debug helper: one level (the root) through swz_sample_points, keys vs positions vs oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O
import schwarzwald_amd as swz

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(99)
xyz = rng.random((n, 3))
xyz = np.clip(np.vstack([xyz, 0.25 + 0.01 * rng.standard_normal((n // 8, 3))]), 0.0, 1.0)
spacing = O.spacing_from_diagonal(*UNIT, 250)
keys, _ = O.index_points(xyz, *UNIT)
perm = np.lexsort((np.arange(len(keys)), keys)).astype(np.uint32)
skeys = keys[perm]
with swz.Context(0) as ctx:
    ctx.set_option("SWZ_MD_SPARSE_LIMIT", "0")
    for k, v in [a.split("=") for a in sys.argv[2:]]:
        ctx.set_option(k, v)
    ctx.set_option("SWZ_MD_KEYS", "0")
    ref = ctx.sample_points(swz.MIN_DISTANCE, 1000, skeys, perm, xyz, 0, -1, *UNIT, spacing)
    ctx.set_option("SWZ_MD_KEYS", None)
    ctx.set_option("SWZ_DEBUG", "1")
    got = ctx.sample_points(swz.MIN_DISTANCE, 1000, skeys, perm, xyz, 0, -1, *UNIT, spacing)
bad = np.flatnonzero(ref != got)
print("taken ref %d got %d, differ %d" % (ref.sum(), got.sum(), bad.size))
P = xyz[perm]
s2 = float(np.float32(spacing) * np.float32(spacing))
cell_shift = 63 - 21
cells = skeys >> np.uint64(cell_shift)
starts = np.flatnonzero(np.r_[True, cells[1:] != cells[:-1]])
cell_of = np.cumsum(np.r_[True, cells[1:] != cells[:-1]]) - 1
sizes = np.diff(np.r_[starts, len(skeys)])
for i in bad[:12]:
    d2 = ((P[:i] - P[i]) ** 2).sum(1)
    near = np.flatnonzero((d2 < s2) & (ref[:i] == 1))
    c = cell_of[i]
    print("pos %d ref %d got %d cell %d (size %d, offset in cell %d)" % (i, ref[i], got[i], c, sizes[c], i - starts[c]))
    for q in near[:4]:
        cq = cell_of[q]
        nacc = int(ref[starts[cq]:starts[cq] + sizes[cq]].sum())
        rank = int(ref[starts[cq]:q].sum())
        print("    earlier accepted within s: pos %d cell %d (size %d, offset %d, accepted in that cell %d, this is #%d) got-flag %d d=%.1f cells" % (
            q, cq, sizes[cq], q - starts[cq], nacc, rank, got[q], np.sqrt(d2[q]) * 2097152))
