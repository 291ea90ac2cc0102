"""Debug helper: key sweep vs oracle under a few option sets; prints where they differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O
import schwarzwald_amd as swz

UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 250
mppn = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
rng = np.random.default_rng(99)
xyz = rng.random((n, 3))
if os.environ.get('BLOB'):
    xyz = np.clip(np.vstack([xyz, 0.25 + 0.01 * rng.standard_normal((n // 8, 3))]), 0.0, 1.0)
spacing = O.spacing_from_diagonal(*UNIT, d)
o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, mppn, spacing)
sets = [
    {"SWZ_MD_KEYS": "0"},
    {},
    {"SWZ_MD_LAZY": "0"},
    {"SWZ_MD_LAZY": "0", "SWZ_MD_KEYS_BAND": "1e9"},
    {"SWZ_MD_LAZY": "0", "SWZ_MD_ABLATE": "8"},
    {"SWZ_MD_BIG": "1"},
    {"SWZ_MD_BIG": "0"},
    {"SWZ_MD_DENSITY": "0"},
]
with swz.Context(0) as ctx:
    ctx.set_option("SWZ_MD_SPARSE_LIMIT", "0")
    if os.environ.get("DBG"):
        ctx.set_option("SWZ_DEBUG", "1")
    for opts in sets:
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
            bad = np.flatnonzero(g.level != o["level"])
            msg = "%d differ" % bad.size
            if bad.size:
                gl, ol = g.level[bad], o["level"][bad]
                for lv in range(-1, 6):
                    extra = int(np.sum(gl == lv))
                    missed = int(np.sum(ol == lv))
                    if extra or missed:
                        msg += " | L%d: gpu took %d wrongly, missed %d" % (lv, extra, missed)
            print(opts, "->", msg, "rounds", g.stats["min_distance_rounds"], flush=True)
        except Exception as ex:
            print(opts, "-> EXC", ex, flush=True)
        for k in opts:
            ctx.set_option(k, None)
