"""Debugging aid for swz_mdblock.hip: one small tile against the oracle with a set of options, first mismatches printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import schwarzwald_amd as swz
UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
rng = np.random.default_rng(11)
xyz = rng.random((int(sys.argv[1]) if len(sys.argv) > 1 else 400000, 3))
d, mppn = 250, 2000
spacing = O.spacing_from_diagonal(*UNIT, d)
o = O.tile(xyz, *UNIT, O.MIN_DISTANCE, mppn, spacing)
ctx = swz.Context(0)
import itertools
sets = [{"SWZ_SP_BLOCK": "0"}, {}, {}, {"SWZ_SP_BLOCK_WIDE": "1"}, {"SWZ_MD_SPARSE_LIMIT": "1000"}]
for opts in sets:
    for k, v in opts.items():
        ctx.set_option(k, v)
    g = ctx.tile(xyz, *UNIT, swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=mppn, spacing_at_root=spacing))
    bad = np.nonzero(g.level != o["level"])[0]
    print(opts, "mismatches", bad.size, "of", g.level.size, "first", bad[:8], "gpu", g.level[bad[:8]], "oracle", o["level"][bad[:8]], flush=True)
    for k in opts:
        ctx.set_option(k, None)

