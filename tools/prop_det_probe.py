"""Debug helper: property mode twice on the same input, where do the results differ?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as O
import schwarzwald_amd as swz
d, max_points = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(d)
xyz = rng.random((400000, 3))
UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
sp = O.spacing_from_diagonal(*UNIT, d)
ctx = swz.Context(0)
params = swz.TileParams(sampler=swz.MIN_DISTANCE, max_points_per_node=max_points, spacing_at_root=sp, flags=swz.FLAG_MIN_DISTANCE_PROPERTY)
r = ctx.tile(xyz, *UNIT, params)
for k in range(3):
    r2 = ctx.tile(xyz, *UNIT, params)
    diff = np.nonzero(r.level != r2.level)[0]
    print("run", k, "differences:", diff.size, "levels", np.unique(r.level[diff]), np.unique(r2.level[diff]), "first", diff[:5])
