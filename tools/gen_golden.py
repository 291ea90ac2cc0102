#!/usr/bin/env python3
"""Generates tests/golden/ (run in the build container, where /root/reference exists).

Two kinds of fixtures, kept apart:
  ref_*.npz           outputs of the REFERENCE's own code: oracle/_ref/libswzref.so is compiled by oracle/Makefile from
                      core/datastructures/MortonIndex.h, util/algorithms/Algorithm.h and util/containers/Range.h where they
                      lie under /root/reference (nothing is copied); inputs are seeded here.
  oracle_tile_*.json  SHA-256 digests of the oracle's tile outputs for small seeded batches.  They pin the oracle (and,
                      through the GPU tests, the HIP path) against silent drift; they are NOT reference outputs -- the
                      tiler itself cannot be built here.
Fixtures are data only (inputs + expected outputs)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
UNIT = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])


def ref_morton_index(R):
    rng = np.random.default_rng(20261001)
    n = 4096
    keys = rng.integers(0, 2**63, n, dtype=np.uint64)
    keys[:4] = [0, 2**63 - 1, 0x7000000000000000, 0x1249249249249249]
    level = rng.integers(0, 21, n).astype(np.uint32)
    octant = rng.integers(0, 8, n).astype(np.uint8)
    trunc = np.array([R.ref_truncate_to_level(int(k), int(l), 21) for k, l in zip(keys, level)], dtype=np.uint64)
    octs = np.array([R.ref_get_octant_at_level(int(k), int(l), 21) for k, l in zip(keys, level)], dtype=np.uint8)
    seto = np.array([R.ref_set_octant_at_level(int(k), int(l), int(o), 21) for k, l, o in zip(keys, level, octant)],
                    dtype=np.uint64)
    names = []
    buf = C.create_string_buffer(64)
    for k in keys[:256]:
        R.ref_morton64_to_string(int(k), 21, buf, 64)
        names.append(buf.value.decode())
    np.savez_compressed(os.path.join(OUT, "ref_morton_index.npz"), keys=keys, level=level, octant=octant,
                        truncate_to_level=trunc, get_octant_at_level=octs, set_octant_at_level=seto,
                        to_string_first_256=np.array(names))


def ref_random_grid(R):
    """RandomSortedGridSampling's partition (Sampling.h:253-284) through the reference's stable_partition_with_jumps."""
    rng = np.random.default_rng(7)
    cases = {}
    for tag, n, level in (("uniform_l3", 8000, 3), ("uniform_l6", 8000, 6), ("clustered_l5", 12000, 5)):
        xyz = rng.random((n, 3)) if tag.startswith("uniform") else np.clip(0.5 + 0.05 * rng.standard_normal((n, 3)), 0, 1)
        keys = O.index_points(xyz, *UNIT)[0]
        perm = O.sort_by_key(keys)
        k = np.ascontiguousarray(keys[perm])
        i = np.ascontiguousarray(perm.astype(np.uint32))
        ok, oi = k.copy(), i.copy()
        taken = R.ref_partition_first_of_cell(ok.ctypes.data_as(C.POINTER(C.c_uint64)), oi.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              n, level)
        cases[tag + "_level"] = np.array(level)
        cases[tag + "_sorted_keys"] = k
        cases[tag + "_sorted_idx"] = i
        cases[tag + "_taken"] = np.array(taken)
        cases[tag + "_out_idx"] = oi
    np.savez_compressed(os.path.join(OUT, "ref_random_grid_partition.npz"), **cases)


def ref_algorithm(R):
    rng = np.random.default_rng(11)
    vals = rng.integers(-1000, 1000, 5000).astype(np.int32)
    out = vals.copy()
    pivot = R.ref_stable_partition_take_multiples(out.ctypes.data_as(C.POINTER(C.c_int32)), len(out), 7)
    ranges = [np.sort(rng.integers(-10**6, 10**6, m)).astype(np.int32) for m in (0, 1, 17, 1000, 4096, 3)]
    ptrs = (C.POINTER(C.c_int32) * len(ranges))(*[r.ctypes.data_as(C.POINTER(C.c_int32)) for r in ranges])
    sizes = (C.c_int64 * len(ranges))(*[len(r) for r in ranges])
    merged = np.empty(sum(len(r) for r in ranges), dtype=np.int32)
    R.ref_merge_ranges_i32(ptrs, sizes, len(ranges), merged.ctypes.data_as(C.POINTER(C.c_int32)))
    np.savez_compressed(os.path.join(OUT, "ref_algorithm.npz"), values=vals, modulus=np.array(7), partitioned=out,
                        pivot=np.array(pivot), merged=merged, **{"range_%d" % i: r for i, r in enumerate(ranges)})


def tile_cases():
    """(name, n, seed, kind, max_points, d, strategy, concurrency): inputs are regenerated from the seed."""
    out = []
    for sampler in ("RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED"):
        out.append((sampler + "_accurate_d32", 65536, 1, "uniform", 64, 32, "ACCURATE", 8))
        out.append((sampler + "_accurate_d250", 65536, 2, "uniform", 2000, 250, "ACCURATE", 8))
        out.append((sampler + "_fast_d250", 65536, 3, "uniform", 300, 250, "FAST", 8))
        out.append((sampler + "_accurate_clustered", 50000, 4, "clustered", 500, 120, "ACCURATE", 8))
    return out


def tile_input(n, seed, kind):
    if kind == "uniform":
        return O.generate_uniform(0x5C4A72A1D + seed, n)
    rng = np.random.default_rng(seed)
    return np.clip(np.vstack([rng.random((n // 2, 3)), 0.3 + 0.02 * rng.standard_normal((n - n // 2, 3))]), 0.0, 1.0)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def oracle_tiles():
    res = {}
    for name, n, seed, kind, max_points, d, strategy, conc in tile_cases():
        xyz = tile_input(n, seed, kind)
        spacing = O.spacing_from_diagonal(*UNIT, d)
        o = O.tile(xyz, *UNIT, getattr(O, name.split("_accurate")[0].split("_fast")[0]), max_points, spacing,
                   strategy=getattr(O, strategy), fast_concurrency=conc)
        assert o["status"] == 0, (name, o["status"])
        res[name] = {"n": n, "seed": seed, "kind": kind, "max_points_per_node": max_points, "diagonal_fraction": d,
                     "strategy": strategy, "fast_concurrency": conc, "input_sha256": digest(xyz),
                     "keys_sha256": digest(o["keys"]), "perm_sha256": digest(o["perm"]), "level_sha256": digest(o["level"]),
                     "dup_sha256": digest(o["dup"]), "num_nodes": int(o["stats"]["num_nodes"]),
                     "max_level": int(o["stats"]["max_level"]),
                     "level_histogram": np.bincount(o["level"].astype(np.int64) + 1).tolist()}
    json.dump(res, open(os.path.join(OUT, "oracle_tile_digests.json"), "w"), indent=1, sort_keys=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = O.ref()
    if R is None:
        sys.exit("oracle/_ref/libswzref.so missing: run `make -C oracle` where /root/reference exists")
    ref_morton_index(R)
    ref_random_grid(R)
    ref_algorithm(R)
    oracle_tiles()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
