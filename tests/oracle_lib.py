"""ctypes bindings of the parity oracle (oracle/liboracle.so) and of the partial reference build
(oracle/_ref/libswzref.so).  TEST INFRASTRUCTURE ONLY -- never imported by schwarzwald_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

RANDOM_GRID, GRID_CENTER, MIN_DISTANCE, JITTERED = 0, 1, 2, 3
TAKE_ALL_WHEN_BELOW_MAX, ALWAYS_ADHERE = 0, 1
ACCURATE, FAST = 0, 1
SAMPLER_NAMES = {RANDOM_GRID: "RANDOM_GRID", GRID_CENTER: "GRID_CENTER", MIN_DISTANCE: "MIN_DISTANCE",
                 JITTERED: "JITTERED"}
ERR_JITTER_GRID_TOO_SMALL, ERR_JITTER_NODE_TOO_DEEP, ERR_REROOT_UNSUPPORTED, ERR_BAD_ARG = -2, -3, -4, -5

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i8p = C.POINTER(C.c_int8)
_u8p = C.POINTER(C.c_uint8)


class TileParams(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("max_points_per_node", C.c_uint64), ("spacing_at_root", C.c_float),
                ("max_depth", C.c_uint32), ("strategy", C.c_int32), ("fast_concurrency", C.c_uint32)]


class TileStats(C.Structure):
    _fields_ = [("num_nodes", C.c_uint64), ("points_visited", C.c_uint64), ("max_level", C.c_int32),
                ("fast_start_levels", C.c_int32)]


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _vec3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
        L = C.CDLL(path)
        L.orc_calculate_morton_index.restype = C.c_uint64
        L.orc_calculate_morton_index.argtypes = [_dp, _dp, _dp, C.c_uint32]
        L.orc_calculate_morton_index_naive.restype = C.c_uint64
        L.orc_calculate_morton_index_naive.argtypes = [_dp, _dp, _dp, C.c_uint32]
        L.orc_index_points.restype = None
        L.orc_index_points.argtypes = [_dp, C.c_uint64, _dp, _dp, C.c_uint32, _u64p]
        L.orc_sort_by_key.restype = None
        L.orc_sort_by_key.argtypes = [_u64p, C.c_uint64, _u32p]
        L.orc_get_octant_bounds.restype = None
        L.orc_get_octant_bounds.argtypes = [C.c_uint8, _dp, _dp, _dp, _dp]
        L.orc_get_bounds_from_morton_index.restype = None
        L.orc_get_bounds_from_morton_index.argtypes = [C.c_uint64, C.c_uint32, _dp, _dp, C.c_uint32, _dp, _dp]
        L.orc_partition_points_into_child_octants.restype = None
        L.orc_partition_points_into_child_octants.argtypes = [_u64p, C.c_uint64, C.c_uint32, C.c_uint32, _u64p]
        L.orc_truncate_to_level.restype = C.c_uint64
        L.orc_truncate_to_level.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_get_octant_at_level.restype = C.c_uint8
        L.orc_get_octant_at_level.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_set_octant_at_level.restype = C.c_uint64
        L.orc_set_octant_at_level.argtypes = [C.c_uint64, C.c_uint32, C.c_uint8, C.c_uint32]
        L.orc_to_grid_index.restype = None
        L.orc_to_grid_index.argtypes = [C.c_uint64, C.c_uint32, _u64p]
        L.orc_get_prev_power_of_two.restype = C.c_uint32
        L.orc_get_prev_power_of_two.argtypes = [C.c_uint32]
        L.orc_required_morton_index_depth.restype = C.c_int32
        L.orc_required_morton_index_depth.argtypes = [C.c_int, C.c_int32, _dp, _dp, C.c_float]
        L.orc_sample_points.restype = C.c_int64
        L.orc_sample_points.argtypes = [C.c_int, C.c_uint64, _u64p, _u32p, C.c_uint64, _dp, C.c_uint64, C.c_int32,
                                        C.c_uint32, _dp, _dp, C.c_float, C.c_int]
        L.orc_sparse_grid_greedy.restype = None
        L.orc_sparse_grid_greedy.argtypes = [_dp, _u32p, C.c_uint64, _dp, _dp, C.c_float, _u8p]
        L.orc_tile.restype = C.c_int32
        L.orc_tile.argtypes = [_dp, C.c_uint64, _dp, _dp, C.POINTER(TileParams), _u64p, _u32p, _i8p, _u32p,
                               C.POINTER(TileStats)]
        L.orc_tile_mt.restype = C.c_int32
        L.orc_tile_mt.argtypes = [_dp, C.c_uint64, _dp, _dp, C.POINTER(TileParams), C.c_uint32, _u64p, _u32p, _i8p, _u32p,
                                  C.POINTER(TileStats)]
        L.orc_tile_mt_timed.restype = C.c_int32
        L.orc_tile_mt_timed.argtypes = [_dp, C.c_uint64, _dp, _dp, C.POINTER(TileParams), C.c_uint32, _u64p, _u32p, _i8p, _u32p,
                                        C.POINTER(TileStats), _dp]
        L.orc_stable_partition_take_multiples.restype = C.c_int64
        L.orc_stable_partition_take_multiples.argtypes = [C.POINTER(C.c_int32), C.c_int64, C.c_int32]
        L.orc_merge_ranges_i32.restype = None
        L.orc_merge_ranges_i32.argtypes = [C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64), C.c_int64,
                                           C.POINTER(C.c_int32)]
        L.orc_bin_persist_points.restype = C.c_int32
        L.orc_bin_persist_points.argtypes = [C.c_char_p, _u32p, C.c_uint64, _dp, C.POINTER(C.c_void_p)]
        L.orc_bin_retrieve_points.restype = C.c_int32
        L.orc_bin_retrieve_points.argtypes = [C.c_char_p, _u32p, _u64p, _dp, C.POINTER(C.c_void_p)]
        L.orc_las_decode.restype = C.c_int32
        L.orc_las_decode.argtypes = [_u8p, C.c_uint64, C.POINTER(LasLayout), _dp, C.POINTER(C.c_void_p)]
        L.orc_generate_uniform.restype = None
        L.orc_generate_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _dp]
        L.orc_tiler_create.restype = C.c_void_p
        L.orc_tiler_create.argtypes = [_dp, _dp, C.POINTER(TileParams)]
        L.orc_tiler_destroy.restype = None
        L.orc_tiler_destroy.argtypes = [C.c_void_p]
        L.orc_tiler_add_batch.restype = C.c_int32
        L.orc_tiler_add_batch.argtypes = [C.c_void_p, _dp, C.c_uint64]
        L.orc_tiler_finalize.restype = C.c_int32
        L.orc_tiler_finalize.argtypes = [C.c_void_p]
        L.orc_tiler_counts.restype = None
        L.orc_tiler_counts.argtypes = [C.c_void_p, _u64p, _u64p, _u64p, _u64p]
        L.orc_tiler_stats.restype = None
        L.orc_tiler_stats.argtypes = [C.c_void_p, C.POINTER(TileStats)]
        L.orc_tiler_export.restype = None
        L.orc_tiler_export.argtypes = [C.c_void_p, _i8p, _u64p, _u64p, _u64p, _u32p, _dp]
        _lib = L
    return _lib


_ref = None


def ref():
    """Partial reference build (MortonIndex.h / Algorithm.h compiled from /root/reference), or None."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libswzref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_truncate_to_level.restype = C.c_uint64
        R.ref_truncate_to_level.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        R.ref_get_octant_at_level.restype = C.c_uint8
        R.ref_get_octant_at_level.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        R.ref_set_octant_at_level.restype = C.c_uint64
        R.ref_set_octant_at_level.argtypes = [C.c_uint64, C.c_uint32, C.c_uint8, C.c_uint32]
        R.ref_morton_from_levels.restype = C.c_uint64
        R.ref_morton_from_levels.argtypes = [_u8p, C.c_uint32]
        R.ref_morton_ctor.restype = C.c_uint64
        R.ref_morton_ctor.argtypes = [C.c_uint64, C.c_uint32]
        R.ref_morton64_to_string.restype = None
        R.ref_morton64_to_string.argtypes = [C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint32]
        R.ref_morton64_from_string.restype = C.c_uint64
        R.ref_morton64_from_string.argtypes = [C.c_char_p]
        R.ref_partition_first_of_cell.restype = C.c_int64
        R.ref_partition_first_of_cell.argtypes = [_u64p, _u32p, C.c_int64, C.c_uint32]
        R.ref_stable_partition_take_multiples.restype = C.c_int64
        R.ref_stable_partition_take_multiples.argtypes = [C.POINTER(C.c_int32), C.c_int64, C.c_int32]
        R.ref_merge_ranges_i32.restype = None
        R.ref_merge_ranges_i32.argtypes = [C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64), C.c_int64,
                                           C.POINTER(C.c_int32)]
        R.ref_split_range_into_chunks.restype = None
        R.ref_split_range_into_chunks.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        _ref = R
    return _ref


# ----------------------------------------------------------------------------- numpy-level helpers
def morton_index(p, bmin, bmax, levels=21, naive=False):
    f = lib().orc_calculate_morton_index_naive if naive else lib().orc_calculate_morton_index
    return int(f(_vec3(p), _vec3(bmin), _vec3(bmax), levels))


def index_points(xyz, bmin, bmax, levels=21):
    """Returns (keys, clamped_xyz); the input is not modified."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).copy()
    n = xyz.shape[0]
    keys = np.empty(n, dtype=np.uint64)
    lib().orc_index_points(_ptr(xyz, _dp), n, _vec3(bmin), _vec3(bmax), levels, _ptr(keys, _u64p))
    return keys, xyz


def sort_by_key(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    perm = np.empty(keys.shape[0], dtype=np.uint32)
    lib().orc_sort_by_key(_ptr(keys, _u64p), keys.shape[0], _ptr(perm, _u32p))
    return perm


def octant_bounds(octant, bmin, bmax):
    omin, omax = (C.c_double * 3)(), (C.c_double * 3)()
    lib().orc_get_octant_bounds(octant, _vec3(bmin), _vec3(bmax), omin, omax)
    return list(omin), list(omax)


def bounds_from_morton_index(key, bmin, bmax, depth, levels=21):
    omin, omax = (C.c_double * 3)(), (C.c_double * 3)()
    lib().orc_get_bounds_from_morton_index(int(key), levels, _vec3(bmin), _vec3(bmax), depth, omin, omax)
    return list(omin), list(omax)


def partition_child_octants(sorted_keys, level, levels=21):
    k = np.ascontiguousarray(sorted_keys, dtype=np.uint64)
    off = np.zeros(9, dtype=np.uint64)
    lib().orc_partition_points_into_child_octants(_ptr(k, _u64p), k.shape[0], level, levels, _ptr(off, _u64p))
    return [int(x) for x in off]


def sample_points(sampler, max_points, keys, idx, xyz, node_key, node_level, bmin, bmax, spacing_at_root,
                  behaviour=TAKE_ALL_WHEN_BELOW_MAX, levels=21):
    """Returns (taken_count_or_error, keys_partitioned, idx_partitioned)."""
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    i = np.ascontiguousarray(idx, dtype=np.uint32).copy()
    x = np.ascontiguousarray(xyz, dtype=np.float64)
    r = lib().orc_sample_points(sampler, max_points, _ptr(k, _u64p), _ptr(i, _u32p), k.shape[0], _ptr(x, _dp),
                                int(node_key), node_level, levels, _vec3(bmin), _vec3(bmax),
                                C.c_float(spacing_at_root), behaviour)
    return int(r), k, i


def sparse_grid_greedy(xyz, idx, nmin, nmax, spacing):
    x = np.ascontiguousarray(xyz, dtype=np.float64)
    i = np.ascontiguousarray(idx, dtype=np.uint32)
    acc = np.zeros(i.shape[0], dtype=np.uint8)
    lib().orc_sparse_grid_greedy(_ptr(x, _dp), _ptr(i, _u32p), i.shape[0], _vec3(nmin), _vec3(nmax),
                                 C.c_float(spacing), _ptr(acc, _u8p))
    return acc


def tile(xyz, bmin, bmax, sampler, max_points_per_node, spacing_at_root, max_depth=100, strategy=ACCURATE,
         fast_concurrency=8, want_dup=None, threads=1):
    """Runs the oracle tiler.  Returns dict(status, keys, perm, level, dup, stats, xyz_clamped)."""
    x = np.ascontiguousarray(xyz, dtype=np.float64).copy()
    n = x.shape[0]
    keys = np.empty(n, dtype=np.uint64)
    perm = np.empty(n, dtype=np.uint32)
    level = np.empty(n, dtype=np.int8)
    dup = np.zeros(n, dtype=np.uint32)
    params = TileParams(sampler, max_points_per_node, spacing_at_root, max_depth, strategy, fast_concurrency)
    stats = TileStats()
    stage = (C.c_double * 3)()
    st = lib().orc_tile_mt_timed(_ptr(x, _dp), n, _vec3(bmin), _vec3(bmax), C.byref(params), int(threads), _ptr(keys, _u64p),
                                 _ptr(perm, _u32p), _ptr(level, _i8p), _ptr(dup, _u32p), C.byref(stats), stage)
    return dict(status=int(st), keys=keys, perm=perm, level=level, dup=dup, xyz_clamped=x,
                stage_seconds=dict(index=float(stage[0]), sort=float(stage[1]), tiling=float(stage[2])),
                stats=dict(num_nodes=int(stats.num_nodes), points_visited=int(stats.points_visited),
                           max_level=int(stats.max_level), fast_start_levels=int(stats.fast_start_levels)))


class Tiler:
    """Multi-batch oracle tiler (orc_tiler_*): one TilingAlgorithm object fed batch after batch."""

    def __init__(self, bmin, bmax, sampler, max_points_per_node, spacing_at_root, max_depth=100, strategy=ACCURATE,
                 fast_concurrency=8):
        params = TileParams(sampler, max_points_per_node, spacing_at_root, max_depth, strategy, fast_concurrency)
        self._h = lib().orc_tiler_create(_vec3(bmin), _vec3(bmax), C.byref(params))
        assert self._h

    def close(self):
        if self._h:
            lib().orc_tiler_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def add_batch(self, xyz):
        """Returns the status (0 = OK, negative = ORC_ERR_*)."""
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3).copy()
        return int(lib().orc_tiler_add_batch(self._h, _ptr(x, _dp), x.shape[0]))

    def finalize(self):
        return int(lib().orc_tiler_finalize(self._h))

    def counts(self):
        v = [C.c_uint64() for _ in range(4)]
        lib().orc_tiler_counts(self._h, *[C.byref(x) for x in v])
        return dict(num_nodes=int(v[0].value), num_stored=int(v[1].value), num_points=int(v[2].value),
                    unsorted_cached_nodes=int(v[3].value))

    def stats(self):
        s = TileStats()
        lib().orc_tiler_stats(self._h, C.byref(s))
        return dict(num_nodes=int(s.num_nodes), points_visited=int(s.points_visited), max_level=int(s.max_level),
                    fast_start_levels=int(s.fast_start_levels))

    def export(self):
        """dict(level, key, offset, count, ids, xyz): node table ordered by (level, key) and the ids per node."""
        c = self.counts()
        nn, ns, npts = c["num_nodes"], c["num_stored"], c["num_points"]
        lv = np.empty(max(nn, 1), dtype=np.int8)
        key = np.empty(max(nn, 1), dtype=np.uint64)
        off = np.empty(max(nn, 1), dtype=np.uint64)
        cnt = np.empty(max(nn, 1), dtype=np.uint64)
        ids = np.empty(max(ns, 1), dtype=np.uint32)
        xyz = np.empty((max(npts, 1), 3), dtype=np.float64)
        lib().orc_tiler_export(self._h, _ptr(lv, _i8p), _ptr(key, _u64p), _ptr(off, _u64p), _ptr(cnt, _u64p),
                               _ptr(ids, _u32p), _ptr(xyz, _dp))
        return dict(level=lv[:nn], key=key[:nn], offset=off[:nn], count=cnt[:nn], ids=ids[:ns], xyz=xyz[:npts])


# attribute columns: name -> (bit of BinaryPersistence's bitmask, dtype, row width)
ATTRIBUTES = {
    "rgb": (0, np.uint8, 3), "normal": (1, np.float32, 3), "intensity": (2, np.uint16, 1),
    "classification": (3, np.uint8, 1), "edge_of_flight_line": (4, np.uint8, 1), "gps_time": (5, np.float64, 1),
    "number_of_returns": (6, np.uint8, 1), "return_number": (7, np.uint8, 1), "point_source_id": (8, np.uint16, 1),
    "scan_direction_flag": (9, np.uint8, 1), "scan_angle_rank": (10, np.int8, 1), "user_data": (11, np.uint8, 1),
}


def _columns(attrs):
    cols = (C.c_void_p * 12)()
    keep = {}
    for name, arr in (attrs or {}).items():
        idx, dt, width = ATTRIBUTES[name]
        a = np.ascontiguousarray(arr, dtype=dt)
        keep[name] = a
        cols[idx] = a.ctypes.data
    return cols, keep


def bin_persist_points(path, point_refs, xyz, attrs=None):
    """BinaryPersistence::persist_points (uncompressed) of the points `point_refs` of a batch."""
    refs = np.ascontiguousarray(point_refs, dtype=np.uint32)
    x = np.ascontiguousarray(xyz, dtype=np.float64)
    cols, keep = _columns(attrs)
    st = lib().orc_bin_persist_points(os.fsencode(path), _ptr(refs, _u32p), refs.shape[0], _ptr(x, _dp), cols)
    assert st == 0, st


def bin_retrieve_points(path):
    """BinaryPersistence::retrieve_points: (bitmask, xyz, attrs)."""
    mask, count = C.c_uint32(), C.c_uint64()
    st = lib().orc_bin_retrieve_points(os.fsencode(path), C.byref(mask), C.byref(count), None, None)
    assert st == 0, st
    n = int(count.value)
    xyz = np.empty((n, 3), dtype=np.float64)
    out = {}
    for name, (idx, dt, width) in ATTRIBUTES.items():
        if mask.value & (1 << idx):
            out[name] = np.empty((n, width) if width > 1 else n, dtype=dt)
    cols, keep = _columns(out)
    st = lib().orc_bin_retrieve_points(os.fsencode(path), C.byref(mask), C.byref(count), _ptr(xyz, _dp), cols)
    assert st == 0, st
    return int(mask.value), xyz, keep


class LasLayout(C.Structure):
    _fields_ = [("scale", C.c_double * 3), ("offset", C.c_double * 3), ("min", C.c_double * 3), ("max", C.c_double * 3),
                ("point_format", C.c_uint32), ("record_bytes", C.c_uint32)]


LAS_ATTRIBUTES = [a for a in ATTRIBUTES if a != "normal"]


def las_decode(records, n, scale, offset, bmin, bmax, point_format, record_bytes, names=LAS_ATTRIBUTES):
    """position_from_las_point + las_read_points_into on raw LAS 1.2 point records: (xyz, attrs)."""
    rec = np.ascontiguousarray(records, dtype=np.uint8)
    lay = LasLayout(_vec3(scale), _vec3(offset), _vec3(bmin), _vec3(bmax), point_format, record_bytes)
    xyz = np.empty((n, 3), dtype=np.float64)
    out = {}
    for name in names:
        idx, dt, width = ATTRIBUTES[name]
        out[name] = np.zeros((n, width) if width > 1 else n, dtype=dt)
    cols, keep = _columns(out)
    st = lib().orc_las_decode(_ptr(rec, _u8p), n, C.byref(lay), _ptr(xyz, _dp), cols)
    assert st == 0, st
    return xyz, keep


def generate_uniform(seed, n, first_point=0):
    xyz = np.empty((n, 3), dtype=np.float64)
    lib().orc_generate_uniform(seed, first_point, n, _ptr(xyz, _dp))
    return xyz


def spacing_from_diagonal(bmin, bmax, diagonal_fraction):
    """TilerProcess.cpp:598-604: (float)(cubic.extent().length() / diagonal_fraction)."""
    e = np.asarray(bmax, dtype=np.float64) - np.asarray(bmin, dtype=np.float64)
    length = np.sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2])
    return float(np.float32(length / diagonal_fraction))
