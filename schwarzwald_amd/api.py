"""ctypes binding of libswz_gpu.so (include/swz_gpu.h).

Host-buffer methods take numpy arrays; *_device methods take raw device pointers (ints, e.g.
torch.Tensor.data_ptr()).  Every failure raises SwzError with the library's message; a missing
library raises at load time -- nothing here computes on the CPU.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

RANDOM_GRID, GRID_CENTER, MIN_DISTANCE, JITTERED = 0, 1, 2, 3
SAMPLERS = {"RANDOM_GRID": RANDOM_GRID, "GRID_CENTER": GRID_CENTER, "MIN_DISTANCE": MIN_DISTANCE,
            "JITTERED": JITTERED}
TAKE_ALL_WHEN_COUNT_BELOW_MAX_POINTS, ALWAYS_ADHERE_TO_MIN_SPACING = 0, 1
ACCURATE, FAST = 0, 1
ERR_BAD_ARG = 2
ERR_INTERNAL = 7
ERR_TILER_FAILED = 8
ERR_PEER_FAILED = 100  # raised by the multi-GPU driver on the ranks that did not fail themselves

_HERE = os.path.dirname(os.path.abspath(__file__))


class SwzError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("swz error %d: %s" % (code, message))
        self.code = code


class _TileParams(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("max_points_per_node", C.c_uint64), ("spacing_at_root", C.c_float),
                ("max_depth", C.c_uint32), ("strategy", C.c_int32), ("fast_concurrency", C.c_uint32),
                ("flags", C.c_uint32)]


ABI_VERSION = 3  # SWZ_ABI_VERSION of include/swz_gpu.h
FLAG_MIN_DISTANCE_PROPERTY = 1


class _TileStats(C.Structure):
    _fields_ = [("num_nodes", C.c_uint64), ("points_visited", C.c_uint64), ("max_level", C.c_int32),
                ("fast_start_levels", C.c_int32), ("num_levels", C.c_uint32), ("min_distance_rounds", C.c_uint32)]


class _ShardInfo(C.Structure):
    _fields_ = [("global_points", C.c_uint64), ("d_ghost_xyz", C.c_void_p), ("num_ghosts", C.c_uint64)]


class _TilerInfo(C.Structure):
    _fields_ = [("num_points", C.c_uint64), ("num_stored", C.c_uint64), ("num_nodes", C.c_uint64),
                ("num_batches", C.c_uint64), ("rekey_inversions", C.c_uint64), ("fast_start_levels", C.c_int32),
                ("staged_bytes", C.c_uint64), ("staged_wait_ms", C.c_double)]


class _TilerShardInfo(C.Structure):
    _fields_ = [("global_new_points", C.c_uint64), ("global_root_stored", C.c_uint64), ("d_ghost_xyz", C.c_void_p),
                ("num_ghosts", C.c_uint64)]


class _KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("algorithmic_bytes", C.c_uint64)]


@dataclass
class TileParams:
    sampler: int = MIN_DISTANCE
    max_points_per_node: int = 20000      # --max-points-per-node default, executable/main.cpp:230-232
    spacing_at_root: float = 0.0
    max_depth: int = 100                  # TilerProcess.cpp:624-629
    strategy: int = ACCURATE
    fast_concurrency: int = 8
    flags: int = 0                        # FLAG_MIN_DISTANCE_PROPERTY: see include/swz_gpu.h

    def _c(self):
        return _TileParams(self.sampler, self.max_points_per_node, self.spacing_at_root, self.max_depth,
                           self.strategy, self.fast_concurrency, self.flags)


@dataclass
class TileResult:
    keys: np.ndarray      # sorted Morton keys
    perm: np.ndarray      # original index of each sorted position
    level: np.ndarray     # node level that persists the point (-1 = root)
    dup: np.ndarray       # FAST duplicate mask (zeros for ACCURATE)
    stats: dict
    xyz_clamped: np.ndarray


def spacing_from_diagonal(bmin, bmax, diagonal_fraction):
    """TilerProcess.cpp:598-604: (float)(bounds.extent().length() / diagonal_fraction)."""
    e = np.asarray(bmax, dtype=np.float64) - np.asarray(bmin, dtype=np.float64)
    return float(np.float32(np.sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) / diagonal_fraction))


# attribute columns of a point batch: name -> (index = bit of BinaryPersistence's bitmask, numpy dtype, row width)
ATTRIBUTES = {
    "rgb": (0, np.uint8, 3), "normal": (1, np.float32, 3), "intensity": (2, np.uint16, 1),
    "classification": (3, np.uint8, 1), "edge_of_flight_line": (4, np.uint8, 1), "gps_time": (5, np.float64, 1),
    "number_of_returns": (6, np.uint8, 1), "return_number": (7, np.uint8, 1), "point_source_id": (8, np.uint16, 1),
    "scan_direction_flag": (9, np.uint8, 1), "scan_angle_rank": (10, np.int8, 1), "user_data": (11, np.uint8, 1),
}


class _LasLayout(C.Structure):
    _fields_ = [("scale", C.c_double * 3), ("offset", C.c_double * 3), ("min", C.c_double * 3), ("max", C.c_double * 3),
                ("point_format", C.c_uint32), ("record_bytes", C.c_uint32)]


class _AttributeColumns(C.Structure):
    _fields_ = [("column", C.c_void_p * 12)]


def _host_columns(attrs, n=None):
    """dict name -> array  =>  (swz_attribute_columns, keep-alive list of contiguous arrays)"""
    cols, keep = _AttributeColumns(), {}
    for name, arr in (attrs or {}).items():
        idx, dt, width = ATTRIBUTES[name]
        a = np.ascontiguousarray(arr, dtype=dt).reshape(-1, width) if width > 1 else np.ascontiguousarray(arr, dtype=dt).reshape(-1)
        if n is not None and a.shape[0] != n:
            raise ValueError("attribute %s has %d rows, expected %d" % (name, a.shape[0], n))
        keep[name] = a
        cols.column[idx] = a.ctypes.data
    return cols, keep


def device_columns(ptrs):
    """dict name -> device pointer  =>  swz_attribute_columns"""
    cols = _AttributeColumns()
    for name, ptr in (ptrs or {}).items():
        cols.column[ATTRIBUTES[name][0]] = int(ptr)
    return cols


def node_name(level, key):
    """"r" + octant digits of a node (TilingAlgorithms.cpp:139)."""
    buf = C.create_string_buffer(24)
    if load_library().swz_node_name(int(level), int(key), buf) != 0:
        raise ValueError("bad node level %d" % level)
    return buf.value.decode()


def node_name_entwine(level, key):
    """Entwine's "D-X-Y-Z" name of a node (OctreeNodeIndex.h:556-573)."""
    buf = C.create_string_buffer(72)
    if load_library().swz_node_name_entwine(int(level), int(key), buf) != 0:
        raise ValueError("bad node level %d" % level)
    return buf.value.decode()


def node_from_entwine_name(name):
    """(level, key) of an Entwine node name; ValueError for a malformed name or more than 21 levels."""
    lv, key = C.c_int8(), C.c_uint64()
    if load_library().swz_node_from_entwine_name(name.encode(), C.byref(lv), C.byref(key)) != 0:
        raise ValueError("not an Entwine node name: %r" % name)
    return int(lv.value), int(key.value)


def node_bounds(level, key, root_min, root_max):
    """Box of a node, descending octant by octant from the root box like get_octant_bounds."""
    mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
    if load_library().swz_node_bounds(int(level), int(key), _vec3(root_min), _vec3(root_max), mn, mx) != 0:
        raise ValueError("bad node level %d" % level)
    return list(mn), list(mx)


def node_geometric_error(level, spacing_at_root):
    """Cesium tileset geometricError of a node: spacing_at_root / 2^depth."""
    return float(load_library().swz_node_geometric_error(int(level), C.c_float(spacing_at_root)))


def required_morton_index_depth(sampler, node_level, root_min, root_max, spacing_at_root):
    """required_morton_index_depth of the reference (Sampling.cpp:29-62) as the library evaluates it."""
    L = load_library()
    L.swz_required_morton_index_depth.restype = C.c_int32
    L.swz_required_morton_index_depth.argtypes = [C.c_int, C.c_int32, _dp, _dp, C.c_float]
    return int(L.swz_required_morton_index_depth(int(sampler), int(node_level), _vec3(root_min), _vec3(root_max),
                                                 C.c_float(spacing_at_root)))


class _TilesetNode(C.Structure):
    _fields_ = [("level", C.c_int8), ("has_content", C.c_uint8), ("is_tileset_root", C.c_uint8), ("reserved", C.c_uint8),
                ("num_children", C.c_uint32), ("key", C.c_uint64), ("parent", C.c_int64), ("first_child", C.c_int64),
                ("geometric_error", C.c_double), ("bounds_min", C.c_double * 3), ("bounds_max", C.c_double * 3)]


def tileset_build(node_level, node_key, root_min, root_max, spacing_at_root, global_offset=None):
    """Tileset tree of a node table (Cesium3DTilesPersistence.cpp:80-156): list of dicts ordered by (level, key)."""
    L = load_library()
    lv = np.ascontiguousarray(node_level, dtype=np.int8)
    key = np.ascontiguousarray(node_key, dtype=np.uint64)
    off = _vec3(global_offset) if global_offset is not None else None
    num = C.c_uint64()
    args = (lv.shape[0], lv.ctypes.data_as(_i8p), key.ctypes.data_as(_u64p), _vec3(root_min), _vec3(root_max),
            C.c_float(spacing_at_root), off)
    if L.swz_tileset_build(*args, 0, None, C.byref(num)) != 0:
        raise ValueError("bad node table")
    buf = (_TilesetNode * max(int(num.value), 1))()
    if L.swz_tileset_build(*args, int(num.value), buf, C.byref(num)) != 0:
        raise ValueError("bad node table")
    return [dict(level=int(t.level), key=int(t.key), has_content=bool(t.has_content), is_tileset_root=bool(t.is_tileset_root),
                 num_children=int(t.num_children), parent=int(t.parent), first_child=int(t.first_child),
                 geometric_error=float(t.geometric_error), bounds_min=list(t.bounds_min), bounds_max=list(t.bounds_max))
            for t in buf[:int(num.value)]]


def bin_write_node(path, xyz, attrs=None, compressed=False):
    """BinaryPersistence::persist_points for one node (host only, no GPU needed)."""
    x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    cols, keep = _host_columns(attrs, x.shape[0])
    st = load_library().swz_bin_write_node(None, os.fsencode(path), x.shape[0], x.ctypes.data_as(_dp), C.byref(cols),
                                           int(bool(compressed)))
    if st != 0:
        raise SwzError(st, "swz_bin_write_node(%s) failed" % path)


def bin_read_node(path, compressed=False):
    """BinaryPersistence::retrieve_points: returns (xyz, dict of attribute arrays)."""
    L = load_library()
    mask, count = C.c_uint32(), C.c_uint64()
    st = L.swz_bin_read_header(None, os.fsencode(path), int(bool(compressed)), C.byref(mask), C.byref(count))
    if st != 0:
        raise SwzError(st, "swz_bin_read_header(%s) failed" % path)
    n = int(count.value)
    xyz = np.empty((n, 3), dtype=np.float64)
    out = {}
    for name, (idx, dt, width) in ATTRIBUTES.items():
        if mask.value & (1 << idx):
            out[name] = np.empty((n, width) if width > 1 else n, dtype=dt)
    cols, keep = _host_columns(out, n)
    st = L.swz_bin_read_node(None, os.fsencode(path), int(bool(compressed)), xyz.ctypes.data_as(_dp), C.byref(cols))
    if st != 0:
        raise SwzError(st, "swz_bin_read_node(%s) failed" % path)
    return xyz, keep


def library_path():
    return os.environ.get("SWZ_GPU_LIBRARY", os.path.join(_HERE, "lib", "libswz_gpu.so"))


# the all-gather callback of swz_shard_joint_root_begin: int (*)(void* arg, const void* mine, uint64_t bytes, void* all)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)

_lib = None
_dp, _u64p, _u32p = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
_i8p, _u8p = C.POINTER(C.c_int8), C.POINTER(C.c_uint8)


def _preload_hip_runtime():
    """A PyTorch-ROCm wheel ships its own libamdhip64.so and loads it by path, so a process that loads
    libswz_gpu.so first (resolving the system's libamdhip64.so.7) and imports torch later would run two HIP
    runtimes, the second of which finds no devices.  Loading torch's copy first (same SONAME) makes both use one.
    Without torch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Loads libswz_gpu.so; raises OSError when it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise OSError("libswz_gpu.so not found at %s -- build it with `make -C schwarzwald_amd/csrc` "
                      "(there is no CPU fallback)" % path)
    _preload_hip_runtime()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.swz_abi_version.restype = C.c_int
    if L.swz_abi_version() != ABI_VERSION:
        raise OSError("libswz_gpu.so has ABI version %d, this binding is written for %d (include/swz_gpu.h) -- rebuild it "
                      "with `make -C schwarzwald_amd/csrc`" % (L.swz_abi_version(), ABI_VERSION))
    L.swz_create.argtypes = [C.POINTER(vp), C.c_int]
    L.swz_destroy.argtypes = [vp]
    L.swz_last_error.restype = C.c_char_p
    L.swz_last_error.argtypes = [vp]
    L.swz_set_stream.argtypes = [vp, vp]
    L.swz_release_workspace.argtypes = [vp]
    L.swz_workspace_bytes.restype = C.c_uint64
    L.swz_workspace_bytes.argtypes = [vp]
    L.swz_morton_encode.argtypes = [vp, _dp, C.c_uint64, _dp, _dp, _u64p]
    L.swz_morton_encode_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, vp]
    L.swz_sort_by_key.argtypes = [vp, _u64p, C.c_uint64, _u32p, _u64p]
    L.swz_sort_by_key_device.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.swz_sample_points.argtypes = [vp, C.c_int, C.c_uint64, _u64p, _u32p, C.c_uint64, _dp, C.c_uint64, C.c_uint64,
                                    C.c_int32, _dp, _dp, C.c_float, C.c_int, _u8p, _u64p]
    L.swz_tile.argtypes = [vp, _dp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), _u64p, _u32p, _i8p, _u32p,
                           C.POINTER(_TileStats)]
    L.swz_tile_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), vp, vp, vp, vp,
                                  C.POINTER(_TileStats)]
    L.swz_tile_nodes_begin_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), _u64p, _u64p, C.POINTER(_TileStats)]
    L.swz_tile_nodes_end_device.argtypes = [vp, vp, vp, vp, C.c_uint64, _i8p, _u64p, _u64p, _u64p]
    L.swz_build_node_lists.argtypes = [vp, _u64p, _i8p, C.c_uint64, _u32p, C.c_uint64, _i8p, _u64p, _u64p, _u64p,
                                       _u64p]
    L.swz_generate_uniform_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    L.swz_build_node_lists_device.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint64, _i8p, _u64p, _u64p, _u64p, _u64p]
    L.swz_attribute_row_bytes.restype = C.c_uint32
    L.swz_attribute_row_bytes.argtypes = [C.c_int]
    cols = C.POINTER(_AttributeColumns)
    L.swz_gather_payload_device.argtypes = [vp, vp, vp, C.c_uint64, vp, cols, vp, cols]
    L.swz_bin_write_node.argtypes = [vp, C.c_char_p, C.c_uint64, _dp, cols, C.c_int]
    L.swz_bin_read_header.argtypes = [vp, C.c_char_p, C.c_int, _u32p, _u64p]
    L.swz_bin_read_node.argtypes = [vp, C.c_char_p, C.c_int, _dp, cols]
    L.swz_bin_persist_nodes.argtypes = [vp, C.c_char_p, C.c_uint64, _i8p, _u64p, _u64p, _u64p, _dp, cols, C.c_int]
    L.swz_node_name.argtypes = [C.c_int8, C.c_uint64, C.c_char_p]
    L.swz_node_name_entwine.argtypes = [C.c_int8, C.c_uint64, C.c_char_p]
    L.swz_node_from_entwine_name.argtypes = [C.c_char_p, _i8p, _u64p]
    L.swz_node_bounds.argtypes = [C.c_int8, C.c_uint64, _dp, _dp, _dp, _dp]
    L.swz_node_geometric_error.argtypes = [C.c_int8, C.c_float]
    L.swz_node_geometric_error.restype = C.c_double
    L.swz_tileset_build.argtypes = [C.c_uint64, _i8p, _u64p, _dp, _dp, C.c_float, _dp, C.c_uint64,
                                    C.POINTER(_TilesetNode), _u64p]
    L.swz_tileset_build.restype = C.c_int
    L.swz_las_decode_device.argtypes = [vp, vp, C.c_uint64, C.POINTER(_LasLayout), vp, cols]
    L.swz_partition_by_octant_device.argtypes = [vp, vp, C.c_uint64, vp, _u64p]
    L.swz_shard_begin_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), C.POINTER(_ShardInfo),
                                         _u64p]
    L.swz_shard_presort_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), C.c_uint64]
    L.swz_shard_root_taken_device.argtypes = [vp, vp]
    L.swz_shard_finish_device.argtypes = [vp, vp, vp, vp, C.POINTER(_TileStats)]
    L.swz_shard_fast_begin_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), _u32p]
    L.swz_shard_fast_run.argtypes = [vp, C.c_int32, _u64p]
    L.swz_shard_fast_root_candidates_device.argtypes = [vp, vp, vp]
    L.swz_shard_fast_set_root_device.argtypes = [vp, vp]
    L.swz_shard_fast_finish_device.argtypes = [vp, vp, vp, vp, vp, C.POINTER(_TileStats)]
    L.swz_sample_points_device.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, C.c_uint64, vp, C.c_uint64, C.c_uint64, C.c_int32, _dp, _dp,
                                           C.c_float, C.c_int, vp, _u64p]
    L.swz_tiler_create.argtypes = [vp, _dp, _dp, C.POINTER(_TileParams), C.c_uint64, C.POINTER(vp)]
    L.swz_tiler_destroy.argtypes = [vp]
    L.swz_tiler_add_batch_device.argtypes = [vp, vp, C.c_uint64, C.POINTER(_TileStats)]
    L.swz_tiler_stage_batch.argtypes = [vp, vp, C.c_uint64, cols]
    L.swz_tiler_tile_staged.argtypes = [vp, C.POINTER(_TileStats)]
    L.swz_tiler_add_batch.argtypes = [vp, vp, C.c_uint64, cols, C.POINTER(_TileStats)]
    L.swz_tiler_finalize.argtypes = [vp, C.POINTER(_TileStats)]
    L.swz_tiler_get_info.argtypes = [vp, C.POINTER(_TilerInfo)]
    L.swz_tiler_export_device.argtypes = [vp, vp, vp, vp]
    L.swz_tiler_node_table.argtypes = [vp, C.c_uint64, _i8p, _u64p, _u64p, _u64p, _u64p]
    L.swz_tiler_pools_device.argtypes = [vp, C.POINTER(vp), cols]
    L.swz_tiler_shard_begin_device.argtypes = [vp, vp, C.c_uint64, cols, C.POINTER(_TilerShardInfo), _u64p]
    L.swz_tiler_shard_finish.argtypes = [vp, C.POINTER(_TileStats)]
    L.swz_tiler_level_count.argtypes = [vp, C.c_int, _u64p]
    L.swz_tiler_poison.argtypes = [vp, C.c_char_p]
    L.swz_tiler_pool_residency.argtypes = [vp, _u64p, _u64p]
    L.swz_tiler_store_residency.argtypes = [vp, _u64p, _u64p]
    L.swz_shard_joint_root_possible.argtypes = [vp, C.POINTER(_TileParams), _dp, _dp]
    L.swz_shard_joint_root_begin.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp]
    L.swz_shard_joint_root_probe.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp, C.POINTER(C.c_int)]
    L.swz_shard_joint_root_meet.argtypes = [vp, C.c_int]
    L.swz_shard_joint_root_end.argtypes = [vp]
    L.swz_tiler_shard_fast_histogram.argtypes = [vp, _u32p]
    L.swz_fast_start_level_from_counts.argtypes = [_u64p, C.c_uint32, C.POINTER(C.c_int32)]
    L.swz_tiler_shard_set_start_level.argtypes = [vp, C.c_int32]
    L.swz_tiler_shard_fast_finalize_local.argtypes = [vp, C.POINTER(_TileStats)]
    L.swz_tiler_shard_fast_set_root.argtypes = [vp, vp]
    L.swz_tiler_level_positions_device.argtypes = [vp, C.c_int, vp]
    L.swz_host_alloc_pinned.argtypes = [C.c_uint64, C.POINTER(vp)]
    L.swz_host_free_pinned.argtypes = [vp]
    L.swz_profile_enable.argtypes = [vp, C.c_int]
    L.swz_profile_reset.argtypes = [vp]
    L.swz_profile_get.argtypes = [vp, C.POINTER(_KernelStat), C.c_uint32, _u32p]
    for name in ("swz_create", "swz_destroy", "swz_set_stream", "swz_release_workspace", "swz_morton_encode",
                 "swz_morton_encode_device", "swz_sort_by_key", "swz_sort_by_key_device", "swz_sample_points",
                 "swz_tile", "swz_tile_device", "swz_tile_nodes_begin_device", "swz_tile_nodes_end_device", "swz_build_node_lists", "swz_generate_uniform_device",
                 "swz_profile_enable", "swz_profile_reset", "swz_profile_get", "swz_partition_by_octant_device",
                 "swz_shard_begin_device", "swz_shard_root_taken_device", "swz_shard_finish_device",
                 "swz_build_node_lists_device", "swz_gather_payload_device", "swz_bin_write_node",
                 "swz_bin_read_header", "swz_bin_read_node", "swz_bin_persist_nodes", "swz_node_name",
                 "swz_las_decode_device", "swz_shard_presort_device", "swz_node_name_entwine",
                 "swz_node_from_entwine_name", "swz_node_bounds", "swz_tiler_create", "swz_tiler_destroy",
                 "swz_tiler_add_batch_device", "swz_tiler_stage_batch", "swz_tiler_tile_staged", "swz_tiler_add_batch",
                 "swz_tiler_finalize", "swz_tiler_get_info", "swz_tiler_export_device", "swz_tiler_node_table",
                 "swz_tiler_pools_device", "swz_host_alloc_pinned", "swz_host_free_pinned",
                 "swz_tiler_shard_begin_device", "swz_tiler_shard_finish", "swz_tiler_level_count",
                 "swz_tiler_level_positions_device", "swz_tiler_poison", "swz_tiler_pool_residency", "swz_tiler_store_residency",
                 "swz_tiler_shard_fast_histogram", "swz_fast_start_level_from_counts", "swz_tiler_shard_set_start_level",
                 "swz_tiler_shard_fast_finalize_local", "swz_tiler_shard_fast_set_root", "swz_shard_joint_root_possible",
                 "swz_shard_joint_root_begin", "swz_shard_joint_root_probe", "swz_shard_joint_root_meet", "swz_shard_joint_root_end",
                 "swz_shard_fast_begin_device", "swz_shard_fast_run", "swz_shard_fast_root_candidates_device",
                 "swz_shard_fast_set_root_device", "swz_shard_fast_finish_device", "swz_sample_points_device"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _vec3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _stats_dict(s):
    return dict(num_nodes=int(s.num_nodes), points_visited=int(s.points_visited), max_level=int(s.max_level),
                fast_start_levels=int(s.fast_start_levels), num_levels=int(s.num_levels),
                min_distance_rounds=int(s.min_distance_rounds))


class Context:
    """One swz_ctx (one GPU).  Not thread-safe: serialise calls per context."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        st = self._lib.swz_create(C.byref(self._ctx), int(device))
        if st != 0:
            raise SwzError(st, self._lib.swz_last_error(None).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.swz_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st):
        if st != 0:
            raise SwzError(st, self._lib.swz_last_error(self._ctx).decode())

    # ------------------------------------------------------------------ plumbing
    def set_stream(self, hip_stream):
        """Run on an existing HIP stream (an integer handle such as torch.cuda.current_stream().cuda_stream; 0 is the
        device's default stream, which is PyTorch's default).  None restores the context's own non-blocking stream."""
        handle = C.c_void_p(-1) if hip_stream is None else C.c_void_p(int(hip_stream))
        self._check(self._lib.swz_set_stream(self._ctx, handle))

    def set_option(self, name, value):
        """Debug / tuning switch of this context (what the SWZ_* environment variables seed at creation); None removes it."""
        self._lib.swz_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        self._check(self._lib.swz_set_option(self._ctx, name.encode(), None if value is None else str(value).encode()))

    def copy_to_host(self, d_ptr, nbytes):
        """nbytes of device memory as a numpy uint8 array (swz_copy_to_host: for pointers the library hands out)"""
        out = np.empty(int(nbytes), dtype=np.uint8)
        self._lib.swz_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(self._lib.swz_copy_to_host(self._ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(int(d_ptr)), int(nbytes)))
        return out

    def release_workspace(self):
        self._check(self._lib.swz_release_workspace(self._ctx))

    def workspace_bytes(self):
        return int(self._lib.swz_workspace_bytes(self._ctx))

    def profile_enable(self, on=True):
        self._check(self._lib.swz_profile_enable(self._ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.swz_profile_reset(self._ctx))

    def profile_get(self):
        buf = (_KernelStat * 64)()
        num = C.c_uint32()
        self._check(self._lib.swz_profile_get(self._ctx, buf, 64, C.byref(num)))
        return {buf[i].name.decode(): dict(launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                                           algorithmic_bytes=int(buf[i].algorithmic_bytes))
                for i in range(min(num.value, 64))}

    # ------------------------------------------------------------------ host-buffer entry points
    def morton_encode(self, xyz, bmin, bmax):
        """index_points<21>(ClampToBounds).  Returns (keys, clamped_xyz); the input is not modified."""
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3).copy()
        keys = np.empty(x.shape[0], dtype=np.uint64)
        self._check(self._lib.swz_morton_encode(self._ctx, x.ctypes.data_as(_dp), x.shape[0], _vec3(bmin),
                                                _vec3(bmax), keys.ctypes.data_as(_u64p)))
        return keys, x

    def sort_by_key(self, keys):
        """Returns (perm, sorted_keys): perm orders the input by (key, original index)."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        perm = np.empty(k.shape[0], dtype=np.uint32)
        ks = np.empty(k.shape[0], dtype=np.uint64)
        self._check(self._lib.swz_sort_by_key(self._ctx, k.ctypes.data_as(_u64p), k.shape[0],
                                              perm.ctypes.data_as(_u32p), ks.ctypes.data_as(_u64p)))
        return perm, ks

    def sample_points(self, sampler, max_points_per_node, keys, idx, xyz, node_key, node_level, root_min, root_max,
                      spacing_at_root, behaviour=TAKE_ALL_WHEN_COUNT_BELOW_MAX_POINTS):
        """sample_points(...) of Sampling.h:799-821 on a Morton-sorted range.  Returns the taken flags."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        i = np.ascontiguousarray(idx, dtype=np.uint32)
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        taken = np.zeros(k.shape[0], dtype=np.uint8)
        num = C.c_uint64()
        self._check(self._lib.swz_sample_points(self._ctx, sampler, max_points_per_node, k.ctypes.data_as(_u64p),
                                                i.ctypes.data_as(_u32p), k.shape[0], x.ctypes.data_as(_dp),
                                                x.shape[0], int(node_key), int(node_level), _vec3(root_min),
                                                _vec3(root_max), C.c_float(spacing_at_root), behaviour,
                                                taken.ctypes.data_as(_u8p), C.byref(num)))
        assert int(num.value) == int(taken.sum())
        return taken

    def tile(self, xyz, bmin, bmax, params):
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3).copy()
        n = x.shape[0]
        keys = np.empty(n, dtype=np.uint64)
        perm = np.empty(n, dtype=np.uint32)
        level = np.empty(n, dtype=np.int8)
        dup = np.zeros(n, dtype=np.uint32)
        stats = _TileStats()
        p = params._c()
        self._check(self._lib.swz_tile(self._ctx, x.ctypes.data_as(_dp), n, _vec3(bmin), _vec3(bmax), C.byref(p),
                                       keys.ctypes.data_as(_u64p), perm.ctypes.data_as(_u32p),
                                       level.ctypes.data_as(_i8p), dup.ctypes.data_as(_u32p), C.byref(stats)))
        return TileResult(keys, perm, level, dup, _stats_dict(stats), x)

    def build_node_lists(self, keys_sorted, level):
        k = np.ascontiguousarray(keys_sorted, dtype=np.uint64)
        lv = np.ascontiguousarray(level, dtype=np.int8)
        n = k.shape[0]
        order = np.empty(n, dtype=np.uint32)
        cap = max(n, 1)
        nl = np.empty(cap, dtype=np.int8)
        nk = np.empty(cap, dtype=np.uint64)
        no = np.empty(cap, dtype=np.uint64)
        nc = np.empty(cap, dtype=np.uint64)
        num = C.c_uint64()
        self._check(self._lib.swz_build_node_lists(self._ctx, k.ctypes.data_as(_u64p), lv.ctypes.data_as(_i8p), n,
                                                   order.ctypes.data_as(_u32p), cap, nl.ctypes.data_as(_i8p),
                                                   nk.ctypes.data_as(_u64p), no.ctypes.data_as(_u64p),
                                                   nc.ctypes.data_as(_u64p), C.byref(num)))
        m = int(num.value)
        return order, dict(level=nl[:m].copy(), key=nk[:m].copy(), offset=no[:m].copy(), count=nc[:m].copy())

    # ------------------------------------------------------------------ device-resident entry points
    def generate_uniform_device(self, seed, first_point, n, d_xyz):
        self._check(self._lib.swz_generate_uniform_device(self._ctx, int(seed), int(first_point), int(n),
                                                          C.c_void_p(d_xyz)))

    def morton_encode_device(self, d_xyz, n, bmin, bmax, d_keys):
        self._check(self._lib.swz_morton_encode_device(self._ctx, C.c_void_p(d_xyz), int(n), _vec3(bmin),
                                                       _vec3(bmax), C.c_void_p(d_keys)))

    def sort_by_key_device(self, d_keys, n, d_perm, d_keys_sorted=None):
        self._check(self._lib.swz_sort_by_key_device(self._ctx, C.c_void_p(d_keys), int(n), C.c_void_p(d_perm),
                                                     C.c_void_p(d_keys_sorted)))

    def tile_nodes_device(self, d_xyz, n, bmin, bmax, params, alloc):
        """One batch as node files (swz_tile_nodes_begin_device / _end_device): works where tile() fails with
        ERR_REROOT_UNSUPPORTED.  alloc(num_stored) returns the device pointers (d_keys, d_ids, d_level) for the files'
        contents (any may be None).  Returns (stats, node table: level / key / offset / count arrays, num_stored)."""
        p = params._c()
        stats = _TileStats()
        ns, nn = C.c_uint64(), C.c_uint64()
        self._check(self._lib.swz_tile_nodes_begin_device(self._ctx, C.c_void_p(d_xyz), int(n), _vec3(bmin), _vec3(bmax), C.byref(p),
                                                          C.byref(ns), C.byref(nn), C.byref(stats)))
        try:
            d_keys, d_ids, d_level = alloc(int(ns.value))
        except BaseException:
            self._lib.swz_tile_nodes_end_device(self._ctx, None, None, None, 0, None, None, None, None)
            raise
        cap = max(int(nn.value), 1)
        nl = np.empty(cap, dtype=np.int8)
        nk = np.empty(cap, dtype=np.uint64)
        no = np.empty(cap, dtype=np.uint64)
        nc = np.empty(cap, dtype=np.uint64)
        self._check(self._lib.swz_tile_nodes_end_device(self._ctx, C.c_void_p(d_keys), C.c_void_p(d_ids), C.c_void_p(d_level), cap,
                                                        nl.ctypes.data_as(_i8p), nk.ctypes.data_as(_u64p), no.ctypes.data_as(_u64p),
                                                        nc.ctypes.data_as(_u64p)))
        m = int(nn.value)
        return _stats_dict(stats), dict(level=nl[:m].copy(), key=nk[:m].copy(), offset=no[:m].copy(), count=nc[:m].copy()), int(ns.value)

    def tile_device(self, d_xyz, n, bmin, bmax, params, d_keys, d_perm, d_level, d_dup=None):
        stats = _TileStats()
        p = params._c()
        self._check(self._lib.swz_tile_device(self._ctx, C.c_void_p(d_xyz), int(n), _vec3(bmin), _vec3(bmax),
                                              C.byref(p), C.c_void_p(d_keys), C.c_void_p(d_perm),
                                              C.c_void_p(d_level), C.c_void_p(d_dup), C.byref(stats)))
        return _stats_dict(stats)

    def build_node_lists_device(self, d_keys_sorted, d_level, n, d_order):
        """Device version of build_node_lists: d_order (u32 x n, device) receives the order; returns the node table."""
        cap = max(int(n), 1)
        nl = np.empty(cap, dtype=np.int8)
        nk = np.empty(cap, dtype=np.uint64)
        no = np.empty(cap, dtype=np.uint64)
        nc = np.empty(cap, dtype=np.uint64)
        num = C.c_uint64()
        self._check(self._lib.swz_build_node_lists_device(self._ctx, C.c_void_p(d_keys_sorted), C.c_void_p(d_level), int(n),
                                                          C.c_void_p(d_order), cap, nl.ctypes.data_as(_i8p),
                                                          nk.ctypes.data_as(_u64p), no.ctypes.data_as(_u64p),
                                                          nc.ctypes.data_as(_u64p), C.byref(num)))
        m = int(num.value)
        return dict(level=nl[:m].copy(), key=nk[:m].copy(), offset=no[:m].copy(), count=nc[:m].copy())

    def gather_payload_device(self, d_perm, d_order, n, d_xyz, d_attrs_in, d_xyz_out, d_attrs_out):
        """Row i of every output column = row perm[order[i]] of the input column (dicts name -> device pointer)."""
        cin, cout = device_columns(d_attrs_in), device_columns(d_attrs_out)
        self._check(self._lib.swz_gather_payload_device(self._ctx, C.c_void_p(d_perm), C.c_void_p(d_order), int(n),
                                                        C.c_void_p(d_xyz), C.byref(cin), C.c_void_p(d_xyz_out),
                                                        C.byref(cout)))

    def las_decode_device(self, d_records, n, scale, offset, bmin, bmax, point_format, record_bytes, d_xyz, d_attrs=None):
        """Raw LAS 1.2 point records (formats 0-3, device memory) -> positions and attribute columns on the device."""
        lay = _LasLayout(_vec3(scale), _vec3(offset), _vec3(bmin), _vec3(bmax), int(point_format), int(record_bytes))
        cols = device_columns(d_attrs)
        self._check(self._lib.swz_las_decode_device(self._ctx, C.c_void_p(d_records), int(n), C.byref(lay),
                                                    C.c_void_p(d_xyz), C.byref(cols)))

    def bin_persist_nodes(self, directory, nodes, xyz, attrs=None, compressed=False):
        """One BinaryPersistence file per node of a node table from the gathered (host) payload."""
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        cols, keep = _host_columns(attrs, x.shape[0])
        nl = np.ascontiguousarray(nodes["level"], dtype=np.int8)
        nk = np.ascontiguousarray(nodes["key"], dtype=np.uint64)
        no = np.ascontiguousarray(nodes["offset"], dtype=np.uint64)
        nc = np.ascontiguousarray(nodes["count"], dtype=np.uint64)
        self._check(self._lib.swz_bin_persist_nodes(self._ctx, os.fsencode(directory), nl.shape[0], nl.ctypes.data_as(_i8p),
                                                    nk.ctypes.data_as(_u64p), no.ctypes.data_as(_u64p),
                                                    nc.ctypes.data_as(_u64p), x.ctypes.data_as(_dp), C.byref(cols),
                                                    int(bool(compressed))))

    # ------------------------------------------------------------------ sharded batches (one context per GPU)
    def partition_by_octant_device(self, d_keys, n, d_perm):
        """Groups point indices by level-0 octant (stable).  Returns the 8 octant counts."""
        counts = (C.c_uint64 * 8)()
        self._check(self._lib.swz_partition_by_octant_device(self._ctx, C.c_void_p(d_keys), int(n),
                                                             C.c_void_p(d_perm), counts))
        return [int(v) for v in counts]

    def shard_presort_device(self, d_xyz_local, n, bmin, bmax, params, ghost_capacity):
        """The ghost-independent part of shard_begin_device (index + sort + gather), done ahead of time."""
        p = params._c()
        self._check(self._lib.swz_shard_presort_device(self._ctx, C.c_void_p(d_xyz_local), int(n), _vec3(bmin), _vec3(bmax),
                                                       C.byref(p), int(ghost_capacity)))

    def shard_begin_device(self, d_xyz_local, n, bmin, bmax, params, global_points, d_ghost_xyz=None, num_ghosts=0):
        """Indexes + sorts the shard and samples the root node.  Returns how many local points the root took."""
        info = _ShardInfo(int(global_points), C.c_void_p(d_ghost_xyz), int(num_ghosts))
        p = params._c()
        taken = C.c_uint64()
        self._check(self._lib.swz_shard_begin_device(self._ctx, C.c_void_p(d_xyz_local), int(n), _vec3(bmin),
                                                     _vec3(bmax), C.byref(p), C.byref(info), C.byref(taken)))
        return int(taken.value)

    def shard_root_taken_device(self, d_xyz_out):
        self._check(self._lib.swz_shard_root_taken_device(self._ctx, C.c_void_p(d_xyz_out)))

    def shard_finish_device(self, d_keys, d_perm, d_level):
        stats = _TileStats()
        self._check(self._lib.swz_shard_finish_device(self._ctx, C.c_void_p(d_keys), C.c_void_p(d_perm),
                                                      C.c_void_p(d_level), C.byref(stats)))
        return _stats_dict(stats)

    # -- FAST (TilingAlgorithmV3) on a sharded batch (include/swz_gpu.h, swz_shard_fast_*)
    def shard_fast_begin_device(self, d_xyz_local, n, bmin, bmax, params):
        """Indexes + sorts the shard's points.  Returns its points per 6-octant prefix (8^6 counts)."""
        counts = np.zeros(1 << 18, dtype=np.uint32)
        p = params._c()
        self._check(self._lib.swz_shard_fast_begin_device(self._ctx, C.c_void_p(d_xyz_local), int(n), _vec3(bmin), _vec3(bmax),
                                                          C.byref(p), counts.ctypes.data_as(_u32p)))
        return counts

    def shard_fast_run(self, start_level):
        """The levels from the start level down and the local reconstruction.  Returns the points of this shard's level-0 nodes."""
        out = C.c_uint64()
        self._check(self._lib.swz_shard_fast_run(self._ctx, int(start_level), C.byref(out)))
        return int(out.value)

    def shard_fast_root_candidates_device(self, d_keys, d_xyz):
        self._check(self._lib.swz_shard_fast_root_candidates_device(self._ctx, C.c_void_p(d_keys), C.c_void_p(d_xyz)))

    def shard_fast_set_root_device(self, d_taken):
        self._check(self._lib.swz_shard_fast_set_root_device(self._ctx, C.c_void_p(d_taken)))

    def shard_fast_finish_device(self, d_keys, d_perm, d_level, d_dup):
        stats = _TileStats()
        self._check(self._lib.swz_shard_fast_finish_device(self._ctx, C.c_void_p(d_keys), C.c_void_p(d_perm), C.c_void_p(d_level),
                                                           C.c_void_p(d_dup), C.byref(stats)))
        return _stats_dict(stats)

    def sample_points_device(self, sampler, max_points_per_node, d_keys, d_idx, n, d_xyz, num_points, node_key, node_level, root_min,
                             root_max, spacing_at_root, behaviour, d_taken):
        """swz_sample_points on device buffers.  Returns the number of taken points."""
        num = C.c_uint64()
        self._check(self._lib.swz_sample_points_device(self._ctx, int(sampler), int(max_points_per_node), C.c_void_p(d_keys),
                                                       C.c_void_p(d_idx), int(n), C.c_void_p(d_xyz), int(num_points), int(node_key),
                                                       int(node_level), _vec3(root_min), _vec3(root_max), C.c_float(spacing_at_root),
                                                       int(behaviour), C.c_void_p(d_taken), C.byref(num)))
        return int(num.value)

    # -- the MIN_DISTANCE root of a sharded batch swept by all ranks at once (one process per GPU; include/swz_gpu.h)
    def shard_joint_root_possible(self, bmin, bmax, params):
        p = params._c()
        return bool(self._lib.swz_shard_joint_root_possible(self._ctx, C.byref(p), _vec3(bmin), _vec3(bmax)))

    def shard_joint_root_begin(self, shard, num_shards, all_gather_bytes):
        """all_gather_bytes(mine: bytes) -> list of every rank's bytes, in rank order (a collective of the driver)."""
        def cb(_arg, mine, nbytes, out):
            try:
                parts = all_gather_bytes(C.string_at(mine, nbytes))
                C.memmove(out, b"".join(parts), nbytes * len(parts))
                return 0
            except Exception:  # the library turns this into an error status on every rank
                return 1
        self._joint_cb = EXCHANGE_FN(cb)  # must outlive the calls that use it
        self._check(self._lib.swz_shard_joint_root_begin(self._ctx, int(shard), int(num_shards), self._joint_cb, None))

    def shard_joint_root_probe(self, shard, num_shards, all_gather_bytes):
        """Collective: True when every rank can map and read the lower ranks' device memory (the joint root needs that)."""
        def cb(_arg, mine, nbytes, out):
            try:
                parts = all_gather_bytes(C.string_at(mine, nbytes))
                C.memmove(out, b"".join(parts), nbytes * len(parts))
                return 0
            except Exception:
                return 1
        fn = EXCHANGE_FN(cb)
        usable = C.c_int(0)
        self._check(self._lib.swz_shard_joint_root_probe(self._ctx, int(shard), int(num_shards), fn, None, C.byref(usable)))
        return bool(usable.value)

    def shard_joint_root_meet(self, ok=True):
        self._check(self._lib.swz_shard_joint_root_meet(self._ctx, 1 if ok else 0))

    def shard_joint_root_end(self):
        self._check(self._lib.swz_shard_joint_root_end(self._ctx))
        self._joint_cb = None


def fast_start_level_from_counts(counts, fast_concurrency):
    """The start level TilingAlgorithmV3 derives from the 2^18-bin prefix histogram of a batch's keys
    (TilingAlgorithms.cpp:1473-1535); counts: the histogram summed over all shards."""
    L = load_library()
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    assert c.shape[0] == 1 << 18
    out = C.c_int32(-1)
    st = L.swz_fast_start_level_from_counts(c.ctypes.data_as(_u64p), int(fast_concurrency), C.byref(out))
    if st != 0:
        raise SwzError(st, "swz_fast_start_level_from_counts failed")
    return int(out.value)


def pinned_empty(shape, dtype):
    """numpy array over page-locked host memory (swz_host_alloc_pinned).  The memory is released when the array
    and all its views are gone; keep it alive while asynchronous copies from it are in flight."""
    import weakref
    L = load_library()
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dt.itemsize, 1)
    ptr = C.c_void_p()
    st = L.swz_host_alloc_pinned(nbytes, C.byref(ptr))
    if st != 0:
        raise SwzError(st, "swz_host_alloc_pinned(%d bytes) failed" % nbytes)
    buf = (C.c_char * nbytes).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=count).reshape(shape)
    weakref.finalize(arr, L.swz_host_free_pinned, C.c_void_p(ptr.value))
    return arr


class Tiler:
    """swz_tiler: one data set tiled batch after batch on one context (multi-batch semantics of the reference's
    TilingAlgorithm objects, TilingAlgorithms.cpp:50-109, 272-275, 1362-1453, 1661-1784)."""

    def __init__(self, ctx, bmin, bmax, params, capacity_hint=0):
        self._ctx = ctx
        self._lib = ctx._lib
        self._t = C.c_void_p()
        p = params._c()
        ctx._check(self._lib.swz_tiler_create(ctx._ctx, _vec3(bmin), _vec3(bmax), C.byref(p), int(capacity_hint),
                                              C.byref(self._t)))
        self._keep = []

    def poison(self, why="poisoned by the caller"):
        """Marks the tiler failed (what a failing batch does by itself): every later call raises ERR_TILER_FAILED."""
        self._ctx._check(self._lib.swz_tiler_poison(self._t, why.encode()))

    def pool_residency(self):
        """(bytes of the pools in device memory, bytes spilled to mapped page-locked host memory)"""
        dev, host = C.c_uint64(), C.c_uint64()
        self._ctx._check(self._lib.swz_tiler_pool_residency(self._t, C.byref(dev), C.byref(host)))
        return int(dev.value), int(host.value)

    def reserve(self, total_points):
        """room in the pools for total_points points of the data set, now (swz_tiler_reserve)"""
        self._lib.swz_tiler_reserve.argtypes = [C.c_void_p, C.c_uint64]
        self._ctx._check(self._lib.swz_tiler_reserve(self._t, int(total_points)))

    def store_residency(self):
        """(bytes of the node store in device memory, bytes placed in mapped page-locked host memory)"""
        dev, host = C.c_uint64(), C.c_uint64()
        self._ctx._check(self._lib.swz_tiler_store_residency(self._t, C.byref(dev), C.byref(host)))
        return int(dev.value), int(host.value)

    def close(self):
        if getattr(self, "_t", None):
            if getattr(self._ctx, "_ctx", None):  # a context that is gone took the tiler's memory with it
                self._lib.swz_tiler_destroy(self._t)
            self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_batch_device(self, d_xyz, n):
        stats = _TileStats()
        self._ctx._check(self._lib.swz_tiler_add_batch_device(self._t, C.c_void_p(d_xyz), int(n), C.byref(stats)))
        return _stats_dict(stats)

    def stage_batch(self, xyz, attrs=None):
        """Asynchronous copy of a host batch (use pinned_empty arrays for real overlap) into the device pools."""
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        cols, keep = _host_columns(attrs, x.shape[0])
        self._keep.append((x, keep))
        self._ctx._check(self._lib.swz_tiler_stage_batch(self._t, C.c_void_p(x.ctypes.data), x.shape[0], C.byref(cols)))

    def tile_staged(self):
        stats = _TileStats()
        self._ctx._check(self._lib.swz_tiler_tile_staged(self._t, C.byref(stats)))
        if self._keep:
            self._keep.pop(0)
        return _stats_dict(stats)

    def add_batch(self, xyz, attrs=None):
        self.stage_batch(xyz, attrs)
        return self.tile_staged()

    def finalize(self):
        stats = _TileStats()
        self._ctx._check(self._lib.swz_tiler_finalize(self._t, C.byref(stats)))
        return _stats_dict(stats)

    def info(self):
        i = _TilerInfo()
        self._ctx._check(self._lib.swz_tiler_get_info(self._t, C.byref(i)))
        return {k: getattr(i, k) for k, _ in _TilerInfo._fields_}

    def node_table(self):
        cap = max(int(self.info()["num_nodes"]), 1)
        nl = np.empty(cap, dtype=np.int8)
        nk = np.empty(cap, dtype=np.uint64)
        no = np.empty(cap, dtype=np.uint64)
        nc = np.empty(cap, dtype=np.uint64)
        num = C.c_uint64()
        self._ctx._check(self._lib.swz_tiler_node_table(self._t, cap, nl.ctypes.data_as(_i8p), nk.ctypes.data_as(_u64p),
                                                        no.ctypes.data_as(_u64p), nc.ctypes.data_as(_u64p), C.byref(num)))
        m = int(num.value)
        return dict(level=nl[:m].copy(), key=nk[:m].copy(), offset=no[:m].copy(), count=nc[:m].copy())

    def export_device(self, d_keys, d_ids, d_level):
        self._ctx._check(self._lib.swz_tiler_export_device(self._t, C.c_void_p(d_keys), C.c_void_p(d_ids),
                                                           C.c_void_p(d_level)))

    # -- one tiler per GPU of a multi-GPU run (schwarzwald_amd/sharded.py drives these)
    def shard_begin_device(self, d_xyz, n, d_attrs, global_new_points, global_root_stored, d_ghost_xyz=None, num_ghosts=0):
        """Indexes + sorts this shard's part of the batch and decides the root node.  Returns the points of the
        root's file on this shard afterwards."""
        info = _TilerShardInfo(int(global_new_points), int(global_root_stored), C.c_void_p(d_ghost_xyz), int(num_ghosts))
        cols = device_columns(d_attrs)
        out = C.c_uint64()
        self._ctx._check(self._lib.swz_tiler_shard_begin_device(self._t, C.c_void_p(d_xyz), int(n), C.byref(cols),
                                                                C.byref(info), C.byref(out)))
        return int(out.value)

    def shard_finish(self):
        stats = _TileStats()
        self._ctx._check(self._lib.swz_tiler_shard_finish(self._t, C.byref(stats)))
        return _stats_dict(stats)

    # -- FAST (TilingAlgorithmV3) on a sharded data set: the driver sums the shards' prefix histograms, derives the start
    # level, and at the end samples the root from the level-0 files of all shards (sharded.ShardedBatchTiler)
    def shard_fast_histogram(self):
        counts = np.zeros(1 << 18, dtype=np.uint32)
        self._ctx._check(self._lib.swz_tiler_shard_fast_histogram(self._t, counts.ctypes.data_as(_u32p)))
        return counts

    def shard_set_start_level(self, start_level):
        self._ctx._check(self._lib.swz_tiler_shard_set_start_level(self._t, int(start_level)))

    def shard_fast_finalize_local(self):
        stats = _TileStats()
        self._ctx._check(self._lib.swz_tiler_shard_fast_finalize_local(self._t, C.byref(stats)))
        return _stats_dict(stats)

    def shard_fast_set_root(self, d_taken):
        self._ctx._check(self._lib.swz_tiler_shard_fast_set_root(self._t, C.c_void_p(d_taken)))

    def level_count(self, level):
        out = C.c_uint64()
        self._ctx._check(self._lib.swz_tiler_level_count(self._t, int(level), C.byref(out)))
        return int(out.value)

    def level_positions_device(self, level, d_xyz_out):
        self._ctx._check(self._lib.swz_tiler_level_positions_device(self._t, int(level), C.c_void_p(d_xyz_out)))

    def pools_device(self):
        """(device pointer of the clamped positions by point id, dict name -> device pointer of the attribute pools)"""
        xyz = C.c_void_p()
        cols = _AttributeColumns()
        self._ctx._check(self._lib.swz_tiler_pools_device(self._t, C.byref(xyz), C.byref(cols)))
        attrs = {name: int(cols.column[idx]) for name, (idx, _, _) in ATTRIBUTES.items() if cols.column[idx]}
        return int(xyz.value or 0), attrs
