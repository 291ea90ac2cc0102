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

_HERE = os.path.dirname(os.path.abspath(__file__))


class SwzError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("swz error %d: %s" % (code, message))
        self.code = code


class _TileParams(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("max_points_per_node", C.c_uint64), ("spacing_at_root", C.c_float),
                ("max_depth", C.c_uint32), ("strategy", C.c_int32), ("fast_concurrency", C.c_uint32)]


class _TileStats(C.Structure):
    _fields_ = [("num_nodes", C.c_uint64), ("points_visited", C.c_uint64), ("max_level", C.c_int32),
                ("fast_start_levels", C.c_int32), ("num_levels", C.c_uint32), ("min_distance_rounds", C.c_uint32)]


class _ShardInfo(C.Structure):
    _fields_ = [("global_points", C.c_uint64), ("d_ghost_xyz", C.c_void_p), ("num_ghosts", C.c_uint64)]


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

    def _c(self):
        return _TileParams(self.sampler, self.max_points_per_node, self.spacing_at_root, self.max_depth,
                           self.strategy, self.fast_concurrency)


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


def library_path():
    return os.environ.get("SWZ_GPU_LIBRARY", os.path.join(_HERE, "lib", "libswz_gpu.so"))


_lib = None
_dp, _u64p, _u32p = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
_i8p, _u8p = C.POINTER(C.c_int8), C.POINTER(C.c_uint8)


def load_library():
    """Loads libswz_gpu.so; raises OSError when it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise OSError("libswz_gpu.so not found at %s -- build it with `make -C schwarzwald_amd/csrc` "
                      "(there is no CPU fallback)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.swz_abi_version.restype = C.c_int
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
    L.swz_build_node_lists.argtypes = [vp, _u64p, _i8p, C.c_uint64, _u32p, C.c_uint64, _i8p, _u64p, _u64p, _u64p,
                                       _u64p]
    L.swz_generate_uniform_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    L.swz_partition_by_octant_device.argtypes = [vp, vp, C.c_uint64, vp, _u64p]
    L.swz_shard_begin_device.argtypes = [vp, vp, C.c_uint64, _dp, _dp, C.POINTER(_TileParams), C.POINTER(_ShardInfo),
                                         _u64p]
    L.swz_shard_root_taken_device.argtypes = [vp, vp]
    L.swz_shard_finish_device.argtypes = [vp, vp, vp, vp, C.POINTER(_TileStats)]
    L.swz_profile_enable.argtypes = [vp, C.c_int]
    L.swz_profile_reset.argtypes = [vp]
    L.swz_profile_get.argtypes = [vp, C.POINTER(_KernelStat), C.c_uint32, _u32p]
    for name in ("swz_create", "swz_destroy", "swz_set_stream", "swz_release_workspace", "swz_morton_encode",
                 "swz_morton_encode_device", "swz_sort_by_key", "swz_sort_by_key_device", "swz_sample_points",
                 "swz_tile", "swz_tile_device", "swz_build_node_lists", "swz_generate_uniform_device",
                 "swz_profile_enable", "swz_profile_reset", "swz_profile_get", "swz_partition_by_octant_device",
                 "swz_shard_begin_device", "swz_shard_root_taken_device", "swz_shard_finish_device"):
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
        self._check(self._lib.swz_set_stream(self._ctx, C.c_void_p(hip_stream)))

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

    def tile_device(self, d_xyz, n, bmin, bmax, params, d_keys, d_perm, d_level, d_dup=None):
        stats = _TileStats()
        p = params._c()
        self._check(self._lib.swz_tile_device(self._ctx, C.c_void_p(d_xyz), int(n), _vec3(bmin), _vec3(bmax),
                                              C.byref(p), C.c_void_p(d_keys), C.c_void_p(d_perm),
                                              C.c_void_p(d_level), C.c_void_p(d_dup), C.byref(stats)))
        return _stats_dict(stats)

    # ------------------------------------------------------------------ sharded batches (one context per GPU)
    def partition_by_octant_device(self, d_keys, n, d_perm):
        """Groups point indices by level-0 octant (stable).  Returns the 8 octant counts."""
        counts = (C.c_uint64 * 8)()
        self._check(self._lib.swz_partition_by_octant_device(self._ctx, C.c_void_p(d_keys), int(n),
                                                             C.c_void_p(d_perm), counts))
        return [int(v) for v in counts]

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
