"""schwarzwald_amd -- MI355X-native implementation of the Schwarzwald tiler hot path.

Morton encode -> radix sort by Morton key -> octree-node partition -> per-node LOD sampling
(RANDOM_GRID / GRID_CENTER / MIN_DISTANCE / JITTERED), written as HIP kernels for gfx950 behind the
C ABI of include/swz_gpu.h.  This Python package is only the ctypes binding of that ABI plus the
multi-GPU sharding driver; there is no CPU implementation in the product.
"""
from .api import (ACCURATE, FLAG_MIN_DISTANCE_PROPERTY, ALWAYS_ADHERE_TO_MIN_SPACING, FAST, GRID_CENTER, JITTERED, MIN_DISTANCE, RANDOM_GRID,
                  SAMPLERS, TAKE_ALL_WHEN_COUNT_BELOW_MAX_POINTS, Context, SwzError, TileParams, TileResult,
                  ATTRIBUTES, bin_read_node, bin_write_node, library_path, load_library, node_bounds,
                  node_from_entwine_name, node_geometric_error, node_name, node_name_entwine,
                  spacing_from_diagonal, Tiler, pinned_empty, tileset_build)

__all__ = ["tileset_build", "FLAG_MIN_DISTANCE_PROPERTY", "Context", "Tiler", "pinned_empty", "SwzError", "TileParams", "TileResult", "load_library", "library_path", "SAMPLERS",
           "RANDOM_GRID", "GRID_CENTER", "MIN_DISTANCE", "JITTERED", "ACCURATE", "FAST",
           "TAKE_ALL_WHEN_COUNT_BELOW_MAX_POINTS", "ALWAYS_ADHERE_TO_MIN_SPACING", "spacing_from_diagonal", "ATTRIBUTES", "bin_write_node", "bin_read_node",
           "node_name", "node_name_entwine", "node_from_entwine_name", "node_bounds", "node_geometric_error"]
