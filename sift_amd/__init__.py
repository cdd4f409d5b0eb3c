"""MI355X-native SIFT engine: Python plumbing over the C ABI (include/sift_amd.h)."""
from .api import (  # noqa: F401
    Batch, Context, Group, SiftError, KEYPOINT_DTYPE, KNN2_DTYPE, COUNTS_DTYPE, LEVEL_DOG, LEVEL_GAUSS, LEVEL_GRAD_MAG, LEVEL_GRAD_DIR,
    EXPORTED_SYMBOLS, LIB_PATH, lib, device_count, min_dist_filter, num_octaves, octave_dims, synth_dead_leaves,
    synth_descriptors, synth_queries, shard_bounds, knn2_exchange_plan,
)
