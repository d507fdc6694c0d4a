"""Shared constants and dictionary keys (values fixed by the reference: thre3d_atom/utils/constants.py:1-28)."""
NUM_COORD_DIMENSIONS = 3
NUM_COLOUR_CHANNELS = 3
NUM_RGBA_CHANNELS = 4
NUM_ATTN_CHANNELS = 1

SEED = 42
ZERO_PLUS = 1e-10
INFINITY = 1e10

# keys of RenderOut.extra
EXTRA_DISPARITY = "disparity"
EXTRA_ACCUMULATED_WEIGHTS = "accumulated_weight"
EXTRA_POINT_DENSITIES = "point_densities"
EXTRA_POINT_OCCUPANCIES = "point_occupancies"
EXTRA_SAMPLE_INTERVALS = "deltas"
EXTRA_POINT_WEIGHTS = "point_weights"
EXTRA_POINT_DEPTHS = "point_depths"

# keys of the checkpoint's extra-info dictionary
CAMERA_BOUNDS = "camera_bounds"
CAMERA_INTRINSICS = "camera_intrinsics"
HEMISPHERICAL_RADIUS = "hemispherical_radius"
EXTRA_INFO = "extra_info"
