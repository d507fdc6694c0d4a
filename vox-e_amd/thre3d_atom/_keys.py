"""Single home of the numeric constants and dictionary keys shared by the package.

The VALUES are fixed by interoperability with TAU-VAILab/Vox-E (checkpoint dictionaries, RenderOut.extra,
`*_camera_params.json`); the public modules `thre3d_atom.utils.constants`,
`thre3d_atom.thre3d_reprs.constants` and `thre3d_atom.data.constants` re-export them under the reference's names.
"""

# ---- geometry / colour ----------------------------------------------------------------------------------
NUM_COORD_DIMENSIONS, NUM_COLOUR_CHANNELS, NUM_RGBA_CHANNELS, NUM_ATTN_CHANNELS = 3, 3, 4, 1
SEED = 42
ZERO_PLUS, INFINITY = 1e-10, 1e10  # epsilon of the disparity / AABB guard, "infinite" last sample interval

# ---- RenderOut.extra ------------------------------------------------------------------------------------
RENDER_EXTRA_KEYS = {
    "EXTRA_DISPARITY": "disparity",
    "EXTRA_ACCUMULATED_WEIGHTS": "accumulated_weight",
    "EXTRA_POINT_DENSITIES": "point_densities",
    "EXTRA_POINT_OCCUPANCIES": "point_occupancies",
    "EXTRA_SAMPLE_INTERVALS": "deltas",
    "EXTRA_POINT_WEIGHTS": "point_weights",
    "EXTRA_POINT_DEPTHS": "point_depths",
}

# ---- checkpoint dictionary ------------------------------------------------------------------------------
CHECKPOINT_KEYS = {
    "THRE3D_REPR": "thre3d_repr",
    "RENDER_PROCEDURE": "render_procedure",
    "RENDER_CONFIG": "render_config",
    "RENDER_CONFIG_TYPE": "render_config_type",
    "STATE_DICT": "state_dict",
    "CONFIG_DICT": "config_dict",
    "EXTRA_INFO": "extra_info",
    "CAMERA_BOUNDS": "camera_bounds",
    "CAMERA_INTRINSICS": "camera_intrinsics",
    "HEMISPHERICAL_RADIUS": "hemispherical_radius",
}
STATE_DICT_NAMES = {"u_DENSITIES": "_densities", "u_FEATURES": "_features", "u_ATTN": "attn"}

# ---- *_camera_params.json -------------------------------------------------------------------------------
CAMERA_JSON_KEYS = {
    "INTRINSIC": "intrinsic", "EXTRINSIC": "extrinsic", "BOUNDS": "bounds", "HEIGHT": "height", "WIDTH": "width",
    "FOCAL": "focal", "ROTATION": "rotation", "TRANSLATION": "translation", "DIRECTION": "dir",
}


def export(namespace: dict, *tables: dict) -> None:
    """bind every NAME -> value of the given tables in `namespace` (used by the re-exporting modules)"""
    for table in tables:
        namespace.update(table)
