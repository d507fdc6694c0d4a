"""Checkpoint dictionary keys (reference: thre3d_atom/thre3d_reprs/constants.py:1-16)."""
THRE3D_REPR = "thre3d_repr"
RENDER_PROCEDURE = "render_procedure"
RENDER_CONFIG = "render_config"
RENDER_CONFIG_TYPE = "render_config_type"
STATE_DICT = "state_dict"
CONFIG_DICT = "config_dict"

# state-dict names of the voxel grid's tensors
u_DENSITIES = "_densities"
u_FEATURES = "_features"
u_ATTN = "attn"
