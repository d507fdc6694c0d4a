"""`thre3d_atom.utils.constants` of the reference, re-exported from thre3d_atom._keys (values fixed by interop)."""
from thre3d_atom import _keys
from thre3d_atom._keys import (  # noqa: F401
    INFINITY,
    NUM_ATTN_CHANNELS,
    NUM_COLOUR_CHANNELS,
    NUM_COORD_DIMENSIONS,
    NUM_RGBA_CHANNELS,
    SEED,
    ZERO_PLUS,
)

_keys.export(globals(), _keys.RENDER_EXTRA_KEYS,
             {k: _keys.CHECKPOINT_KEYS[k] for k in ("CAMERA_BOUNDS", "CAMERA_INTRINSICS", "HEMISPHERICAL_RADIUS", "EXTRA_INFO")})
