"""`thre3d_atom.data.constants` of the reference, re-exported from thre3d_atom._keys."""
from thre3d_atom import _keys

_keys.export(globals(), _keys.CAMERA_JSON_KEYS)
